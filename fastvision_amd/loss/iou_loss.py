"""IoU losses -- API mirror of the reference's loss/iou_loss.py (1 - {IoU,GIoU,DIoU,CIoU}, mean / sum).
The IoU values and their gradients come from the HIP kernel behind ``detection.tools``."""
import torch
import torch.nn as nn

from ..detection.tools import CIOU, DIOU, GIOU, cal_iou

__all__ = ['IOULoss', 'GIOULoss', 'DIOULoss', 'CIOULoss']


class _IoULossBase(nn.Module):
    fn = None

    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, y_pre, y_true, weights=None, mode='xyxy'):
        loss = 1 - type(self).fn(y_pre, y_true, mode=mode)
        if weights is not None:
            loss = loss * weights
        return torch.mean(loss) if self.reduction == 'mean' else torch.sum(loss)


class IOULoss(_IoULossBase):
    fn = staticmethod(cal_iou)


class GIOULoss(_IoULossBase):
    fn = staticmethod(GIOU)


class DIOULoss(_IoULossBase):
    fn = staticmethod(DIOU)


class CIOULoss(_IoULossBase):
    fn = staticmethod(CIOU)
