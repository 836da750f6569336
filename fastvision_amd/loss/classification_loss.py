"""Classification losses -- API mirror of the reference's loss/classification_loss.py.

These two classes are the thin, standalone API surface (``CrossEntropyLoss`` is the CPU plumbing loss of
BASELINE config 1).  On the accelerated path ``Yolov3Loss`` does NOT call them: its class / objectness BCE
terms and their gradients are computed inside the fused HIP loss kernels (csrc/loss.hip), which restate
``BiCrossEntropyLoss`` exactly (1e-8 inside both logs, sum / numel).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['one_hot', 'CrossEntropyLoss', 'BiCrossEntropyLoss']


def one_hot(y, num_classes):
    """datasets/common/id_2_onehot.py:10-15 (torch branch)."""
    col = y.view(-1, 1).long()
    return torch.zeros((col.size(0), num_classes)).to(y).scatter_(1, col, 1)


class CrossEntropyLoss(nn.Module):
    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, y_pre, y_true, weights=None):
        target = one_hot(y_true, y_pre.size(-1)).float()
        loss = -torch.sum(target * F.log_softmax(y_pre, dim=-1), dim=1)
        if weights is not None:
            loss = loss * weights
        return torch.mean(loss) if self.reduction == 'mean' else torch.sum(loss)


class BiCrossEntropyLoss(nn.Module):
    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, y_pre, y_true, already_sigmoid=False, weights=None):
        if y_pre.size(-1) > 1:
            target = one_hot(y_true, y_pre.size(-1)).float().view(-1, 1)
        else:
            target = y_true.float().view(-1, 1)
        p = y_pre.view(-1, 1)
        if not already_sigmoid:
            p = p.sigmoid()
        loss = torch.sum(-target * torch.log(p + 1e-8) - (1 - target) * torch.log(1 - p + 1e-8), dim=1)
        if weights is not None:
            loss = loss * weights
        return torch.sum(loss) / p.numel() if self.reduction == 'mean' else torch.sum(loss)
