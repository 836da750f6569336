"""Non-maximum suppression -- API mirror of the reference's detection/tools/NMS.py, on the HIP kernels
(``fva_nms_candidates`` + ``fva_nms_select``, which also stand in for the torchvision.ops.nms call, NMS.py:18)."""
import torch

from ...detect_ops import nms_batch, NMS_LIBRARY

__all__ = ['non_max_suppression', 'non_max_suppression_images']


def non_max_suppression_images(predictions, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """All images of a batch in one pass: predictions [B,R,5+C] -> list of (scores [n,1], categories [n,1], boxes [n,4])."""
    res = []
    for det, _ in nms_batch(predictions, conf_thres, iou_thres, max_det, NMS_LIBRARY):
        res.append((det[:, 4:5], det[:, 5:6].long(), det[:, 0:4]))
    return res


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """prediction [R,5+C] = x, y, w, h, obj, class scores of ONE image (NMS.py:5-23).  Class-agnostic; the score of a row
    is max_c(cls_c * obj).  Unlike the reference this does not scale ``prediction[:, 5:]`` in place."""
    return non_max_suppression_images(prediction.unsqueeze(0), conf_thres, iou_thres, max_det)[0]
