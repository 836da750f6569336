"""API mirror of the reference's demos/yolov3_u/utils/nms.py on the HIP kernels (``fva_nms_candidates`` +
``fva_nms_select``; the torchvision.ops.nms call of nms.py:48,92 included)."""
import torch

from ....detect_ops import nms_batch, NMS_DEMO, NMS_DEMO_BATCH

__all__ = ['non_max_suppression', 'non_max_suppression_batch']


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """prediction [R,5+C] = xmin, ymin, xmax, ymax, obj, class scores of one image -> [n,6] = xyxy, obj, category
    (nms.py:5-52: per-class NMS through a 4096 * category coordinate gap, ranked by objectness).  The reference also
    scales ``prediction[:, 5:]`` by the objectness in place; this does not touch its input."""
    return nms_batch(prediction.unsqueeze(0), conf_thres, iou_thres, max_det, NMS_DEMO)[0][0]


def non_max_suppression_batch(prediction_batch, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """list of [R,5+C] = x, y, w, h, obj, class scores -> list of CPU [n,6] = xyxy, obj*cls, category (nms.py:54-98)."""
    if len(prediction_batch) == 0:
        return []
    if len({tuple(p.shape) for p in prediction_batch}) == 1:
        dets = nms_batch(torch.stack(list(prediction_batch)), conf_thres, iou_thres, max_det, NMS_DEMO_BATCH)
    else:
        dets = [nms_batch(p.unsqueeze(0), conf_thres, iou_thres, max_det, NMS_DEMO_BATCH)[0] for p in prediction_batch]
    return [d.detach().cpu() for d, _ in dets]
