"""Demo IoU family -- API mirror of the reference's demos/yolov3_u/utils/iou.py: same as the library file except
D/CIoU use un-halved centre sums and a minus sign.  Runs the same HIP kernels with variant=1."""
from ....detection.tools import IOU as _L

cal_iou, cal_iou_batch = _L.cal_iou, _L.cal_iou_batch
xyxy_iou, xywh_iou, wh_iou = _L.xyxy_iou, _L.xywh_iou, _L.wh_iou
xyxy_iou_batch, xywh_iou_batch, wh_iou_batch = _L.xyxy_iou_batch, _L.xywh_iou_batch, _L.wh_iou_batch
GIOU, GIOU_batch = _L.GIOU, _L.GIOU_batch


def DIOU(box1, box2, mode='xyxy', eps=1e-7):
    return _L._pair(box1, box2, 2, mode, eps, variant=1).reshape(-1, 1)


def DIOU_batch(box1, box2, mode='xyxy', eps=1e-7):
    return _L._batch(box1, box2, 2, mode, eps, variant=1)


def CIOU(box1, box2, mode='xyxy', eps=1e-7):
    return _L._pair(box1, box2, 3, mode, eps, variant=1).reshape(-1, 1)


def CIOU_batch(box1, box2, mode='xyxy', eps=1e-7):
    return _L._batch(box1, box2, 3, mode, eps, variant=1)
