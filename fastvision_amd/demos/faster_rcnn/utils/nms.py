"""non_max_suppression of the Faster R-CNN demo (demos/faster_rcnn/utils/nms.py:5-40) on the device NMS kernels: rows
[xmin, ymin, xmax, ymax, category, score] -> rows above ``conf_thres`` that survive per-class greedy suppression (a 4096 * category
coordinate gap), best score first, at most ``max_det``."""
import torch

from ....detect_ops import NMS_DEMO, nms_batch

__all__ = ['non_max_suppression']


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, max_det=300):
    if prediction.size(0) == 0:
        return torch.zeros((0, 6), device=prediction.device)
    n = prediction.size(0)
    ncls = int(prediction[:, 4].max().item()) + 1
    # the kernels take [boxes, objectness, class scores]: score = the row's score, category = arg-max of a one-hot row
    rows = torch.zeros((1, n, 5 + max(ncls, 1)), dtype=torch.float32, device=prediction.device)
    rows[0, :, :4] = prediction[:, :4]
    rows[0, :, 4] = prediction[:, 5]
    rows[0, torch.arange(n, device=prediction.device), 5 + prediction[:, 4].long()] = 1.0
    det, _ = nms_batch(rows, conf_thres, iou_thres, max_det, NMS_DEMO)[0]
    return torch.cat([det[:, :4], det[:, 5:6], det[:, 4:5]], dim=1)          # back to [xyxy, category, score]
