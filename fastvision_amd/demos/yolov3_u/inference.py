"""API mirror of the detection post-processing of the reference's demos/yolov3_u/inference.py (``postProcess``,
:58-120).  Image loading / letterboxing (``preProcess``, cv2 + albumentations) belongs to the input pipeline (scope
row f-3) and is not part of this package."""
import torch

from ...detect_ops import yolo_decode, nms_batch, NMS_DEMO

__all__ = ['postProcess', 'anchor_fn']


def anchor_fn(device='cuda'):
    """the demo's three feature-scale anchor sets, small grid first (inference.py:130-137)"""
    small = torch.tensor([[116, 90], [156, 198], [373, 326]]) / 32
    medium = torch.tensor([[30, 61], [62, 45], [59, 119]]) / 16
    large = torch.tensor([[10, 13], [16, 30], [33, 23]]) / 8
    return small.to(device), medium.to(device), large.to(device)


def postProcess(predict_layers, strides, anchors, conf_thres, iou_thres, resize_ratio, padding_left, padding_top, ori_width,
                ori_height):
    """predict_layers: the model's three [1, A*(5+C), h, w] outputs; returns (scores [n,1], categories [n,1], boxes [n,4])
    in original-image pixels.  One decode kernel (un-letterbox, clamps and the 5-pixel size filter included) and one NMS
    pass; the reference's in-place writes into ``predict_layers`` are not reproduced."""
    heads, anchor_lists = [], []
    for predict, anchor in zip(predict_layers, anchors):
        A = anchor.size(0)
        bs, c, h, w = predict.shape
        heads.append(predict.unflatten(1, (A, c // A)).permute(0, 1, 3, 4, 2))     # [bs,A,h,w,K] view, no copy
        anchor_lists.append([(float(a[0]), float(a[1])) for a in anchor.detach().cpu()])
    rows = yolo_decode(heads, anchor_lists, strides, variant=1,
                       letterbox=(resize_ratio, padding_left, padding_top, ori_width, ori_height, 5.0))
    assert rows.shape[0] == 1, 'postProcess handles one image, as the reference does'
    det, _ = nms_batch(rows, conf_thres, iou_thres, 300, NMS_DEMO)[0]
    return det[:, 4:5], det[:, 5:6], det[:, :4]
