"""Oracle: box conversions, grids and the IoU family (torch tensors, CPU fp32).

Restates the *torch branch* of the reference's tools, quirks included
(SURVEY.md App. B-1..4).  Each function names the reference lines it follows.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math

import torch

EPS = 1e-7


# --------------------------------------------------------------------------- boxes
def xywh2xyxy(b):
    """detection/tools/BOX.py:4-10 -- centre/size -> corners, column-wise."""
    half_w, half_h = b[:, 2] / 2, b[:, 3] / 2
    return torch.stack([b[:, 0] - half_w, b[:, 1] - half_h, b[:, 0] + half_w, b[:, 1] + half_h], dim=1)


def xyxy2xywh(b):
    """detection/tools/BOX.py:12-18."""
    return torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], dim=1)


def xyxy2xywhn(b, height, width):
    """detection/tools/BOX.py:20-26 (note the argument order height, width)."""
    return torch.stack([((b[:, 0] + b[:, 2]) / 2) / width, ((b[:, 1] + b[:, 3]) / 2) / height,
                        (b[:, 2] - b[:, 0]) / width, (b[:, 3] - b[:, 1]) / height], dim=1)


def grid(height, width, mode='xy'):
    """detection/tools/GRID.py:18-29 (torch branch) == demos/yolov3_u/utils/box.py:36-47.

    mode='xy' -> [H, W, 2] with [..., 0] = x (column index), [..., 1] = y (row index);
    any other mode -> the [W, H, 2] transpose of it.
    """
    ys = torch.arange(height).view(height, 1).expand(height, width)
    xs = torch.arange(width).view(1, width).expand(height, width)
    g = torch.stack([xs, ys], dim=2)  # [H, W, (x, y)]
    return g if mode == 'xy' else g.permute(1, 0, 2)


# --------------------------------------------------------------------------- pairwise IoU  [N] x [N] -> [N,1]
def _inter_pair(a, b):
    iw = (torch.minimum(a[:, 2], b[:, 2]) - torch.maximum(a[:, 0], b[:, 0])).clamp(0)
    ih = (torch.minimum(a[:, 3], b[:, 3]) - torch.maximum(a[:, 1], b[:, 1])).clamp(0)
    return iw * ih


def xyxy_iou(a, b, eps=EPS):
    """detection/tools/IOU.py:73-87 -- eps sits INSIDE the height factor of both areas (:74-75)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1] + eps)
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1] + eps)
    inter = _inter_pair(a, b)
    union = area_a + area_b - inter + eps
    return (inter / union).reshape(-1, 1)


def xywh_iou(a, b, eps=EPS):
    """detection/tools/IOU.py:29-40."""
    return xyxy_iou(xywh2xyxy(a), xywh2xyxy(b), eps)


def wh_iou(a, b, eps=EPS):
    """detection/tools/IOU.py:108-120."""
    inter = torch.minimum(a[:, 0], b[:, 0]) * torch.minimum(a[:, 1], b[:, 1])
    union = a[:, 0] * a[:, 1] + b[:, 0] * b[:, 1] - inter + eps
    return (inter / union).reshape(-1, 1)


def cal_iou(a, b, mode='xyxy', eps=EPS):
    """detection/tools/IOU.py:7-15."""
    if mode == 'xyxy':
        return xyxy_iou(a, b, eps)
    if mode == 'xywh':
        return xywh_iou(a, b, eps)
    if mode == 'wh':
        return wh_iou(a, b, eps)
    raise Exception('mode must be xyxy or xywh or wh')


# --------------------------------------------------------------------------- batched IoU  [N] x [M] -> [N,M]
def xyxy_iou_batch(a, b, eps=EPS):
    """detection/tools/IOU.py:142-153 -- no eps-in-height quirk here."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = (torch.minimum(a[:, None, 2], b[:, 2]) - torch.maximum(a[:, None, 0], b[:, 0])).clamp(0)
    ih = (torch.minimum(a[:, None, 3], b[:, 3]) - torch.maximum(a[:, None, 1], b[:, 1])).clamp(0)
    inter = iw * ih
    return inter / (area_a[:, None] + area_b - inter + eps)


def xywh_iou_batch(a, b, eps=EPS):
    """detection/tools/IOU.py:42-53."""
    return xyxy_iou_batch(xywh2xyxy(a), xywh2xyxy(b), eps)


def wh_iou_batch(a, b, eps=EPS):
    """detection/tools/IOU.py:177-189."""
    inter = torch.minimum(a[:, None, 0], b[:, 0]) * torch.minimum(a[:, None, 1], b[:, 1])
    return inter / ((a[:, 0] * a[:, 1])[:, None] + b[:, 0] * b[:, 1] - inter + eps)


def cal_iou_batch(a, b, mode='xyxy', eps=EPS):
    """detection/tools/IOU.py:17-25."""
    if mode == 'xyxy':
        return xyxy_iou_batch(a, b, eps)
    if mode == 'xywh':
        return xywh_iou_batch(a, b, eps)
    if mode == 'wh':
        return wh_iou_batch(a, b, eps)
    raise Exception('mode must be xyxy or xywh or wh')


# --------------------------------------------------------------------------- G/D/C-IoU (pairwise)
def _convex_wh(a, b):
    cw = torch.maximum(a[:, 2], b[:, 2]) - torch.minimum(a[:, 0], b[:, 0])
    ch = torch.maximum(a[:, 3], b[:, 3]) - torch.minimum(a[:, 1], b[:, 1])
    return cw, ch


def GIOU(a, b, mode='xyxy', eps=EPS):
    """detection/tools/IOU.py:204-239 (torch branch): plain areas (no quirk), minus sign; returns [N]."""
    if mode == 'xywh':
        a, b = xywh2xyxy(a), xywh2xyxy(b)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    inter = _inter_pair(a, b)
    union = area_a + area_b - inter + eps
    cw, ch = _convex_wh(a, b)
    convex = cw * ch + eps
    return inter / union - (convex - union) / convex


def DIOU(a, b, mode='xyxy', eps=EPS, demo=False):
    """detection/tools/IOU.py:294-343: returns iou **+** rho^2/c^2 (sign as written, :341).

    demo=True follows demos/yolov3_u/utils/iou.py:334-341 instead: centre coordinates are
    corner *sums* (not halved) and the sign is minus.
    """
    if mode == 'xywh':
        a, b = xywh2xyxy(a), xywh2xyxy(b)
    iou = xyxy_iou(a, b, eps)
    cw, ch = _convex_wh(a, b)
    c2 = cw ** 2 + ch ** 2 + eps
    cxa, cya, cxb, cyb = a[:, 0] + a[:, 2], a[:, 1] + a[:, 3], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    if not demo:
        cxa, cya, cxb, cyb = cxa * 0.5, cya * 0.5, cxb * 0.5, cyb * 0.5
    rho2 = (cxa - cxb) ** 2 + (cya - cyb) ** 2
    term = rho2.view(-1, 1) / c2.view(-1, 1)
    return iou - term if demo else iou + term


def CIOU(a, b, mode='xyxy', eps=EPS, demo=False):
    """detection/tools/IOU.py:397-440: diou - alpha*v, alpha under no_grad (:436-437)."""
    if mode == 'xywh':
        a, b = xywh2xyxy(a), xywh2xyxy(b)
    iou = xyxy_iou(a, b, eps)
    diou = DIOU(a, b, 'xyxy', eps, demo=demo)
    wa, ha = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    wb, hb = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(wb / (hb + eps)) - torch.atan(wa / (ha + eps)), 2)
    v = v.view(-1, 1)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return diou - alpha * v
