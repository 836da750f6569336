"""RPN proposal layer on the HIP kernels -- `RPN.filter_proposals` of the reference
(demos/faster_rcnn/models/rpn.py:162-209): decode against the anchor grid + softmax objectness + clamp (`fva_rpn_decode`), then
per image the rpn_pre_nms_top_n best rows, torchvision-style NMS and the first rpn_post_nms_top_n survivors
(`fva_nms_candidates` / `fva_nms_select`, the validation side's kernels), returned as xywh like the reference does.
"""
import ctypes as C

import torch

from . import _lib
from .detect_ops import nms_batch
from .ops import _p, _stream, require_gpu

__all__ = ['filter_proposals', 'rpn_proposal_rows']

_MODE = dict(box_mode=1, score_mode=1, rethreshold=0, class_gap=0.0)


def rpn_proposal_rows(cls, dxdydwdh, base_anchors_wh):
    """cls [B,H,W,A,2], dxdydwdh [B,H,W,A,4] (the reference's permuted views; contiguous NHWC head outputs are exactly that),
    base_anchors_wh [A,2] already divided by the stride -> [B, H*W*A, 6] = clamped xyxy, foreground score, 1."""
    require_gpu(cls, 'rpn_proposal_rows')
    B, H, W, A = cls.shape[:4]
    cls = cls.detach().float().contiguous()
    d = dxdydwdh.detach().float().contiguous()
    anchors = torch.as_tensor(base_anchors_wh, dtype=torch.float32).to(cls.device).contiguous()
    rows = torch.empty((B, H * W * A, 6), dtype=torch.float32, device=cls.device)
    _lib.call('fva_rpn_decode', _p(cls), _p(d), _p(anchors), _p(rows), B, H, W, A, _stream())
    return rows


def filter_proposals(cls, dxdydwdh, base_anchors_wh, pre_nms_top_n=2000, post_nms_top_n=2000, nms_thresh=0.7):
    """-> list over images of [n, 4] xywh in feature-map cells, best score first (rpn.py:186-209)."""
    rows = rpn_proposal_rows(cls, dxdydwdh, base_anchors_wh)
    R = rows.size(1)
    # every row is a candidate (scores are probabilities > -1); the max_nms best enter NMS, max_det survive
    dets = nms_batch(rows, -1.0, nms_thresh, min(post_nms_top_n, R), dict(_MODE, max_nms=min(pre_nms_top_n, R)))
    out = []
    for det, _ in dets:
        xyxy = det[:, :4]
        out.append(torch.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0],
                                xyxy[:, 3] - xyxy[:, 1]], dim=1))
    return out
