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

__all__ = ['filter_proposals', 'rpn_proposal_rows', 'rpn_match', 'rpn_sample', 'fast_match', 'fast_select_samples']

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
    k = min(pre_nms_top_n, R)
    # Only the k best rows of an image enter NMS.  The candidate stage ranks what it is given against each other (quadratic), so
    # it is handed a score floor just below the smallest per-image k-th best score: every image keeps at least its k best rows
    # (ties included), ~k instead of all R = h*w*A rows get ranked.  (-1 = no floor when k == R.)
    floor = -1.0
    if k < R:
        kth = rows[..., 4].topk(k, dim=1).values[:, -1].min()
        floor = float(torch.nextafter(kth, kth.new_tensor(-1.0)))
    dets = nms_batch(rows, floor, nms_thresh, min(post_nms_top_n, R), dict(_MODE, max_nms=k))
    out = []
    for det, _ in dets:
        xyxy = det[:, :4]
        out.append(torch.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0],
                                xyxy[:, 3] - xyxy[:, 1]], dim=1))
    return out


def rpn_match(anchor_xywh, targets, batch, feature_height, feature_width, pos_thr=0.7, neg_thr=0.3):
    """The anchor labelling of RPN.computet_loss (rpn.py:255-277) on the device: anchor_xywh [.., 4] (make_anchors_xywh),
    targets [T, 6] = image index, class, normalised xywh -> [B, Na] int64: >= 0 index of the matched box among its image's
    boxes, -1 negative, -2 ignored."""
    require_gpu(anchor_xywh, 'rpn_match')
    anchors = anchor_xywh.detach().reshape(-1, 4).float().contiguous()
    tg = targets.detach().to(device=anchors.device, dtype=torch.float32).contiguous()
    Na, T = anchors.size(0), tg.size(0)
    labels = torch.empty((batch, Na), dtype=torch.int32, device=anchors.device)
    ws = torch.empty(max(T, 1), dtype=torch.int32, device=anchors.device)
    _lib.call('fva_rpn_match', _p(anchors), Na, _p(tg) if T else None, T, batch, int(feature_height), int(feature_width),
              float(pos_thr), float(neg_thr), _p(labels), _p(ws), _stream())
    return labels.long()


def rpn_sample(labels_image, positives_per_image=128, negatives_per_image=128, perm_pos=None, perm_neg=None):
    """rpn.py:279-290 for one image's labels: indices of the sampled positive / negative anchors (torch.randperm draws unless
    the permutations are given)."""
    pos = torch.nonzero(labels_image >= 0).flatten()
    neg = torch.nonzero(labels_image == -1).flatten()
    n_pos = min(pos.numel(), positives_per_image)
    n_neg = min(neg.numel(), max(negatives_per_image, positives_per_image + negatives_per_image - n_pos))
    given = perm_pos is not None or perm_neg is not None
    perm_pos = torch.randperm(pos.numel(), device=labels_image.device) if perm_pos is None else perm_pos
    perm_neg = torch.randperm(neg.numel(), device=labels_image.device) if perm_neg is None else perm_neg
    if given:      # a foreign permutation (parity tests share the reference's draws): its used prefix must index inside the candidate lists
        for name, perm, n, cnt in (('positive', perm_pos, n_pos, pos.numel()), ('negative', perm_neg, n_neg, neg.numel())):
            if n and (perm.numel() < n or int(perm[:n].max()) >= cnt or int(perm[:n].min()) < 0):
                raise ValueError(f'rpn_sample: the {name} permutation does not fit {cnt} candidates ({n} to draw)')
    return pos[perm_pos[:n_pos]], neg[perm_neg[:n_neg]]


def fast_match(proposals_xywh, targets, image, pos_thr=0.5, neg_thr=0.5, neg_floor=0.1):
    """Labelling of one image's proposals against its boxes (fast.py:113-127) on the device: [N] int64."""
    require_gpu(proposals_xywh, 'fast_match')
    prop = proposals_xywh.detach().float().contiguous()
    tg = targets.detach().to(device=prop.device, dtype=torch.float32).contiguous()
    N, T = prop.size(0), tg.size(0)
    labels = torch.empty(N, dtype=torch.int32, device=prop.device)
    _lib.call('fva_fast_match', _p(prop) if N else None, N, _p(tg) if T else None, T, int(image), float(pos_thr), float(neg_thr),
              float(neg_floor), _p(labels) if N else None, _stream())
    return labels.long()


def fast_select_samples(proposals, targets, pos_thr=0.5, neg_thr=0.5, positives_per_image=16, negatives_per_image=48, perms=None):
    """Fast.select_positive_negative_samples (fast.py:100-166): proposals = list over images of [n, 4] xywh (filter_proposals),
    targets [T, 6] = image, class, xywh in feature cells -> (positives [P, 10], negatives [Q, 5]) like the reference.  The
    labelling runs in the HIP matcher; gathering the few dozen sampled rows is torch indexing."""
    dev = proposals[0].device
    tg = targets.to(device=dev, dtype=torch.float32)
    all_pos, all_neg = [], []
    for b, prop in enumerate(proposals):
        lab = fast_match(prop, tg, b, pos_thr, neg_thr)
        pos, neg = rpn_sample(lab, positives_per_image, negatives_per_image, *(perms[b] if perms is not None else (None, None)))
        mine = tg[tg[:, 0] == b]
        boxes, pp = mine[:, 2:][lab[pos]], prop[pos].float()
        reg = torch.stack([(boxes[:, 0] - pp[:, 0]) / pp[:, 2], (boxes[:, 1] - pp[:, 1]) / pp[:, 3],
                           torch.log(boxes[:, 2] / pp[:, 2] + 1e-7), torch.log(boxes[:, 3] / pp[:, 3] + 1e-7)], 1)
        all_pos.append(torch.cat([torch.full((pos.numel(), 1), float(b), device=dev), pp, reg, mine[:, 1:2][lab[pos]]], 1))
        all_neg.append(torch.cat([torch.full((neg.numel(), 1), float(b), device=dev), prop[neg].float()], 1))
    return torch.cat(all_pos, 0), torch.cat(all_neg, 0)
