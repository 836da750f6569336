"""CPU restatement of the RPN proposal layer (scope row f-4): demos/faster_rcnn/models/rpn.py:110-186 (`dxdydwdh2xywh`,
`xywh2xyxy`, `make_anchors_xywh`, `filter_proposals`).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parity status: everything up to and including the per-image top-k is PINNED by tests/golden/rpn_proposals.npz, captured from
the reference's own RPN class run in the build container (oracle/make_golden.py rpn); `torchvision.ops.nms` underneath it is
absent from this image -> the golden run used oracle.detect.nms as its stand-in (parity unpinned for the suppression itself,
as for row f-2).  Quirk restated on purpose: both extents are decoded with exp(d[..., 2]) (rpn.py:118-119).
"""
import numpy as np
import torch

from .detect import nms


def make_anchors_xywh(base_anchors_wh, height, width):
    """rpn.py:147-160: anchor centres are the integer cell coordinates (x, y); wh = base anchors / stride. -> [1,H,W,A,4]"""
    a = torch.as_tensor(base_anchors_wh, dtype=torch.float32)
    A = a.shape[0]
    ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing='ij')
    xy = torch.stack([xs, ys], dim=-1).float().view(1, height, width, 1, 2).expand(1, height, width, A, 2)
    wh = a.view(1, 1, 1, A, 2).expand(1, height, width, A, 2)
    return torch.cat([xy, wh], dim=4)


def proposal_rows(cls, dxdydwdh, anchor_xywh, height, width):
    """rpn.py:162-180: [B, H*W*A, 5] = score, clamped x1, y1, x2, y2"""
    xywh = dxdydwdh.clone()
    xywh[..., 0] = dxdydwdh[..., 0] * anchor_xywh[..., 2] + anchor_xywh[..., 0]
    xywh[..., 1] = dxdydwdh[..., 1] * anchor_xywh[..., 3] + anchor_xywh[..., 1]
    xywh[..., 2] = torch.exp(dxdydwdh[..., 2]) * anchor_xywh[..., 2]
    xywh[..., 3] = torch.exp(dxdydwdh[..., 2]) * anchor_xywh[..., 3]
    score = torch.softmax(cls, dim=4)[..., 1]
    rows = torch.cat([score[..., None], xywh], dim=4).view(cls.size(0), -1, 5)
    x, y, w, h = rows[..., 1].clone(), rows[..., 2].clone(), rows[..., 3].clone(), rows[..., 4].clone()
    rows[..., 1] = (x - w / 2).clamp(min=0, max=width - 1)
    rows[..., 2] = (y - h / 2).clamp(min=0, max=height - 1)
    rows[..., 3] = (x + w / 2).clamp(min=0, max=width - 1)
    rows[..., 4] = (y + h / 2).clamp(min=0, max=height - 1)
    return rows


def filter_proposals(cls, dxdydwdh, base_anchors_wh, pre_nms_top_n=2000, post_nms_top_n=2000, nms_thresh=0.7):
    """rpn.py:162-209 -> list over images of [n, 4] xywh (feature-map cells), best score first"""
    B, H, W = cls.shape[:3]
    rows = proposal_rows(cls.float(), dxdydwdh.float(), make_anchors_xywh(base_anchors_wh, H, W), H, W)
    out = []
    for b in range(B):
        p = rows[b]
        _, idx = p[:, 0].topk(min(pre_nms_top_n, p.size(0)))
        p = p[idx]
        keep = nms(p[:, 1:], p[:, 0], nms_thresh)[:post_nms_top_n]
        p = p[keep]
        xyxy = p[:, 1:]
        out.append(torch.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0],
                                xyxy[:, 3] - xyxy[:, 1]], dim=1))
    return out


# ------------------------------------------------------------------------------------------------ matcher (rpn.py:209-290)
def batch_iou(xywh1, xywh2, eps=1e-7):
    """rpn.py:209-226: [N,4] x [M,4] (xywh) -> [N,M]"""
    def xyxy(b):
        return torch.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1)
    a, b = xyxy(xywh1), xyxy(xywh2)
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    inter = (torch.minimum(a[:, None, 2], b[:, 2]) - torch.maximum(a[:, None, 0], b[:, 0])).clamp(0) * \
            (torch.minimum(a[:, None, 3], b[:, 3]) - torch.maximum(a[:, None, 1], b[:, 1])).clamp(0)
    union = area1[:, None] + area2 - inter + eps
    return inter / union


def rpn_match(anchor_xywh, targets, batch, feature_height, feature_width, pos_thr=0.7, neg_thr=0.3):
    """The labelling inside computet_loss (rpn.py:255-277): [B, Na] int64 -- >= 0: index of the matched box among the image's
    boxes, -1 negative, -2 ignored.  (An image without boxes makes the reference raise; here it is all -2.)"""
    anchors = anchor_xywh.reshape(-1, 4).float()
    out = torch.full((batch, anchors.size(0)), -2, dtype=torch.int64)
    scale = torch.tensor([feature_width, feature_height, feature_width, feature_height])
    for b in range(batch):
        tg = targets[targets[:, 0] == b]
        if tg.size(0) == 0:
            continue
        iou = batch_iou(anchors, tg[:, 2:] * scale)
        best, idx = torch.max(iou, dim=1)
        lab = out[b]
        m = best > pos_thr
        lab[m] = idx[m]
        lab[best < neg_thr] = -1
        _, best_anchor = torch.max(iou, dim=0)
        for t in range(best_anchor.size(0)):
            lab[best_anchor[t]] = t
    return out


def rpn_sample(labels_image, positives_per_image=128, negatives_per_image=128, perm_pos=None, perm_neg=None):
    """rpn.py:279-290 for one image: indices of the sampled positive and negative anchors.  perm_*: the permutations the
    reference draws with torch.randperm (injectable so that the draw can be shared with the path under test)."""
    pos = torch.nonzero(labels_image >= 0).flatten()
    neg = torch.nonzero(labels_image == -1).flatten()
    n_pos = min(pos.numel(), positives_per_image)
    n_neg = min(neg.numel(), max(negatives_per_image, positives_per_image + negatives_per_image - n_pos))
    perm_pos = torch.randperm(pos.numel()) if perm_pos is None else perm_pos
    perm_neg = torch.randperm(neg.numel()) if perm_neg is None else perm_neg
    return pos[perm_pos[:n_pos]], neg[perm_neg[:n_neg]]


def xywh2dxdydwdh(target_xywh, anchor_xywh, eps=1e-7):
    """rpn.py:123-131: regression targets of matched (box, anchor) pairs"""
    return torch.stack([(target_xywh[:, 0] - anchor_xywh[:, 0]) / anchor_xywh[:, 2], (target_xywh[:, 1] - anchor_xywh[:, 1]) / anchor_xywh[:, 3],
                        torch.log(target_xywh[:, 2] / anchor_xywh[:, 2] + eps), torch.log(target_xywh[:, 3] / anchor_xywh[:, 3] + eps)], 1)


# ------------------------------------------------------------------------------------------------ Fast head samples (fast.py:100-166)
def fast_match(proposals_xywh, image_boxes_xywh, pos_thr=0.5, neg_thr=0.5, neg_floor=0.1):
    """fast.py:113-127 for one image: [N] int64 -- >= 0 matched box (best IoU >= pos_thr), -1 negative (neg_floor <= best <
    neg_thr), -2 ignored"""
    lab = torch.full((proposals_xywh.size(0),), -2, dtype=torch.int64)
    if proposals_xywh.size(0) == 0 or image_boxes_xywh.size(0) == 0:
        return lab
    best, idx = torch.max(batch_iou(proposals_xywh, image_boxes_xywh), dim=1)
    m = best >= pos_thr
    lab[m] = idx[m]
    lab[(best < neg_thr) & (best >= neg_floor)] = -1
    return lab


def fast_select_samples(proposals, targets, pos_thr=0.5, neg_thr=0.5, positives_per_image=16, negatives_per_image=48, perms=None):
    """select_positive_negative_samples (fast.py:100-166): proposals = list over images of [n, 4] xywh, targets [T, 6] with xywh in
    feature cells -> (positives [P, 10] = image, proposal xywh, regression target, class; negatives [Q, 5] = image, proposal
    xywh).  perms: per image (perm_pos, perm_neg) instead of torch.randperm."""
    all_pos, all_neg = [], []
    for b, prop in enumerate(proposals):
        tg = targets[targets[:, 0] == b]
        lab = fast_match(prop, tg[:, 2:], pos_thr, neg_thr)
        pos, neg = rpn_sample(lab, positives_per_image, negatives_per_image, *(perms[b] if perms is not None else (None, None)))
        boxes = tg[:, 2:][lab[pos]]
        all_pos.append(torch.cat([torch.full((pos.numel(), 1), float(b)), prop[pos], xywh2dxdydwdh(boxes, prop[pos]).view(-1, 4),
                                  tg[:, 1:2][lab[pos]]], 1))
        all_neg.append(torch.cat([torch.full((neg.numel(), 1), float(b)), prop[neg]], 1))
    return torch.cat(all_pos, 0), torch.cat(all_neg, 0)
