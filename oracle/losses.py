"""Oracle: target assignment and the two YOLOv3 training losses (CPU fp32 torch).

* library loss -- loss/yolov3_loss.py:8-124 with loss/classification_loss.py:36-65,
  loss/iou_loss.py:83-107 and datasets/common/id_2_onehot.py:4-17;
* demo loss -- demos/yolov3_u/utils/lossv3.py:7-119.

Written to be differentiated by autograd so that head gradients can be compared with the
analytic HIP backward.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from . import boxes


def one_hot(idx, num_classes):
    """datasets/common/id_2_onehot.py:10-15 (torch branch): scatter 1 into zeros, same dtype/device as idx."""
    col = idx.view(-1, 1).long()
    return torch.zeros((col.size(0), num_classes)).to(idx).scatter_(1, col, 1)


def bce_probs(p, target, reduction='mean'):
    """loss/classification_loss.py:42-65 with already_sigmoid=True.

    ``p`` are probabilities.  If the last dim is >1, ``target`` holds class indices and is one-hot
    expanded (:44-46).  'mean' divides the total by p.numel() (:63-64); 1e-8 sits inside both logs (:54).
    """
    if p.size(-1) > 1:
        t = one_hot(target, p.size(-1)).float().view(-1, 1)
    else:
        t = target.float().view(-1, 1)
    p = p.view(-1, 1)
    per = -t * torch.log(p + 1e-8) - (1 - t) * torch.log(1 - p + 1e-8)
    total = per.sum()
    return total / p.numel() if reduction == 'mean' else total


def ciou_loss(pred, target, mode='xyxy', reduction='mean'):
    """loss/iou_loss.py:88-107: mean (or sum) of 1 - CIoU, unit weights."""
    per = 1 - boxes.CIOU(pred, target, mode=mode)
    return per.mean() if reduction == 'mean' else per.sum()


# ------------------------------------------------------------------------------------------ library loss
def build_target(head_shapes, y_true, anchors_per_level, strides):
    """loss/yolov3_loss.py:75-124.

    head_shapes: per level (B, A, H, W, 5+C); y_true: [T,6] = [img, cls, xc, yc, w, h] normalised.
    Returns per level: (b[M] i64, gxy[M,2] i64, a[M] i64), cls[M] i64, xywh[M,4] f32, anchors[M,2] f32,
    rows ordered target-major / anchor-minor (boolean-mask order, :105).
    """
    locs, cats, xywhs, matched = [], [], [], []
    for shape, anc, stride in zip(head_shapes, anchors_per_level, strides):
        _, _, gh, gw, _ = shape
        anc = anc.reshape(-1, 2) / stride                                   # :88-89 feature-scale anchors
        n_anc = anc.size(0)
        scale = torch.tensor([gw, gh, gw, gh], dtype=y_true.dtype)          # :92  [W,H,W,H]
        tgt = y_true.clone()
        tgt[:, 2:] = y_true[:, 2:] * scale                                  # :94-95
        ratio = tgt[:, None, 4:] / anc                                      # :98  [T,A,2]
        keep = torch.max(ratio, 1 / ratio).max(2)[0] < 4                    # :99  [T,A]
        rows = torch.cat([tgt.unsqueeze(1).repeat(1, n_anc, 1),
                          torch.arange(n_anc).view(1, n_anc, 1).repeat(tgt.size(0), 1, 1).to(tgt)], dim=2)
        sel = rows[keep]                                                    # :105 [M,7]
        b, cls, a = sel[:, 0].long(), sel[:, 1].long(), sel[:, 6].long()    # :107-111
        xy, wh = sel[:, 2:4], sel[:, 4:6]
        gxy = torch.floor(xy).long()                                        # :113
        off = xy - gxy.float()                                              # :114 (before the clamp)
        gxy[:, 0].clamp_(0, gw - 1)                                         # :116
        gxy[:, 1].clamp_(0, gh - 1)                                         # :117
        locs.append((b, gxy, a))
        cats.append(cls)
        xywhs.append(torch.cat([off, wh], dim=1))
        matched.append(anc[a])
    return locs, cats, xywhs, matched


def yolov3_loss(y_pred, y_true, anchors_per_level, strides, ratio_box, ratio_conf, ratio_cls, parts=False):
    """loss/yolov3_loss.py:29-72.  y_pred: list of [B,A,H,W,5+C]; returns a [1] tensor.

    The IoU written into the objectness target is NOT detached (:60-61); duplicates follow
    index_put semantics (last write wins in the forward value).
    """
    locs, cats, xywhs, matched = build_target([p.shape for p in y_pred], y_true, anchors_per_level, strides)
    l_cls = torch.zeros(1).to(y_pred[0])
    l_box = torch.zeros(1).to(y_pred[0])
    l_conf = torch.zeros(1).to(y_pred[0])
    for lvl, pre in enumerate(y_pred):
        b, gxy, a = locs[lvl]
        rows = pre[b, a, gxy[:, 1], gxy[:, 0]]                               # :44
        tconf = torch.zeros_like(pre[..., 4:5])
        if b.size(0):
            l_cls = l_cls + bce_probs(rows[:, 5:].sigmoid(), cats[lvl])     # :50-52
            pxywh = torch.cat([rows[:, 0:2].sigmoid(), torch.exp(rows[:, 2:4]) * matched[lvl]], dim=1)
            l_box = l_box + ciou_loss(pxywh, xywhs[lvl], mode='xywh')       # :57-58
            tconf[b, a, gxy[:, 1], gxy[:, 0]] = boxes.cal_iou(pxywh, xywhs[lvl], mode='xywh')   # :60-61
        l_conf = l_conf + bce_probs(pre[..., 4:5].sigmoid().view(-1, 1), tconf.view(-1, 1))    # :63-64
    bs = y_pred[0].size(0)
    total = (l_box * ratio_box + l_conf * ratio_conf + l_cls * ratio_cls) * bs                  # :66-72
    if parts:
        return total, (l_box.detach(), l_conf.detach(), l_cls.detach())
    return total


# ------------------------------------------------------------------------------------------ demo loss
def demo_assign(target_all, anchor, gh, gw):
    """demos/yolov3_u/utils/lossv3.py:44-61: per-level target table.

    Returns (tgt[T,6] feature-scale, gxy[T,2] float (floor, UNclamped), off[T,2], best_a[T] i64, anc[T,2]).
    """
    tgt = target_all.clone()
    tgt[:, 2:] = tgt[:, 2:] * torch.tensor([gw, gh, gw, gh]).to(tgt)        # :45-46
    best = torch.max(boxes.wh_iou_batch(tgt[:, 4:], anchor), dim=1)[1]      # :51-52 first max on ties
    gxy = torch.floor(tgt[:, 2:4])                                          # :56
    return tgt, gxy, tgt[:, 2:4] - gxy, best, anchor[best, :]


def demo_ignore_mask(pred5, tgt, anchor):
    """lossv3.py:63-101: pred5 = [B,H,W,A,5+C] view.  Returns mask [B,H,W,A,1] in {-1,0} (positives not yet set)."""
    bsz, gh, gw, na, _ = pred5.shape
    pxy = torch.sigmoid(pred5[..., 0:2])
    pwh = torch.exp(pred5[..., 2:4]) * anchor.view(1, 1, 1, na, 2)
    cell = boxes.grid(gh, gw, mode='xy').view(1, gh, gw, 1, 2).to(pxy)
    pxywh = torch.cat([pxy + cell, pwh], dim=4)
    masks = []
    for img in range(bsz):
        t_img = tgt[tgt[:, 0] == img][:, 2:6]
        best_iou = torch.max(boxes.xywh_iou_batch(pxywh[img].reshape(-1, 4), t_img), dim=1)[0]   # :93-94
        m = torch.zeros((best_iou.size(0), 1)).to(pred5)
        m[best_iou > 0.5] = -1                                               # :97
        masks.append(m.view(1, gh, gw, na, 1))
    return torch.cat(masks, 0)


def demo_loss(predict_layers, target_all, anchors, parts=False):
    """demos/yolov3_u/utils/lossv3.py:18-119 (without its print).  predict_layers: 3 x [B,255,H,W] raw NCHW."""
    z = lambda: torch.zeros(1).to(predict_layers[0])
    l_xy, l_wh, l_cls, l_conf = z(), z(), z(), z()
    for lvl, raw in enumerate(predict_layers):
        anchor = anchors[lvl]
        na = anchor.size(0)
        bsz, _, gh, gw = raw.shape
        pred = raw.permute(0, 2, 3, 1).view(bsz, gh, gw, na, -1)             # :42
        tgt, gxy, off, best, anc = demo_assign(target_all, anchor, gh, gw)
        bi, gy, gx = tgt[:, 0].long(), gxy[:, 1].long(), gxy[:, 0].long()
        rows = pred[bi, gy, gx, best]                                        # :71
        l_xy = l_xy + F.binary_cross_entropy_with_logits(rows[:, 0:2], off)  # :71-73
        l_wh = l_wh + F.mse_loss(rows[:, 2:4], torch.log(tgt[:, 4:6] / anc + 1e-14))     # :76-78
        onehot = torch.zeros_like(rows[:, 5:])
        onehot[range(len(onehot)), tgt[:, 1].long()] = 1                     # :82-83
        l_cls = l_cls + F.binary_cross_entropy_with_logits(rows[:, 5:], onehot)
        mask = demo_ignore_mask(pred, tgt, anchor)
        mask[bi, gy, gx, best] = 1                                           # :101
        valid = mask != -1
        l_conf = l_conf + F.binary_cross_entropy_with_logits(pred[..., 4:5][valid], mask[valid])  # :104-106
    total = l_xy * 2.0 + l_wh + l_cls + l_conf                               # :111-117
    if parts:
        return total, (l_xy.detach(), l_wh.detach(), l_cls.detach(), l_conf.detach())
    return total
