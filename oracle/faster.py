"""CPU restatement of the reference's Faster R-CNN training forward (scope row f-4): demos/faster_rcnn/models/faster.py:94-104 and
what it calls -- vgg.py:49-72 (backbone), rpn.py:318-345 (RPN forward), rpn.py:227-316 (its loss), fast.py:209-247 (Fast head,
training branch), fast.py:173-201 (its loss).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch fp32 ops on the CPU over the PARAMETERS of a model object (any
module tree with the reference's parameter names: the product's mirror classes hold exactly those); none of the product's compute
is used.  Pinned by tests/golden/faster_step.npz (the reference's own model run in the build container): tests/
test_oracle_faster_golden.py.  torchvision's nms / roi_align are the restatements of oracle/detect.py and oracle/roi_align.py
(parity unpinned for those two, as everywhere).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import roi_align as RA
from . import rpn as R


class _RoiAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, boxes, size):
        ctx.boxes, ctx.shape = boxes.detach().numpy().copy(), tuple(feat.shape)
        return torch.from_numpy(RA.roi_align(feat.detach().numpy(), ctx.boxes, (size, size)))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(RA.roi_align_backward(g.contiguous().numpy(), ctx.boxes, ctx.shape)), None, None


def backbone_features(backbone, images):
    """vgg.py:49-72: five stages of conv + ReLU, a 2x2 max-pool after each of the first four"""
    x = images
    for i, stage in enumerate((backbone.vgg1, backbone.vgg2, backbone.vgg3, backbone.vgg4, backbone.vgg5)):
        for layer in stage:
            if isinstance(layer, torch.nn.Conv2d):
                x = F.relu(F.conv2d(x, layer.weight, layer.bias, 1, 1))
        if i < 4:
            x = F.max_pool2d(x, 2, 2)
    return x


def focal(logits, labels, gamma=2):
    p = torch.softmax(logits, dim=1).gather(1, labels.view(-1, 1))
    return (-torch.pow(1 - p, gamma) * p.log()).mean()


def training_losses(model, images, targets, perms):
    """-> (proposals, loss_rpn_cls, loss_rpn_box, loss_fast_cls, loss_fast_box).  perms: 2B (perm_pos, perm_neg) pairs, the RPN's
    per image first, then the Fast head's."""
    B = images.size(0)
    rpn, fast = model.rpn, model.fast
    feature = backbone_features(model.backbone, images)
    _, _, h, w = feature.shape
    x = F.relu(F.conv2d(feature, rpn.conv3x3.weight, rpn.conv3x3.bias, 1, 1))
    cls = F.conv2d(x, rpn.classifier.weight, rpn.classifier.bias).permute(0, 2, 3, 1).reshape(B, h, w, -1, 2)
    d = F.conv2d(x, rpn.regressor.weight, rpn.regressor.bias).permute(0, 2, 3, 1).reshape(B, h, w, -1, 4)
    base = rpn.base_anchors.float()
    anchors = R.make_anchors_xywh(base, h, w).view(-1, 4)
    proposals = R.filter_proposals(cls.detach(), d.detach(), base, rpn.rpn_pre_nms_top_n, rpn.rpn_post_nms_top_n, rpn.rpn_nms_thresh)
    labels = R.rpn_match(anchors, targets, B, h, w, rpn.rpn_positive_iou_thres, rpn.rpn_negative_iou_thres)
    scale = torch.tensor([w, h, w, h], dtype=torch.float32)
    cls_rows, cls_tg, box_rows, box_tg = [], [], [], []
    for b in range(B):
        pos, neg = R.rpn_sample(labels[b], rpn.rpn_positives_per_image, rpn.rpn_negatives_per_image, *perms[b])
        cls_rows.append(torch.cat([cls[b].reshape(-1, 2)[neg], cls[b].reshape(-1, 2)[pos]], 0))
        cls_tg.append(torch.cat([torch.zeros_like(neg), torch.ones_like(pos)], 0))
        boxes = (targets[targets[:, 0] == b][:, 2:] * scale)[labels[b][pos]]
        box_rows.append(d[b].reshape(-1, 4)[pos])
        box_tg.append(R.xywh2dxdydwdh(boxes, anchors[pos]))
    loss_rpn_cls = focal(torch.cat(cls_rows, 0), torch.cat(cls_tg, 0))
    loss_rpn_box = F.smooth_l1_loss(torch.cat(box_rows, 0), torch.cat(box_tg, 0), reduction='mean')
    # ---- Fast head (fast.py:209-247)
    tg = targets.clone()
    tg[:, 2:] = tg[:, 2:] * scale
    positives, negatives = R.fast_select_samples(proposals, tg, fast.fast_positive_iou_thres, fast.fast_negative_iou_thres,
                                                 fast.fast_positives_per_image, fast.fast_negatives_per_image, perms=perms[B:])

    def heads(rois_xywh):
        rois = torch.cat([rois_xywh[:, :1], rois_xywh[:, 1:3] - rois_xywh[:, 3:5] / 2, rois_xywh[:, 1:3] + rois_xywh[:, 3:5] / 2], 1)
        hidden = torch.flatten(_RoiAlign.apply(feature, rois, fast.fast_roi_pool), 1)
        for layer in fast.module_after_roi:
            if isinstance(layer, torch.nn.Linear):
                hidden = F.relu(F.linear(hidden, layer.weight, layer.bias))          # Dropout at p = 0 is the identity
        return F.linear(hidden, fast.classifier.weight, fast.classifier.bias), F.linear(hidden, fast.regressor.weight, fast.regressor.bias)
    pos_cls, pos_box = heads(positives[:, :5])
    neg_cls, _ = heads(negatives)
    if fast.fast_multi_reg_head:
        pos_box = pos_box.view(pos_box.size(0), -1, 4)[torch.arange(pos_box.size(0)), (positives[:, 9] + 1).long()]
    std = torch.tensor((0.1, 0.1, 0.2, 0.2))
    loss_fast_box = F.smooth_l1_loss(pos_box, positives[:, 5:9] / std, reduction='mean')
    logits = torch.cat([pos_cls, neg_cls], 0)
    lab = torch.cat([positives[:, 9] + 1, torch.zeros(negatives.size(0))], 0).long()
    loss_fast_cls = F.cross_entropy(logits, lab, reduction='mean')
    return proposals, loss_rpn_cls, loss_rpn_box, loss_fast_cls, loss_fast_box
