"""Region proposal network of the reference's Faster R-CNN demo on the HIP kernels -- API mirror of
demos/faster_rcnn/models/rpn.py (classes FocalLoss, RPN; same constructor arguments, parameter names and return values).

    feature [B, C, h, w]  --conv3x3 + ReLU-->  --1x1 classifier / regressor-->  cls [B,h,w,A,2], dxdydwdh [B,h,w,A,4]
    proposals = filter_proposals(...)                      (rpn_ops: decode kernel + the validation side's top-k / NMS)
    training: anchors labelled by rpn_ops.rpn_match (HIP matcher), sampled like rpn.py:279-290, focal loss on the sampled
    logits + smooth-L1 on the positives' regression -- a few hundred rows: torch ops on the device, autograd through the heads.

``perms``: optional per-image (perm_pos, perm_neg) to use instead of ``torch.randperm`` (parity tests share the reference's draws).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....ops import HeadFn, get_compute_dtype
from ....rpn_ops import filter_proposals, rpn_match, rpn_sample
from ....vgg_ops import conv_bias_relu

__all__ = ['FocalLoss', 'RPN']


class FocalLoss(nn.Module):
    """-alpha * (1 - p_t)^gamma * log p_t over softmax probabilities, alpha = 1 per class unless given (rpn.py:8-64)."""

    def __init__(self, class_num, alpha=None, gamma=2, size_average=True):
        super().__init__()
        self.register_buffer('alpha', torch.ones(class_num) if alpha is None else torch.as_tensor(alpha, dtype=torch.float32).flatten(), persistent=False)
        self.gamma, self.class_num, self.size_average = gamma, class_num, size_average

    def forward(self, inputs, targets):
        if inputs.is_cuda and self.size_average and bool((self.alpha == 1).all()):
            from ....fc_ops import focal_mean
            return focal_mean(inputs, targets, float(self.gamma))              # value + gradient in one launch (fva_row_loss)
        p_t = torch.softmax(inputs, dim=1).gather(1, targets.view(-1, 1))
        loss = -self.alpha.to(inputs.device)[targets.view(-1)].view(-1, 1) * torch.pow(1 - p_t, self.gamma) * p_t.log()
        return loss.mean() if self.size_average else loss.sum()


class RPN(nn.Module):
    def __init__(self, training=False, base_anchors=None, backbone_stride=16, in_channels=512, rpn_pre_nms_top_n=2000,
                 rpn_post_nms_top_n=2000, rpn_nms_thresh=0.7, rpn_positive_iou_thres=0.7, rpn_negative_iou_thres=0.3,
                 rpn_positives_per_image=128, rpn_negatives_per_image=128):
        super().__init__()
        self.training = training
        self.base_anchors = base_anchors / backbone_stride                    # [A, 2] in feature cells (a plain attribute, as in the reference)
        num_anchors = base_anchors.size(0)
        self.rpn_pre_nms_top_n, self.rpn_post_nms_top_n, self.rpn_nms_thresh = rpn_pre_nms_top_n, rpn_post_nms_top_n, rpn_nms_thresh
        self.rpn_positive_iou_thres, self.rpn_negative_iou_thres = rpn_positive_iou_thres, rpn_negative_iou_thres
        self.rpn_positives_per_image, self.rpn_negatives_per_image = rpn_positives_per_image, rpn_negatives_per_image
        self.conv3x3 = nn.Conv2d(in_channels, in_channels, (3, 3), (1, 1), (1, 1), bias=True)
        self.classifier = nn.Conv2d(in_channels, num_anchors * 2, (1, 1), bias=True)
        self.regressor = nn.Conv2d(in_channels, num_anchors * 4, (1, 1), bias=True)
        for layer in (self.conv3x3, self.classifier, self.regressor):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self.focal_loss = FocalLoss(class_num=2)

    def make_anchors_xywh(self, feature_height, feature_width, device):
        """[1, h, w, A, 4]: anchor centres are the integer cell coordinates (x, y), sizes the base anchors"""
        A = self.base_anchors.size(0)
        ys, xs = torch.meshgrid(torch.arange(feature_height, device=device), torch.arange(feature_width, device=device), indexing='ij')
        xy = torch.stack([xs, ys], -1).float().view(1, feature_height, feature_width, 1, 2).expand(1, feature_height, feature_width, A, 2)
        wh = self.base_anchors.to(device).float().view(1, 1, 1, A, 2).expand_as(xy)
        return torch.cat([xy, wh], dim=4)

    def filter_proposals(self, cls, dxdydwdh, anchor_xywh=None, feature_height=None, feature_width=None):
        return filter_proposals(cls, dxdydwdh, self.base_anchors, self.rpn_pre_nms_top_n, self.rpn_post_nms_top_n, self.rpn_nms_thresh)

    def computet_loss(self, predict_cls, predict_dxdydwdh, anchor_xywh, targets, perms=None):
        bs, fh, fw = predict_cls.shape[:3]
        anchors = anchor_xywh.reshape(-1, 4)
        labels = rpn_match(anchors, targets, bs, fh, fw, self.rpn_positive_iou_thres, self.rpn_negative_iou_thres)
        scale = torch.tensor([fw, fh, fw, fh], dtype=torch.float32, device=anchors.device)
        cls_rows, cls_tg, box_rows, box_tg = [], [], [], []
        for b in range(bs):
            pos, neg = rpn_sample(labels[b], self.rpn_positives_per_image, self.rpn_negatives_per_image, *(perms[b] if perms else (None, None)))
            logits, deltas = predict_cls[b].reshape(-1, 2), predict_dxdydwdh[b].reshape(-1, 4)
            cls_rows.append(torch.cat([logits[neg], logits[pos]], 0))
            cls_tg.append(torch.cat([torch.zeros_like(neg), torch.ones_like(pos)], 0))
            boxes = (targets[targets[:, 0] == b][:, 2:].to(anchors) * scale)[labels[b][pos]]
            a = anchors[pos]
            box_rows.append(deltas[pos])
            box_tg.append(torch.stack([(boxes[:, 0] - a[:, 0]) / a[:, 2], (boxes[:, 1] - a[:, 1]) / a[:, 3],
                                       torch.log(boxes[:, 2] / a[:, 2] + 1e-7), torch.log(boxes[:, 3] / a[:, 3] + 1e-7)], 1))
        loss_cls = self.focal_loss(torch.cat(cls_rows, 0), torch.cat(cls_tg, 0))
        from ....fc_ops import smooth_l1_mean
        loss_box = smooth_l1_mean(torch.cat(box_rows, 0), torch.cat(box_tg, 0))
        return loss_cls, loss_box

    def forward(self, feature_backbone, targets=None, perms=None):
        bs, c, h, w = feature_backbone.shape
        dtype = get_compute_dtype()
        feature_rpn = conv_bias_relu(feature_backbone, self.conv3x3, dtype)
        output_cls = HeadFn.apply(feature_rpn, self.classifier.weight, self.classifier.bias, dtype).view(bs, h, w, -1, 2)
        output_dxdydwdh = HeadFn.apply(feature_rpn, self.regressor.weight, self.regressor.bias, dtype).view(bs, h, w, -1, 4)
        anchor_xywh = self.make_anchors_xywh(h, w, feature_backbone.device)
        proposals = self.filter_proposals(output_cls, output_dxdydwdh)
        if self.training:
            loss_cls, loss_box = self.computet_loss(output_cls, output_dxdydwdh, anchor_xywh, targets, perms)
            return proposals, loss_cls, loss_box
        return proposals
