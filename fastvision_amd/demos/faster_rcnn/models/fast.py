"""Second stage of the reference's Faster R-CNN demo on the HIP ops -- API mirror of demos/faster_rcnn/models/fast.py (class Fast;
same constructor arguments, parameter names and return values).

    training:  proposals + boxes -> sampled positives / negatives (rpn_ops.fast_select_samples: HIP matcher + the reference's
               sampling rule) -> RoIAlign 7x7 (roi_ops: HIP forward / backward) -> the backbone's two-layer classifier ->
               class logits and box regression -> cross-entropy over positives + negatives, smooth-L1 on the positives'
               regression against targets normalised by std (0.1, 0.1, 0.2, 0.2)
    inference: per image RoIAlign of its proposals, heads, decode, arg-max class, background dropped -> [n, 6] = xywh, class, score

The fully connected layers (25088 -> 4096 -> 4096 -> classes / boxes) run on the library's implicit-GEMM kernels as 1x1 convolutions
over the RoI rows (fc_ops.linear_relu: bias + ReLU epilogue; fc_ops.linear: the biased head convolution), cross-entropy and smooth-L1
with their gradients in one launch each (fva_row_loss / fva_smooth_l1).  ``perms``: optional per-image (perm_pos, perm_neg) instead
of ``torch.randperm``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....fc_ops import cross_entropy_mean, linear, linear_relu, smooth_l1_mean
from ....roi_ops import roi_align
from ....rpn_ops import fast_select_samples

__all__ = ['Fast']

BOX_STD = (0.1, 0.1, 0.2, 0.2)


def _xyxy(xywh):
    return torch.cat([xywh[..., :2] - xywh[..., 2:4] / 2, xywh[..., :2] + xywh[..., 2:4] / 2], dim=-1)


class Fast(nn.Module):
    def __init__(self, training=False, fast_multi_reg_head=False, module_after_roi=None, in_channels=512, num_classes=80,
                 fast_positive_iou_thres=0.5, fast_negative_iou_thres=0.5, fast_positives_per_image=16, fast_negatives_per_image=48,
                 fast_roi_pool=7):
        super().__init__()
        self.training = training
        self.fast_multi_reg_head = fast_multi_reg_head
        self.fast_positive_iou_thres, self.fast_negative_iou_thres = fast_positive_iou_thres, fast_negative_iou_thres
        self.fast_positives_per_image, self.fast_negatives_per_image = fast_positives_per_image, fast_negatives_per_image
        self.fast_roi_pool = fast_roi_pool
        self.module_after_roi = module_after_roi
        self.classifier = nn.Linear(4096, num_classes + 1)
        self.regressor = nn.Linear(4096, (num_classes + 1) * 4 if fast_multi_reg_head else 4)

    def _heads(self, feature_backbone, rois_xyxy):
        """rois [K, 5] = image, x1, y1, x2, y2 -> (class logits [K, classes + 1], box regression [K, 4 or (classes + 1) * 4])"""
        pooled = roi_align(feature_backbone, rois_xyxy, output_size=(self.fast_roi_pool, self.fast_roi_pool))
        hidden = torch.flatten(pooled, 1)
        if hidden.size(0) == 0:
            return hidden.new_zeros((0, self.classifier.out_features)), hidden.new_zeros((0, self.regressor.out_features))
        layers = list(self.module_after_roi)
        i = 0
        while i < len(layers):                       # Linear + ReLU pairs of the VGG classifier fuse into one launch; Dropout stays
            layer = layers[i]
            if isinstance(layer, nn.Linear) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
                hidden = linear_relu(hidden, layer)
                i += 2
            elif isinstance(layer, nn.Linear):
                hidden = linear(hidden, layer)
                i += 1
            else:
                hidden = layer(hidden)
                i += 1
        return linear(hidden, self.classifier), linear(hidden, self.regressor)

    @staticmethod
    def _pick_class_box(box, cls_idx):
        return box.view(box.size(0), -1, 4)[torch.arange(box.size(0), device=box.device), cls_idx.view(-1).long()]

    def compute_loss(self, positive_cls, negative_cls, positive_box, target_txtytwth, target_cls):
        if positive_cls.size(0) == 0:
            return torch.zeros(1).to(positive_cls), torch.zeros(1).to(positive_box)
        std = torch.tensor(BOX_STD).to(target_txtytwth)
        loss_box = smooth_l1_mean(positive_box, target_txtytwth / std)
        logits = torch.cat([positive_cls, negative_cls], dim=0)
        labels = torch.cat([target_cls.view(-1) + 1, torch.zeros(negative_cls.size(0)).to(target_cls)], dim=0).long()
        return cross_entropy_mean(logits, labels), loss_box

    def forward(self, feature_backbone, proposals, targets=None, perms=None):
        bs, c, h, w = feature_backbone.shape
        device = feature_backbone.device
        if self.training:
            targets[..., 2:] = targets[..., 2:] * torch.tensor([w, h, w, h]).to(device=device)      # in place, like the reference
            positives, negatives = fast_select_samples(proposals, targets, self.fast_positive_iou_thres, self.fast_negative_iou_thres,
                                                       self.fast_positives_per_image, self.fast_negatives_per_image, perms=perms)
            pos_rois = torch.cat([positives[:, :1], _xyxy(positives[:, 1:5])], 1)
            neg_rois = torch.cat([negatives[:, :1], _xyxy(negatives[:, 1:5])], 1)
            # the reference runs the head twice (fast.py:227-240: positives, then negatives); rows are independent, so ONE pass over
            # both sets gives the same logits and streams the 205 MB of classifier weights once
            n_pos = pos_rois.size(0)
            all_cls, all_box = self._heads(feature_backbone, torch.cat([pos_rois, neg_rois], 0))
            positive_cls, positive_box, negative_cls = all_cls[:n_pos], all_box[:n_pos], all_cls[n_pos:]
            if self.fast_multi_reg_head:
                positive_box = self._pick_class_box(positive_box, positives[:, 9] + 1)
            return self.compute_loss(positive_cls, negative_cls, positive_box, positives[:, 5:9], positives[:, 9:10])
        predicts = []
        std = torch.tensor(BOX_STD, device=device)
        for b in range(bs):
            xywh = proposals[b]
            rois = torch.cat([torch.full((xywh.size(0), 1), float(b), device=device), _xyxy(xywh)], 1)
            cls, box = self._heads(feature_backbone, rois)
            if self.fast_multi_reg_head:
                box = self._pick_class_box(box, cls.argmax(dim=1))
            box = box * std
            decoded = torch.stack([box[:, 0] * xywh[:, 2] + xywh[:, 0], box[:, 1] * xywh[:, 3] + xywh[:, 1],
                                   torch.exp(box[:, 2]) * xywh[:, 2], torch.exp(box[:, 2]) * xywh[:, 3]], 1)   # exp(d[2]) twice: fast.py:96-97
            scores, categories = torch.softmax(cls, dim=1).max(dim=1)
            keep = categories > 0
            predicts.append(torch.cat([decoded[keep], (categories[keep, None] - 1).to(decoded), scores[keep, None]], dim=1))
        return predicts
