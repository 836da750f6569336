"""The reference's Faster R-CNN demo model on the HIP ops -- API mirror of demos/faster_rcnn/models/faster.py (class Faster_Rcnn:
same constructor arguments, sub-module names ``backbone`` / ``rpn`` / ``fast`` and return values).

    images [B,3,H,W] -> VGG16 (stride 16) -> RPN (proposals; training: its two losses) -> Fast head (training: its two losses;
    inference: per-image detections [n, 6] = xywh in feature cells, class, score)

``perms``: optional list of 2*B (perm_pos, perm_neg) pairs -- the RPN's per image, then the Fast head's -- instead of
``torch.randperm`` draws (parity tests share the reference's).
"""
import torch
import torch.nn as nn

from .fast import Fast
from .rpn import RPN
from .vgg import vgg16

__all__ = ['Faster_Rcnn']


class Faster_Rcnn(nn.Module):
    def __init__(self, training=False, in_channels=3, num_classes=80, base_anchors=None, backbone_stride=16, backbone_output_channels=512,
                 backbone_weights='', rpn_positive_iou_thres=0.7, rpn_negative_iou_thres=0.3, rpn_positives_per_image=128,
                 rpn_negatives_per_image=128, rpn_pre_nms_top_n=2000, rpn_post_nms_top_n=2000, rpn_nms_thresh=0.7, fast_multi_reg_head=False,
                 fast_positive_iou_thres=0.5, fast_negative_iou_thres=0.5, fast_positives_per_image=16, fast_negatives_per_image=48,
                 fast_roi_pool=7):
        super().__init__()
        self.training = training
        self.backbone = vgg16(in_channels=in_channels)
        if backbone_weights:
            have, own = torch.load(backbone_weights), self.backbone.state_dict()
            matched = {k: v for k, v in have.items() if k in own and v.size() == own[k].size()}
            self.backbone.load_state_dict(matched, strict=False)
            print('Backbone not matched keys : ', [k for k in have if k not in matched])
        self.rpn = RPN(training=training, base_anchors=base_anchors, backbone_stride=backbone_stride, in_channels=backbone_output_channels,
                       rpn_pre_nms_top_n=rpn_pre_nms_top_n, rpn_post_nms_top_n=rpn_post_nms_top_n, rpn_nms_thresh=rpn_nms_thresh,
                       rpn_positive_iou_thres=rpn_positive_iou_thres, rpn_negative_iou_thres=rpn_negative_iou_thres,
                       rpn_positives_per_image=rpn_positives_per_image, rpn_negatives_per_image=rpn_negatives_per_image)
        self.fast = Fast(training=training, fast_multi_reg_head=fast_multi_reg_head, module_after_roi=self.backbone.classifier,
                         in_channels=backbone_output_channels, num_classes=num_classes, fast_positive_iou_thres=fast_positive_iou_thres,
                         fast_negative_iou_thres=fast_negative_iou_thres, fast_positives_per_image=fast_positives_per_image,
                         fast_negatives_per_image=fast_negatives_per_image, fast_roi_pool=fast_roi_pool)

    def forward(self, images, targets=None, perms=None):
        feature_backbone = self.backbone(images)
        if self.training:
            n = images.size(0)
            proposals, loss_rpn_cls, loss_rpn_box = self.rpn(feature_backbone, targets, perms=perms[:n] if perms else None)
            loss_fast_cls, loss_fast_box = self.fast(feature_backbone, proposals, targets, perms=perms[n:] if perms else None)
            return proposals, loss_rpn_cls, loss_rpn_box, loss_fast_cls, loss_fast_box
        return self.fast(feature_backbone, self.rpn(feature_backbone))
