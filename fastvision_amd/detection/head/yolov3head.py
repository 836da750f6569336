"""YOLOv3 detection head on the MI355X kernels -- API mirror of the reference's detection/head/yolov3head.py.

Keys ``heads.{0,1,2}.weight/bias``.  The biased 1x1 conv writes fp32 [B,H,W,A*(5+C)] once; the reference's
``view(B,A,5+C,H,W).permute(0,1,3,4,2).contiguous()`` (yolov3head.py:63) is returned as a zero-copy strided view
of that buffer with the same shape [B,A,H,W,5+C] and the same values.
"""
import torch.nn as nn

from ... import ops
from ...classfication.models.darknet53 import conv1x1

__all__ = ['Yolov3Head', 'yolov3head']


class Yolov3Head(nn.Module):
    def __init__(self, feature_channels, num_levels, num_anchors_per_level, num_classes):
        super().__init__()
        self.num_levels = num_levels
        self.num_anchors_per_level = num_anchors_per_level
        self.out_channels = num_classes + 5
        self.heads = nn.ModuleList(conv1x1(in_channels=c, out_channels=self.out_channels * a, kernel_size=(1, 1), bias=True)
                                   for c, a in zip(feature_channels, num_anchors_per_level))

    def forward(self, features: list):
        for i in range(self.num_levels):
            out = ops.head_conv(features[i], self.heads[i])                      # [B,H,W,A*K] fp32
            b, h, w, _ = out.shape
            features[i] = out.view(b, h, w, self.num_anchors_per_level[i], self.out_channels).permute(0, 3, 1, 2, 4)
        return features


def yolov3head(feature_channels, num_levels, num_anchors_per_level, num_classes):
    return Yolov3Head(feature_channels, num_levels, num_anchors_per_level, num_classes)
