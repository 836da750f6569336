"""YOLOv3 FPN neck on the MI355X kernels -- API mirror of the reference's detection/neck/yolov3neck.py.

Keys: ``neck1.conv1.conv.weight`` ... ``up1.squeeze.bn.weight`` ... ``conv3.bn.running_var``; concat order is
[upsampled, backbone] (yolov3neck.py:105,110).  Upsample x2 + concat is one HIP kernel writing the halo buffer
the next 1x1 conv reads.
"""
import torch.nn as nn

from ... import ops
from ...classfication.models.darknet53 import ConvBlock1x1, ConvBlock3x3

__all__ = ['YoloBlock', 'UpSampling', 'Yolov3Neck', 'yolov3neck']


class YoloBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = ConvBlock1x1(in_channels=in_channels, out_channels=out_channels)
        self.conv2 = ConvBlock3x3(in_channels=out_channels, out_channels=out_channels * 2)
        self.conv3 = ConvBlock1x1(in_channels=out_channels * 2, out_channels=out_channels)
        self.conv4 = ConvBlock3x3(in_channels=out_channels, out_channels=out_channels * 2)
        self.conv5 = ConvBlock1x1(in_channels=out_channels * 2, out_channels=out_channels)

    def forward(self, x):
        # a chain of this package's blocks only: a 3x3 block's apply pass may ride in the 1x1 launch that follows (ops.defer_apply_scope)
        with ops.defer_apply_scope():
            return self.conv5(self.conv4(self.conv3(self.conv2(self.conv1(x)))))


class UpSampling(nn.Module):
    """1x1 squeeze then nearest x2 (yolov3neck.py:66-74).  Called alone it returns the upsampled map; the neck
    calls ``squeeze`` and fuses the upsample into the concat kernel instead."""

    def __init__(self, in_channels, out_channels, scale_factor=2):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError('fastvision_amd UpSampling supports scale_factor=2')
        self.squeeze = ConvBlock1x1(in_channels=in_channels, out_channels=out_channels)
        self.upsampling = nn.Upsample(scale_factor=scale_factor, mode='nearest')

    def forward(self, x):
        return self.upsampling(self.squeeze(x))


class Yolov3Neck(nn.Module):
    def __init__(self, feature_channels):
        super().__init__()
        c0, c1, c2 = feature_channels
        self.neck1 = YoloBlock(in_channels=c0, out_channels=c0 // 2)
        self.conv1 = ConvBlock3x3(in_channels=c0 // 2, out_channels=c0)
        self.up1 = UpSampling(in_channels=c0 // 2, out_channels=c0 // 4, scale_factor=2)
        self.neck2 = YoloBlock(in_channels=c1 + c0 // 4, out_channels=c1 // 2)
        self.conv2 = ConvBlock3x3(in_channels=c1 // 2, out_channels=c1)
        self.up2 = UpSampling(in_channels=c1 // 2, out_channels=c1 // 4, scale_factor=2)
        self.neck3 = YoloBlock(in_channels=c2 + c1 // 4, out_channels=c2 // 2)
        self.conv3 = ConvBlock3x3(in_channels=c2 // 2, out_channels=c2)

    def forward(self, features: list):
        small, middle, large = features
        s = self.neck1(small)
        middle_cat = ops.upsample2_concat(self.up1.squeeze(s), middle, up_first=True)
        small_to_head = self.conv1(s)
        m = self.neck2(middle_cat)
        large_cat = ops.upsample2_concat(self.up2.squeeze(m), large, up_first=True)
        middle_to_head = self.conv2(m)
        l = self.neck3(large_cat)
        return [small_to_head, middle_to_head, self.conv3(l)]


def yolov3neck(feature_channels):
    return Yolov3Neck(feature_channels)
