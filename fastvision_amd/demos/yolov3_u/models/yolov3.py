"""Demo YOLOv3 -- API mirror of the reference's demos/yolov3_u/models/yolov3.py (NeckV3, HeadV3, YoloV3).

state_dict keys ``neck.neck_small.0.conv.weight`` ... ``head.head_out_small.bias``; concat order is
[backbone, upsampled] (yolov3.py:96,100); the head returns raw NCHW-shaped [B,255,g,g] tensors, which here are
zero-copy views of the fp32 [B,g,g,255] buffer the head kernel writes.
"""
import torch
import torch.nn as nn

from .... import ops
from ....parallel import refuse_dataparallel_replica
from .darknet import ConvBlock1x1, ConvBlock3x3, darknet53

__all__ = ['NeckV3', 'HeadV3', 'YoloV3']


def _five(cin, mid):
    return nn.Sequential(ConvBlock1x1(in_channels=cin, out_channels=mid), ConvBlock3x3(in_channels=mid, out_channels=mid * 2),
                         ConvBlock1x1(in_channels=mid * 2, out_channels=mid), ConvBlock3x3(in_channels=mid, out_channels=mid * 2),
                         ConvBlock1x1(in_channels=mid * 2, out_channels=mid))


class NeckV3(nn.Module):
    def __init__(self, in_channels_small, in_channels_medium, in_channels_large):
        super().__init__()
        s, m, l = in_channels_small, in_channels_medium, in_channels_large
        self.neck_small = _five(s, s // 2)
        self.neck_out_small = ConvBlock3x3(in_channels=s // 2, out_channels=s)
        self.up_sampling_small = nn.Sequential(ConvBlock1x1(in_channels=s // 2, out_channels=s // 4), nn.Upsample(None, 2, 'nearest'))
        self.neck_medium = _five(m + s // 4, m // 2)
        self.neck_out_medium = ConvBlock3x3(in_channels=m // 2, out_channels=m)
        self.up_sampling_medium = nn.Sequential(ConvBlock1x1(in_channels=m // 2, out_channels=m // 4), nn.Upsample(None, 2, 'nearest'))
        self.neck_large = _five(l + m // 4, l // 2)
        self.neck_out_large = ConvBlock3x3(in_channels=l // 2, out_channels=l)

    def forward(self, x_small, x_medium, x_large):
        with ops.defer_apply_scope():        # chains of this package's blocks: a 3x3 block's apply pass may ride in the next 1x1 launch
            neck_small = self.neck_small(x_small)
        neck_out_small = self.neck_out_small(neck_small)
        cat_m = ops.upsample2_concat(self.up_sampling_small[0](neck_small), x_medium, up_first=False)
        with ops.defer_apply_scope():
            neck_medium = self.neck_medium(cat_m)
        neck_out_medium = self.neck_out_medium(neck_medium)
        cat_l = ops.upsample2_concat(self.up_sampling_medium[0](neck_medium), x_large, up_first=False)
        with ops.defer_apply_scope():
            neck_large = self.neck_large(cat_l)
        return neck_out_small, neck_out_medium, self.neck_out_large(neck_large)


class HeadV3(nn.Module):
    def __init__(self, in_channels_small, in_channels_medium, in_channels_large, anchors, num_classes):
        super().__init__()
        self.anchors_small, self.anchors_medium, self.anchors_large = anchors[0], anchors[1], anchors[2]
        per = 5 + num_classes                    # creation order large -> medium -> small (yolov3.py:119-123)
        self.head_out_large = nn.Conv2d(in_channels_large, self.anchors_large.size(0) * per, (1, 1), stride=(1, 1), padding=(0, 0), bias=True)
        self.head_out_medium = nn.Conv2d(in_channels_medium, self.anchors_medium.size(0) * per, (1, 1), stride=(1, 1), padding=(0, 0), bias=True)
        self.head_out_small = nn.Conv2d(in_channels_small, self.anchors_small.size(0) * per, (1, 1), stride=(1, 1), padding=(0, 0), bias=True)

    def forward(self, x_small, x_medium, x_large):
        nchw = lambda t: t.permute(0, 3, 1, 2)       # [B,H,W,N] buffer -> logical [B,N,H,W]
        return (nchw(ops.head_conv(x_small, self.head_out_small)), nchw(ops.head_conv(x_medium, self.head_out_medium)),
                nchw(ops.head_conv(x_large, self.head_out_large)))


class YoloV3(nn.Module):
    def __init__(self, in_channels=3, num_classes=80, anchors=(), backbone_weights=None):
        super().__init__()
        self.anchors = anchors
        self.backbone = darknet53(in_channels=in_channels, num_classes=num_classes, including_top=False)
        if backbone_weights:
            pretrained = torch.load(backbone_weights)
            self.backbone.load_state_dict({k[7:]: v for k, v in pretrained.items()}, False)   # strips 'module.' (yolov3.py:153-159)
        self.neck = NeckV3(1024, 512, 256)
        self.head = HeadV3(1024, 512, 256, anchors, num_classes)

    def forward(self, x):
        refuse_dataparallel_replica(self)        # the demo wraps its model in nn.DataParallel (demos/yolov3_u/train.py:85): one visible device only
        return self.head(*self.neck(*self.backbone(x)))
