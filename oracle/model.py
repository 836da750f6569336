"""Oracle: YOLOv3 (Darknet-53 + FPN neck + 3-scale head) as plain torch.nn on CPU.

Two API surfaces are restated, with the reference's state_dict keys and its
parameter *construction order* (so ``torch.manual_seed(s)`` reproduces the
reference's random init bit-for-bit; checked in tests/test_oracle_golden.py):

* library surface -- classfication/models/darknet53.py:65-141,
  detection/neck/yolov3neck.py:46-118, detection/head/yolov3head.py:42-70,
  detection/models/yolov3.py:6-69;
* demo surface -- demos/yolov3_u/models/darknet.py, demos/yolov3_u/models/yolov3.py:43-175.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

STAGE_BLOCKS = (1, 2, 8, 8, 4)           # darknet53.py:141
LEVEL_STRIDES = [32, 16, 8]              # darknet53.py:106
LEVEL_CHANNELS = [1024, 512, 256]        # darknet53.py:109


class ConvUnit(nn.Module):
    """Conv(bias=False) -> BatchNorm2d -> SiLU  (darknet53.py:22-44 and twins)."""

    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, (k, k), stride=(stride, stride), padding=(k // 2, k // 2), bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.silu(self.bn(self.conv(x)))


class Residual(nn.Module):
    """x + CB3x3(CB1x1(x))  (darknet53.py:46-63; add AFTER the activation)."""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = ConvUnit(channels, channels // 2, 1)
        self.conv2 = ConvUnit(channels // 2, channels, 3)

    def forward(self, x):
        return x + self.conv2(self.conv1(x))


class Backbone(nn.Module):
    """Darknet-53 without the classifier top; returns [res5, res4, res3] (darknet53.py:112-137)."""

    def __init__(self, in_channels=3):
        super().__init__()
        width = 32
        self.conv0 = ConvUnit(in_channels, width, 3)
        for stage, blocks in enumerate(STAGE_BLOCKS, start=1):
            setattr(self, f'conv{stage}', ConvUnit(width, width * 2, 3, stride=2))
            width *= 2
            setattr(self, f'res{stage}', nn.Sequential(*[Residual(width) for _ in range(blocks)]))

    def backbone_strides_per_level(self):
        return list(LEVEL_STRIDES)

    def backbone_channels_per_level(self):
        return list(LEVEL_CHANNELS)

    def forward(self, x):
        x = self.conv0(x)
        taps = []
        for stage in range(1, 6):
            x = getattr(self, f'res{stage}')(getattr(self, f'conv{stage}')(x))
            taps.append(x)
        return [taps[4], taps[3], taps[2]]


def _five(cin, mid, names):
    """The 1x1/3x3/1x1/3x3/1x1 'YoloBlock' (yolov3neck.py:46-64) as (name, module) pairs."""
    spec = [(cin, mid, 1), (mid, mid * 2, 3), (mid * 2, mid, 1), (mid, mid * 2, 3), (mid * 2, mid, 1)]
    return [(n, ConvUnit(a, b, k)) for n, (a, b, k) in zip(names, spec)]


class _Named(nn.Module):
    def __init__(self, pairs):
        super().__init__()
        for n, m in pairs:
            self.add_module(n, m)

    def forward(self, x):
        for m in self.children():
            x = m(x)
        return x


class LibNeck(nn.Module):
    """detection/neck/yolov3neck.py:76-114; concat order [upsampled, backbone] (:105,:110)."""

    def __init__(self, ch):
        super().__init__()
        c0, c1, c2 = ch
        names = [f'conv{i}' for i in range(1, 6)]
        self.neck1 = _Named(_five(c0, c0 // 2, names))
        self.conv1 = ConvUnit(c0 // 2, c0, 3)
        self.up1 = _Named([('squeeze', ConvUnit(c0 // 2, c0 // 4, 1))])
        self.neck2 = _Named(_five(c1 + c0 // 4, c1 // 2, names))
        self.conv2 = ConvUnit(c1 // 2, c1, 3)
        self.up2 = _Named([('squeeze', ConvUnit(c1 // 2, c1 // 4, 1))])
        self.neck3 = _Named(_five(c2 + c1 // 4, c2 // 2, names))
        self.conv3 = ConvUnit(c2 // 2, c2, 3)

    def forward(self, feats):
        small, middle, large = feats
        s = self.neck1(small)
        up = F.interpolate(self.up1(s), scale_factor=2, mode='nearest')
        out_s = self.conv1(s)
        m = self.neck2(torch.cat([up, middle], dim=1))
        up = F.interpolate(self.up2(m), scale_factor=2, mode='nearest')
        out_m = self.conv2(m)
        l = self.neck3(torch.cat([up, large], dim=1))
        return [out_s, out_m, self.conv3(l)]


class LibHead(nn.Module):
    """detection/head/yolov3head.py:42-67: biased 1x1 conv then [B,A,5+C,H,W]->[B,A,H,W,5+C] contiguous."""

    def __init__(self, ch, anchors_per_level, num_classes):
        super().__init__()
        self.per_anchor = num_classes + 5
        self.anchors_per_level = list(anchors_per_level)
        self.heads = nn.ModuleList(nn.Conv2d(c, self.per_anchor * a, (1, 1), bias=True)
                                   for c, a in zip(ch, anchors_per_level))

    def forward(self, feats):
        outs = []
        for conv, a, f in zip(self.heads, self.anchors_per_level, feats):
            y = conv(f)
            b, _, h, w = y.shape
            outs.append(y.view(b, a, self.per_anchor, h, w).permute(0, 1, 3, 4, 2).contiguous())
        return outs


class LibYolov3(nn.Module):
    """detection/models/yolov3.py:6-54 (train branch returns the raw head list, :54).

    ``anchors`` are pixel-unit [9,2] (largest level first) and are split per level into
    [A,1,1,2] tensors (:12-17).  The eval/``val=True`` branch decodes boxes (:35-53); the
    reference's ``offset`` helper does not exist, its inferred meaning
    is the [H,W,(x,y)] grid.
    """

    def __init__(self, anchors, num_anchors_per_level=(3, 3, 3), in_channels=3, num_classes=80, training=False):
        super().__init__()
        self.training = training
        flat = anchors.view(-1, 2)
        self.anchors_per_level, start = [], 0
        for n in num_anchors_per_level:
            self.anchors_per_level.append(flat[start:start + n].view(n, 1, 1, 2))
            start += n
        self.num_classes = num_classes
        self.backbone = Backbone(in_channels)
        self.backbone_strides_per_level = self.backbone.backbone_strides_per_level()
        self.backbone_channels_per_level = self.backbone.backbone_channels_per_level()
        self.neck = LibNeck(self.backbone_channels_per_level)
        self.head = LibHead(self.backbone_channels_per_level, num_anchors_per_level, num_classes)

    def forward(self, images, val=False):
        head_out = self.head(self.neck(self.backbone(images)))
        if self.training and not val:
            return head_out
        decoded = []
        for lvl, out in enumerate(head_out):
            b, a, h, w, _ = out.shape
            ys = torch.arange(h).view(h, 1).expand(h, w)
            xs = torch.arange(w).view(1, w).expand(h, w)
            cell = torch.stack([xs, ys], dim=2).to(out).expand_as(out[..., 0:2])
            xy = (out[..., 0:2].sigmoid() + cell) * self.backbone_strides_per_level[lvl]
            wh = torch.exp(out[..., 2:4]) * self.anchors_per_level[lvl].expand_as(out[..., 2:4]).to(out)
            decoded.append(torch.cat((xy, wh, out[..., 4:].sigmoid()), -1).view(b, -1, self.num_classes + 5))
        return head_out, torch.cat(decoded, 1)


# ------------------------------------------------------------------------------------------ demo surface
class DemoNeck(nn.Module):
    """demos/yolov3_u/models/yolov3.py:43-103; concat order [backbone, upsampled] (:96,:100)."""

    def __init__(self, c0=1024, c1=512, c2=256):
        super().__init__()
        idx = ['0', '1', '2', '3', '4']
        self.neck_small = _Named(_five(c0, c0 // 2, idx))
        self.neck_out_small = ConvUnit(c0 // 2, c0, 3)
        self.up_sampling_small = _Named([('0', ConvUnit(c0 // 2, c0 // 4, 1))])
        self.neck_medium = _Named(_five(c1 + c0 // 4, c1 // 2, idx))
        self.neck_out_medium = ConvUnit(c1 // 2, c1, 3)
        self.up_sampling_medium = _Named([('0', ConvUnit(c1 // 2, c1 // 4, 1))])
        self.neck_large = _Named(_five(c2 + c1 // 4, c2 // 2, idx))
        self.neck_out_large = ConvUnit(c2 // 2, c2, 3)

    def forward(self, small, medium, large):
        s = self.neck_small(small)
        out_s = self.neck_out_small(s)
        up = F.interpolate(self.up_sampling_small(s), scale_factor=2, mode='nearest')
        m = self.neck_medium(torch.cat([medium, up], dim=1))
        out_m = self.neck_out_medium(m)
        up = F.interpolate(self.up_sampling_medium(m), scale_factor=2, mode='nearest')
        l = self.neck_large(torch.cat([large, up], dim=1))
        return out_s, out_m, self.neck_out_large(l)


class DemoHead(nn.Module):
    """demos/yolov3_u/models/yolov3.py:105-137; convs are created large -> medium -> small (:119-123)."""

    def __init__(self, anchors, num_classes, c0=1024, c1=512, c2=256):
        super().__init__()
        per = 5 + num_classes
        self.head_out_large = nn.Conv2d(c2, anchors[2].size(0) * per, (1, 1), bias=True)
        self.head_out_medium = nn.Conv2d(c1, anchors[1].size(0) * per, (1, 1), bias=True)
        self.head_out_small = nn.Conv2d(c0, anchors[0].size(0) * per, (1, 1), bias=True)

    def forward(self, s, m, l):
        return self.head_out_small(s), self.head_out_medium(m), self.head_out_large(l)


class DemoYoloV3(nn.Module):
    """demos/yolov3_u/models/yolov3.py:139-175; ``anchors`` = 3 feature-scale [3,2] tensors; raw NCHW heads."""

    def __init__(self, in_channels=3, num_classes=80, anchors=()):
        super().__init__()
        self.anchors = anchors
        self.backbone = Backbone(in_channels)
        self.neck = DemoNeck()
        self.head = DemoHead(anchors, num_classes)

    def forward(self, x):
        return self.head(*self.neck(*self.backbone(x)))


COCO_ANCHORS_PX = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]


def coco_anchors_px():
    """[9,2] pixel anchors, largest level first (demos/yolov3_u/train.py:60-62 before the division)."""
    return torch.tensor(COCO_ANCHORS_PX, dtype=torch.float32).view(-1, 2)


def coco_anchors_feature():
    """The demo's three feature-scale [3,2] anchor tensors (train.py:60-62)."""
    a = coco_anchors_px().view(3, 3, 2)
    return tuple(a[i] / s for i, s in enumerate(LEVEL_STRIDES))
