"""Darknet-53 backbone on the MI355X kernels -- API mirror of the reference's classfication/models/darknet53.py.

Same class names, constructor signatures, state_dict keys (``conv0.conv.weight``, ``res3.7.conv2.bn.running_var`` ...)
and parameter construction order (so ``torch.manual_seed`` reproduces the reference's init).  The modules only
*own* parameters; their forward runs the fused HIP path of ``fastvision_amd.ops``:
conv (MFMA implicit GEMM) -> BatchNorm batch statistics -> SiLU (+ residual add), one autograd node per
ConvBlock / ResidualBlock.
"""
import torch
import torch.nn as nn

from ... import ops

__all__ = ['conv3x3', 'conv1x1', 'normalization', 'activation', 'ConvBlock3x3', 'ConvBlock1x1', 'ResidualBlock',
           'Darknet', 'darknet53']


def conv3x3(in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), groups=1, bias=False):
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, groups=groups, bias=bias)


def conv1x1(in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), groups=1, bias=False):
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, groups=groups, bias=bias)


def normalization(num_features):
    return nn.BatchNorm2d(num_features=num_features)


def activation():
    return nn.SiLU()


def _check_supported(conv):
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    ok = (k in ((1, 1), (3, 3)) and s[0] == s[1] and s[0] in (1, 2) and p == (k[0] // 2, k[1] // 2)
          and conv.groups == 1 and conv.bias is None and conv.dilation == (1, 1) and not (k == (1, 1) and s[0] != 1))
    if not ok:
        raise NotImplementedError('fastvision_amd ConvBlock supports k in {1,3}, pad k//2, stride {1,2} (3x3) / 1 (1x1), '
                                  f'groups 1, no bias; got {conv}')


class _ConvBlock(nn.Module):
    """Conv(bias=False) -> BatchNorm2d -> SiLU (reference darknet53.py:22-44)."""

    act_name = 'relu'     # the library calls the activation ``relu``; the demo twin calls it ``act``

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, groups=1, bias=False):
        super().__init__()
        make = conv3x3 if tuple(kernel_size) == (3, 3) else conv1x1
        self.conv = make(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, groups=groups, bias=bias)
        self.bn = normalization(out_channels)
        setattr(self, self.act_name, activation())
        _check_supported(self.conv)

    def forward(self, x):
        conv = self.conv
        if conv.in_channels <= 3 and conv.out_channels == 32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1):
            return ops.stem(x, conv, self.bn)          # network input: fp32 NCHW images, no input gradient
        return ops.conv_bn_silu(x, conv, self.bn)


class ConvBlock3x3(_ConvBlock):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), groups=1, bias=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, groups, bias)


class ConvBlock1x1(_ConvBlock):
    def __init__(self, in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), groups=1, bias=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, groups, bias)


class ResidualBlock(nn.Module):
    """identity + CB3x3(CB1x1(x)) (reference darknet53.py:46-63), one fused autograd node."""

    block1x1, block3x3 = ConvBlock1x1, ConvBlock3x3

    def __init__(self, in_channels, mid_channels):
        super().__init__()
        self.conv1 = self.block1x1(in_channels=in_channels, out_channels=mid_channels)
        self.conv2 = self.block3x3(in_channels=mid_channels, out_channels=mid_channels * 2)
        self.shortcut = (in_channels == mid_channels * 2)

    def forward(self, x):
        if self.shortcut:
            return ops.residual(x, self.conv1, self.conv2)
        return self.conv2(self.conv1(x))


class Darknet(nn.Module):
    """reference darknet53.py:65-137.  ``including_top=False`` returns [res5, res4, res3]."""

    block3x3, resblock = ConvBlock3x3, ResidualBlock

    def __init__(self, in_channels, num_classes, num_blocks, including_top=True):
        super().__init__()
        self.including_top = including_top
        self.planes = 32
        self.conv0 = self.block3x3(in_channels=in_channels, out_channels=self.planes, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1))
        for stage, (blocks, mid) in enumerate(zip(num_blocks, (32, 64, 128, 256, 512)), start=1):
            setattr(self, f'conv{stage}', self.block3x3(in_channels=self.planes, out_channels=self.planes * 2,
                                                        kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)))
            self.planes *= 2
            setattr(self, f'res{stage}', self._make_layer(num_blocks=blocks, mid_channels=mid))
        if self.including_top:
            self.gap = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(self.planes, num_classes)

    def _make_layer(self, num_blocks, mid_channels):
        return nn.Sequential(*[self.resblock(in_channels=self.planes, mid_channels=mid_channels) for _ in range(num_blocks)])

    def backbone_strides_per_level(self):
        return [32, 16, 8]

    def backbone_channels_per_level(self):
        return [1024, 512, 256]

    def forward(self, x):
        x = self.conv0(x)
        taps = []
        for stage in range(1, 6):
            # inside a stage every block's output is consumed by the next block's conv1 (1x1) first: its apply pass may ride in that
            # launch (ops.defer_apply_scope); the stage's last output is materialised when the scope closes
            with ops.defer_apply_scope():
                x = getattr(self, f'res{stage}')(getattr(self, f'conv{stage}')(x))
            taps.append(x)
        if self.including_top:      # classifier top: outside the accelerated path, plain torch ops on the fp32 copy
            out = torch.flatten(self.gap(taps[4].float()), 1)
            return self.fc(out)
        return [taps[4], taps[3], taps[2]]


def darknet53(in_channels=3, num_classes=1000, including_top=True):
    return Darknet(in_channels=in_channels, num_classes=num_classes, num_blocks=[1, 2, 8, 8, 4], including_top=including_top)
