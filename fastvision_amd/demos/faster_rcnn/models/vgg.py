"""VGG16 backbone of the reference's Faster R-CNN demo on the HIP kernels -- API mirror of demos/faster_rcnn/models/vgg.py
(class VGG, factory vgg16): five stages ``vgg1`` .. ``vgg5`` of ``Conv2d(3x3, bias) -> ReLU`` pairs, a 2x2 max-pool after each of
the first four, the stride-16 ``vgg5`` feature map returned, and the two-layer ``classifier`` the Fast head borrows
(faster.py:78).  Same attribute names and ``state_dict()`` keys (``vgg3.4.weight``, ``classifier.3.bias`` ...), so the reference's
checkpoints load; only the un-normalised variants (``normal=False``) are on this path.

The Conv2d / ReLU modules are containers for parameters and key layout: ``forward`` runs each pair as one fused
``vgg_ops.conv_bias_relu`` (bias + ReLU in the conv epilogue, halo NHWC activations) and ``vgg_ops.max_pool2``.
"""
import torch.nn as nn

from ....vgg_ops import conv_bias_relu, max_pool2

__all__ = ['VGG', 'vgg16']

STAGE_WIDTHS = (64, 128, 256, 512, 512)


class VGG(nn.Module):
    def __init__(self, in_channels, num_classes, num_blocks, channels, normal=False):
        super().__init__()
        if normal:
            raise NotImplementedError('the BatchNorm variants of VGG are not on the Faster R-CNN path (faster.py uses vgg16)')
        width = in_channels
        for stage, (blocks, out) in enumerate(zip(num_blocks, channels), start=1):
            layers = []
            for _ in range(blocks):
                layers += [nn.Conv2d(width, out, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=True), nn.ReLU(inplace=True)]
                width = out
            setattr(self, f'vgg{stage}', nn.Sequential(*layers))
        self.maxpool = nn.MaxPool2d(kernel_size=(2, 2), stride=2)
        self.classifier = nn.Sequential(nn.Linear(channels[3] * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(True),
                                        nn.Dropout())

    @staticmethod
    def _stage(x, seq):
        for layer in seq:
            if isinstance(layer, nn.Conv2d):          # the ReLU that follows it in the Sequential is fused into this call
                x = conv_bias_relu(x, layer)
        return x

    def forward(self, x):
        for stage in (self.vgg1, self.vgg2, self.vgg3, self.vgg4):
            x = max_pool2(self._stage(x, stage))
        return self._stage(x, self.vgg5)


def vgg16(in_channels=3, num_classes=1000):
    return VGG(in_channels=in_channels, num_classes=num_classes, num_blocks=[2, 2, 3, 3, 3], channels=list(STAGE_WIDTHS))
