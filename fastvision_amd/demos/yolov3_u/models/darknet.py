"""Demo twin of Darknet-53 -- API mirror of the reference's demos/yolov3_u/models/darknet.py.

Identical to the library backbone except that the activation attribute is called ``act`` and ResidualBlock has no
``shortcut`` flag; the 312 ``backbone.*`` state_dict keys are shared.
"""
from .... import ops
from ....classfication.models.darknet53 import ConvBlock1x1 as _CB1, ConvBlock3x3 as _CB3, Darknet as _Darknet
from ....classfication.models.darknet53 import ResidualBlock as _Res, activation, conv1x1, conv3x3, normalization

__all__ = ['conv3x3', 'conv1x1', 'normalization', 'activation', 'ConvBlock3x3', 'ConvBlock1x1', 'ResidualBlock', 'Darknet',
           'darknet53']

class ConvBlock3x3(_CB3):
    act_name = 'act'


class ConvBlock1x1(_CB1):
    act_name = 'act'


class ResidualBlock(_Res):
    block1x1, block3x3 = ConvBlock1x1, ConvBlock3x3

    def __init__(self, in_channels, mid_channels):
        super().__init__(in_channels, mid_channels)
        del self.shortcut                      # the demo block always adds the identity (darknet.py:42-55)

    def forward(self, x):
        return ops.residual(x, self.conv1, self.conv2)


class Darknet(_Darknet):
    block3x3, resblock = ConvBlock3x3, ResidualBlock


def darknet53(in_channels=3, num_classes=1000, including_top=True):
    return Darknet(in_channels=in_channels, num_classes=num_classes, num_blocks=[1, 2, 8, 8, 4], including_top=including_top)
