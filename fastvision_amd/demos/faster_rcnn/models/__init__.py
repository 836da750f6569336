from .rpn import RPN, FocalLoss  # noqa: F401
from .vgg import VGG, vgg16  # noqa: F401
