from .fast import Fast  # noqa: F401
from .faster import Faster_Rcnn  # noqa: F401
from .rpn import RPN, FocalLoss  # noqa: F401
from .vgg import VGG, vgg16  # noqa: F401
