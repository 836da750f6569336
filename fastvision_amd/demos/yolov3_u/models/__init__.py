from .yolov3 import YoloV3  # noqa: F401
from .darknet import darknet53  # noqa: F401
