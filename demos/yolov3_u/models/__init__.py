"""Shim package: ``models`` of the reference's demos/yolov3_u (train.py:12 ``from models.yolov3 import YoloV3``)."""
from fastvision_amd.demos.yolov3_u.models import *  # noqa: F401,F403
from fastvision_amd.demos.yolov3_u.models import YoloV3, darknet53  # noqa: F401
