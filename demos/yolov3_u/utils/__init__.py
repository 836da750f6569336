"""Shim package: ``utils`` of the reference's demos/yolov3_u (train.py:10-11,14; inference.py:21)."""
from fastvision_amd.demos.yolov3_u.utils import *  # noqa: F401,F403
from fastvision_amd.demos.yolov3_u.utils import ComputeLoss, grid, mean_average_precision, non_max_suppression, non_max_suppression_batch, xywh2xyxy  # noqa: F401
