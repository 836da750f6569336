"""Shim: `utils.nms` of the reference's demos/yolov3_u resolves to `fastvision_amd.demos.yolov3_u.utils.nms` (the same module object)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module('fastvision_amd.demos.yolov3_u.utils.nms')
