"""Shim package: ``cfg`` of the reference's demos/yolov3_u (train.py:12 ``from cfg._fit import Fit``)."""
