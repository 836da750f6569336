from .yolov3head import yolov3head  # noqa: F401
