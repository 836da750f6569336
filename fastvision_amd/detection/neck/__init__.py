from .yolov3neck import yolov3neck  # noqa: F401
