from .yolov3 import *  # noqa: F401,F403
