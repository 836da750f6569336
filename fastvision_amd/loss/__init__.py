from .yolov3_loss import *          # noqa: F401,F403
from .classification_loss import *  # noqa: F401,F403
from .iou_loss import *             # noqa: F401,F403
