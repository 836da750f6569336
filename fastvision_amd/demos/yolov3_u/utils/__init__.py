from .lossv3 import ComputeLoss  # noqa: F401
from .nms import non_max_suppression, non_max_suppression_batch  # noqa: F401
from .box import grid, xywh2xyxy  # noqa: F401
from .map import mean_average_precision  # noqa: F401
