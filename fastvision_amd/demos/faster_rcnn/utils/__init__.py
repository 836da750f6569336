from .anchor_generator import get_base_anchor  # noqa: F401
from .nms import non_max_suppression  # noqa: F401
from ...yolov3_u.utils.box import grid, xywh2xyxy  # noqa: F401
from ...yolov3_u.utils.map import mean_average_precision  # noqa: F401
