from .BOX import *   # noqa: F401,F403
from .GRID import *  # noqa: F401,F403
from .IOU import *   # noqa: F401,F403
from .NMS import *   # noqa: F401,F403
from .ANCHOR import *   # noqa: F401,F403
