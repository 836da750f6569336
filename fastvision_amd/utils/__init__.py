from .checkpoints import *  # noqa: F401,F403
from .fit import *          # noqa: F401,F403
from .device import *  # noqa: F401,F403
from .seed import *    # noqa: F401,F403
