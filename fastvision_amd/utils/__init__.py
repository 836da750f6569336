from .checkpoints import *  # noqa: F401,F403
from .fit import *          # noqa: F401,F403
