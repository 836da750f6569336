from .fit import *  # noqa: F401,F403
