from .tools import *  # noqa: F401,F403
