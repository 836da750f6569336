from .map import *   # noqa: F401,F403
