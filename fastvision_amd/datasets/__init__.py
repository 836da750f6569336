from .detection_dataloader import *   # noqa: F401,F403
