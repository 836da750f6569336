from .darknet53 import *  # noqa: F401,F403  (mirrors classfication/models/__init__.py of the reference)
