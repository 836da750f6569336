from .lossv3 import ComputeLoss  # noqa: F401
