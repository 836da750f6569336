"""API mirror of the reference's demos/yolov3_u/utils/box.py (host-side layout helpers)."""
import numpy as np
import torch

from ....detection.tools.BOX import xywh2xyxy, xyxy2xywh, xyxy2xywhn  # noqa: F401


def xywhn2xyxy(xywhn, height, width):
    cols = [(xywhn[:, 0] - xywhn[:, 2] / 2) * width, (xywhn[:, 1] - xywhn[:, 3] / 2) * height,
            (xywhn[:, 0] + xywhn[:, 2] / 2) * width, (xywhn[:, 1] + xywhn[:, 3] / 2) * height]
    return torch.stack(cols, dim=1) if isinstance(xywhn, torch.Tensor) else np.stack(cols, axis=1)


def grid(height, width, mode='xy'):
    from ....detection.tools.GRID import grid as _grid
    return _grid(height, width, mode=mode, dtype='torch')
