"""Box format conversions -- API mirror of the reference's detection/tools/BOX.py (host-side layout helpers;
they accept torch tensors on any device or numpy arrays, exactly like the reference)."""
import numpy as np
import torch

__all__ = ['xywh2xyxy', 'xyxy2xywh', 'xyxy2xywhn']


def _stack(cols, like):
    return torch.stack(cols, dim=1) if isinstance(like, torch.Tensor) else np.stack(cols, axis=1)


def xywh2xyxy(xywh):
    hw, hh = xywh[:, 2] / 2, xywh[:, 3] / 2
    return _stack([xywh[:, 0] - hw, xywh[:, 1] - hh, xywh[:, 0] + hw, xywh[:, 1] + hh], xywh)


def xyxy2xywh(xyxy):
    return _stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2,
                   xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]], xyxy)


def xyxy2xywhn(xyxy, heigth, width):
    return _stack([((xyxy[:, 0] + xyxy[:, 2]) / 2) / width, ((xyxy[:, 1] + xyxy[:, 3]) / 2) / heigth,
                   (xyxy[:, 2] - xyxy[:, 0]) / width, (xyxy[:, 3] - xyxy[:, 1]) / heigth], xyxy)
