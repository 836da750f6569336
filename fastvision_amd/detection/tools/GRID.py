"""Cell grids -- API mirror of the reference's detection/tools/GRID.py (host-side helper)."""
import numpy as np
import torch

__all__ = ['grid']


def grid(height, width, mode='xy', dtype='torch'):
    """torch: mode 'xy' -> [H,W,2] with (x, y) per cell, otherwise its [W,H,2] transpose (GRID.py:18-29).
    numpy: np.meshgrid's default 'xy' indexing swaps the two modes (GRID.py:6-16)."""
    if dtype == 'torch':
        ys = torch.arange(height).view(height, 1).expand(height, width)
        xs = torch.arange(width).view(1, width).expand(height, width)
        g = torch.stack([xs, ys], dim=2)
        return g if mode == 'xy' else g.permute(1, 0, 2)
    ys, xs = np.arange(height).reshape(height, 1), np.arange(width).reshape(1, width)
    g = np.stack([np.broadcast_to(xs, (height, width)), np.broadcast_to(ys, (height, width))], axis=2)
    return g.transpose(1, 0, 2) if mode == 'xy' else g
