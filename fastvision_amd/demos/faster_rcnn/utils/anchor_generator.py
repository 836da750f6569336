"""get_base_anchor (demos/faster_rcnn/utils/anchor_generator.py:4-13): the [len(ratios) * len(scales), 2] (w, h) anchor shapes of area
scale^2 and aspect h / w = ratio, ratio-major."""
import math

import numpy as np

__all__ = ['get_base_anchor']


def get_base_anchor(scales, ratios):
    out = []
    for ratio in ratios:
        for scale in scales:
            w = math.sqrt(scale ** 2 / ratio)
            out.append((w, scale ** 2 / w))
    return np.array(out, dtype=np.float32).reshape([-1, 2])
