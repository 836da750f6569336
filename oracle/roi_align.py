"""CPU restatement of RoIAlign as the reference's two-stage head calls it (scope row f-4, first operator).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference call sites: demos/faster_rcnn/models/fast.py:227-231 (training: positive / negative samples) and :258 (inference),
`torchvision.ops.roi_align(feature_backbone, boxes[K,5], output_size=(7, 7))` -- i.e. spatial_scale = 1.0,
sampling_ratio = -1 (adaptive), aligned = False; boxes = (batch index, x1, y1, x2, y2) in feature-map cells; the result
[K, C, 7, 7] is flattened to [K, C*49] for the VGG classifier (fast.py:233).

PARITY UNPINNED: the arithmetic is torchvision's (0.11.2 pinned by the reference's requirements; the module is absent from
this image and from /root/reference).  This file restates its published algorithm (torchvision/csrc/ops/cpu/
roi_align_kernel.cpp, roi_align_common.h):
  * roi_start = box * scale; roi_size = max(end - start, 1) (aligned = False); bin = roi_size / pooled;
  * samples per bin: ceil(roi_size / pooled) per axis (sampling_ratio <= 0), at start + p * bin + (i + .5) * bin / grid;
  * a sample outside [-1, size] contributes 0; coordinates are clamped at 0, the upper neighbour at size - 1 (the sample is
    then moved onto the last row / column); bilinear weights hy*hx, hy*lx, ly*hx, ly*lx; the bin is the mean of its samples;
  * backward: each sample's four weights / count scatter the bin's gradient back (sum over all contributions).
"""
import math

import numpy as np


def _samples(box, pooled, size_h, size_w, spatial_scale=1.0, sampling_ratio=-1):
    """Yield (ph, pw, [(y_low, x_low, y_high, x_high, w1, w2, w3, w4)], count) for one box (float32 arithmetic like the C++)."""
    f = np.float32
    x1, y1, x2, y2 = (f(v) * f(spatial_scale) for v in box)
    roi_w = max(f(x2 - x1), f(1.0))
    roi_h = max(f(y2 - y1), f(1.0))
    ph_n, pw_n = pooled
    bin_h, bin_w = f(roi_h / f(ph_n)), f(roi_w / f(pw_n))
    grid_h = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(f(roi_h / f(ph_n)))))
    grid_w = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(f(roi_w / f(pw_n)))))
    count = max(grid_h * grid_w, 1)
    for ph in range(ph_n):
        for pw in range(pw_n):
            taps = []
            for iy in range(grid_h):
                y = f(y1 + f(ph) * bin_h + f(f(iy) + f(0.5)) * bin_h / f(grid_h))
                for ix in range(grid_w):
                    x = f(x1 + f(pw) * bin_w + f(f(ix) + f(0.5)) * bin_w / f(grid_w))
                    if y < -1.0 or y > size_h or x < -1.0 or x > size_w:
                        continue
                    yy, xx = max(y, f(0.0)), max(x, f(0.0))
                    y_low, x_low = int(yy), int(xx)
                    if y_low >= size_h - 1:
                        y_high = y_low = size_h - 1
                        yy = f(y_low)
                    else:
                        y_high = y_low + 1
                    if x_low >= size_w - 1:
                        x_high = x_low = size_w - 1
                        xx = f(x_low)
                    else:
                        x_high = x_low + 1
                    ly, lx = f(yy - f(y_low)), f(xx - f(x_low))
                    hy, hx = f(f(1.0) - ly), f(f(1.0) - lx)
                    taps.append((y_low, x_low, y_high, x_high, f(hy * hx), f(hy * lx), f(ly * hx), f(ly * lx)))
            yield ph, pw, taps, count


def roi_align(features, boxes, output_size=(7, 7), spatial_scale=1.0, sampling_ratio=-1):
    """features [B, C, H, W] float32, boxes [K, 5] (batch index, x1, y1, x2, y2) -> [K, C, PH, PW] float32."""
    features = np.asarray(features, dtype=np.float32)
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 5)
    B, C, H, W = features.shape
    ph_n, pw_n = output_size
    out = np.zeros((len(boxes), C, ph_n, pw_n), dtype=np.float32)
    for k, box in enumerate(boxes):
        fm = features[int(box[0])]
        for ph, pw, taps, count in _samples(box[1:], (ph_n, pw_n), H, W, spatial_scale, sampling_ratio):
            acc = np.zeros(C, dtype=np.float32)
            for yl, xl, yh, xh, w1, w2, w3, w4 in taps:
                acc += w1 * fm[:, yl, xl] + w2 * fm[:, yl, xh] + w3 * fm[:, yh, xl] + w4 * fm[:, yh, xh]
            out[k, :, ph, pw] = acc / np.float32(count)
    return out


def roi_align_backward(grad_out, boxes, feature_shape, spatial_scale=1.0, sampling_ratio=-1):
    """grad_out [K, C, PH, PW] -> gradient w.r.t. the features [B, C, H, W] (float64 accumulation: order-independent)."""
    grad_out = np.asarray(grad_out, dtype=np.float32)
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 5)
    B, C, H, W = feature_shape
    ph_n, pw_n = grad_out.shape[2:]
    g = np.zeros(feature_shape, dtype=np.float64)
    for k, box in enumerate(boxes):
        gb = g[int(box[0])]
        for ph, pw, taps, count in _samples(box[1:], (ph_n, pw_n), H, W, spatial_scale, sampling_ratio):
            go = grad_out[k, :, ph, pw].astype(np.float64) / count
            for yl, xl, yh, xh, w1, w2, w3, w4 in taps:
                gb[:, yl, xl] += go * w1
                gb[:, yl, xh] += go * w2
                gb[:, yh, xl] += go * w3
                gb[:, yh, xh] += go * w4
    return g.astype(np.float32)
