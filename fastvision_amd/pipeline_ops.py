"""Device-side input pipeline ops (scope row f-3) through the C ABI (``fva_paste_resize_normalize``,
``fva_paste_resize_u8`` -- fastvision_amd/csrc/pipeline.hip): decoded uint8 images in, network input batch out.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

__all__ = ['PasteJob', 'pack_images', 'paste_batch', 'paste_batch_u8', 'value_table', 'canvas_sources']


class PasteJob:
    """One resize-and-place step: source image index, resized size, position on canvas ``canvas``, mirror flags."""
    __slots__ = ('image', 'canvas', 'dst_h', 'dst_w', 'top', 'left', 'flip_h', 'flip_v')

    def __init__(self, image, canvas, dst_h, dst_w, top, left, flip_h=False, flip_v=False):
        self.image, self.canvas, self.dst_h, self.dst_w = image, canvas, int(dst_h), int(dst_w)
        self.top, self.left, self.flip_h, self.flip_v = int(top), int(left), bool(flip_h), bool(flip_v)


def canvas_sources(n, canvas_h, canvas_w, shapes):
    """(offsets, shapes, pitches) describing the top-left ``shapes[i]`` sub-images of a uint8 [n, canvas_h, canvas_w, 3]
    tensor as sources for a further paste pass."""
    offsets = np.arange(n, dtype=np.int64) * (canvas_h * canvas_w * 3)
    return offsets, list(shapes), [canvas_w * 3] * n


def value_table(mean=None, std=None, single=False):
    """float32 [3,256] table of the reference's per-byte arithmetic: ``img / 255.`` (float64) and, with mean/std (float32
    [3]), ``(img / 255. - mean) / std`` rounded to float32 once at the end (augmentation.py:367-371 + the later astype).
    single=True: the demo's float32 ``image.to(torch.float32) / 255.`` (data_gen.py:352-353)."""
    if single:
        return np.ascontiguousarray(np.repeat((np.arange(256, dtype=np.float32) / np.float32(255.)).reshape(1, 256), 3, axis=0))
    v = np.arange(256, dtype=np.uint8).reshape(256, 1) / 255.
    if mean is not None:
        v = (v - np.asarray(mean, dtype=np.float32).reshape(1, 3)) / np.asarray(std, dtype=np.float32).reshape(1, 3)
    else:
        v = np.repeat(v, 3, axis=1)
    return np.ascontiguousarray(v.T).astype(np.float32)


def pack_images(images, pin=True):
    """Concatenate uint8 HWC RGB arrays into one (pinned) byte tensor; returns (buffer, offsets, shapes)."""
    sizes = [int(im.shape[0]) * int(im.shape[1]) * 3 for im in images]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    buf = torch.empty(int(offsets[-1]), dtype=torch.uint8)
    if pin and torch.cuda.is_available():
        buf = buf.pin_memory()
    flat = buf.numpy()
    for im, o, n in zip(images, offsets[:-1], sizes):
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError('images must be uint8 arrays of shape [H, W, 3]')
        flat[o:o + n] = np.ascontiguousarray(im).reshape(-1)
    return buf, offsets[:-1], [(int(im.shape[0]), int(im.shape[1])) for im in images]


def _job_tables(jobs, offsets, shapes, n_canvas, device, pitches=None):
    jobs = sorted(jobs, key=lambda j: j.canvas)            # stable: paste order inside a canvas is kept
    arr = (_lib.PasteJob * max(len(jobs), 1))()
    start = np.zeros(n_canvas + 1, dtype=np.int32)
    for i, j in enumerate(jobs):
        sh, sw = shapes[j.image]
        if j.dst_h < 1 or j.dst_w < 1:
            raise ValueError('paste job with an empty destination')
        a = arr[i]
        a.src_offset, a.src_h, a.src_w, a.dst_h, a.dst_w = int(offsets[j.image]), sh, sw, j.dst_h, j.dst_w
        a.top, a.left, a.flip_h, a.flip_v = j.top, j.left, int(j.flip_h), int(j.flip_v)
        a.src_pitch = int(pitches[j.image]) if pitches is not None else 0
        a.scale_x, a.scale_y = 1.0 / (float(j.dst_w) / float(sw)), 1.0 / (float(j.dst_h) / float(sh))
        start[j.canvas + 1] += 1
    start = np.cumsum(start).astype(np.int32)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device, non_blocking=True)
    return tab, torch.from_numpy(start).to(device, non_blocking=True)


def paste_batch(src, offsets, shapes, jobs, n_canvas, height, width, fill, table, device, pitches=None):
    """src: uint8 byte tensor (host or device) holding the images; returns float32 [n_canvas, 3, height, width] on device."""
    src = src.to(device, non_blocking=True)
    tab, start = _job_tables(jobs, offsets, shapes, n_canvas, device, pitches)
    lut = torch.from_numpy(table).to(device, non_blocking=True) if isinstance(table, np.ndarray) else table
    out = torch.empty(n_canvas, 3, height, width, dtype=torch.float32, device=device)
    _lib.call('fva_paste_resize_normalize', _p(src), _p(tab), _p(start), n_canvas, height, width, int(fill), _p(lut), _p(out), _stream())
    return out


def paste_batch_u8(src, offsets, shapes, jobs, n_canvas, height, width, fill, device, pitches=None):
    """As paste_batch, without the value table: uint8 [n_canvas, height, width, 3] on device (an intermediate image)."""
    src = src.to(device, non_blocking=True)
    tab, start = _job_tables(jobs, offsets, shapes, n_canvas, device, pitches)
    out = torch.empty(n_canvas, height, width, 3, dtype=torch.uint8, device=device)
    _lib.call('fva_paste_resize_u8', _p(src), _p(tab), _p(start), n_canvas, height, width, int(fill), _p(out), _stream())
    return out


# ---- colour / blur stage of the demo's training path (csrc/colour.hip) --------------------------------------------------------
def _colour_jobs(regions, clahe, hsv, blur, perms, device):
    n = len(regions)
    arr = (_lib.ColourJob * n)()
    for i, (h, w) in enumerate(regions):
        a = arr[i]
        a.h, a.w = int(h), int(w)
        a.clahe, a.hsv, a.blur = int(bool(clahe[i])), int(bool(hsv[i])), int(blur[i])
        for c in range(3):
            a.perm[c] = int(perms[i][c])
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device, non_blocking=True)


def hsv_tables(gains):
    """The three byte tables of HueSaturationValue for gains r (float64 [3], data_gen.py:125-131): (x * r0) % 180, clip(x * r1, 0, 255),
    clip(x * r2, 0, 255), truncated to uint8 -- computed on the host exactly as the reference computes them."""
    x = np.arange(0, 256, dtype=np.float64)
    g = np.asarray(gains, dtype=np.float64)
    return np.stack([((x * g[0]) % 180).astype(np.uint8), np.clip(x * g[1], 0, 255).astype(np.uint8), np.clip(x * g[2], 0, 255).astype(np.uint8)])


def clahe_hsv_(canvases, regions, clahe, hsv_luts):
    """In place on uint8 canvases [n,H,W,3] (device): per image [HistEqualize: CLAHE on the luma] then [HueSaturationValue tables].
    regions: (h, w) of every image's valid top-left area; clahe: flags; hsv_luts: per image None or uint8 [3,256]."""
    n, H, W, _ = canvases.shape
    if not any(clahe) and not any(l is not None for l in hsv_luts):
        return canvases
    dev = canvases.device
    jobs = _colour_jobs(regions, clahe, [l is not None for l in hsv_luts], [0] * n, [(0, 1, 2)] * n, dev)
    tabs = np.zeros((n, 3, 256), dtype=np.uint8)
    for i, l in enumerate(hsv_luts):
        if l is not None:
            tabs[i] = l
    luts = torch.from_numpy(tabs).to(dev, non_blocking=True)
    ws = torch.empty(_lib.load().fva_colour_workspace(n), dtype=torch.uint8, device=dev)
    _lib.call('fva_colour_clahe_hsv', _p(canvases), n, H, W, _p(jobs), max(int(r[0]) for r in regions), max(int(r[1]) for r in regions),
              _p(luts), _p(ws), _stream())
    return canvases


def blur_shuffle_normalize(canvases, blur, perms, table):
    """uint8 canvases [n,H,W,3] (device) -> float32 [n,3,H,W]: per image a 3x3 blur (0 none, 1 box, 2 median, 3 Gaussian), the channel
    permutation (output c = input perms[i][c]) and the byte -> value table (x / 255)."""
    n, H, W, _ = canvases.shape
    dev = canvases.device
    jobs = _colour_jobs([(H, W)] * n, [0] * n, [0] * n, blur, perms, dev)
    lut = torch.from_numpy(table).to(dev, non_blocking=True) if isinstance(table, np.ndarray) else table
    out = torch.empty(n, 3, H, W, dtype=torch.float32, device=dev)
    _lib.call('fva_colour_blur_shuffle_normalize', _p(canvases), n, H, W, _p(jobs), _p(lut), _p(out), _stream())
    return out
