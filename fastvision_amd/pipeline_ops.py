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
