"""RoIAlign on the HIP kernel -- the operator the reference's two-stage head takes from torchvision
(demos/faster_rcnn/models/fast.py:227-231,258: ``torchvision.ops.roi_align(feature_backbone, boxes, output_size=(7, 7))``).

    roi_align(features [B,C,H,W], boxes [K,5] = (batch index, x1, y1, x2, y2), output_size, spatial_scale=1.0,
              sampling_ratio=-1) -> [K, C, PH, PW] float32

Same argument meaning and defaults as the torchvision call (aligned = False only, which is what the reference uses).  The
features may be any CUDA tensor; a view of one of this package's halo NHWC buffers (what the conv blocks produce) is read
in place, anything else is packed once.  No CPU path: raises for CPU tensors.
"""
import ctypes as C

import torch

from . import _lib
from .ops import _code, _p, _stream, get_compute_dtype, require_gpu, to_halo

__all__ = ['roi_align']


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, boxes, output_size, spatial_scale, sampling_ratio):
        require_gpu(features, 'roi_align')
        if boxes.dim() != 2 or boxes.size(1) != 5:
            raise RuntimeError(f'roi_align: boxes must be [K, 5] (batch index, x1, y1, x2, y2), got {tuple(boxes.shape)}')
        B, Cc, H, W = features.shape
        PH, PW = _pair(output_size)
        dtype = features.dtype if features.dtype in (torch.float32, torch.bfloat16) else torch.float32
        keep, base, pad = to_halo(features.detach(), dtype, 0)
        rois = boxes.detach().to(device=features.device, dtype=torch.float32).contiguous()
        K = rois.size(0)
        out = torch.empty((K, Cc, PH, PW), dtype=torch.float32, device=features.device)
        _lib.call('fva_roi_align_fwd', _code(dtype), C.c_void_p(base), pad, _p(rois), K, _p(out), B, H, W, Cc, PH, PW,
                  float(spatial_scale), int(sampling_ratio), _stream())
        ctx.rois, ctx.shape, ctx.args, ctx.in_dtype = rois, (B, Cc, H, W), (PH, PW, float(spatial_scale), int(sampling_ratio)), features.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        B, Cc, H, W = ctx.shape
        PH, PW, scale, sampling = ctx.args
        g = grad_out.contiguous().float()
        dfeat = torch.zeros((B, H, W, Cc), dtype=torch.float32, device=g.device)       # dense NHWC; the kernel adds into it
        _lib.call('fva_roi_align_bwd', _p(g), _p(ctx.rois), ctx.rois.size(0), _p(dfeat), B, H, W, Cc, PH, PW, scale, sampling, _stream())
        return dfeat.permute(0, 3, 1, 2).to(ctx.in_dtype), None, None, None, None


def roi_align(features, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    if aligned:
        raise NotImplementedError('roi_align: aligned=True is not on the reference path (fast.py uses the default)')
    if sampling_ratio > 64:
        raise ValueError('roi_align: sampling_ratio above 64 samples per bin and axis is not supported')
    if isinstance(boxes, (list, tuple)):        # torchvision's list-of-[L,4] form: prepend the image index
        boxes = torch.cat([torch.cat([torch.full((b.size(0), 1), i, dtype=b.dtype, device=b.device), b], 1) for i, b in enumerate(boxes)], 0)
    return RoIAlignFn.apply(features, boxes, output_size, spatial_scale, sampling_ratio)
