"""IoU family on the GPU -- API mirror of the reference's detection/tools/IOU.py.

Every function runs the HIP kernels ``fva_iou_pairwise`` / ``fva_iou_batch`` (csrc/loss.hip) with the reference's
quirks kept (eps inside the height factor of the pairwise IoU, DIoU "+" sign, GIoU_batch "+" sign, CIoU alpha
constant).  Pairwise functions are differentiable w.r.t. their FIRST argument.  CPU tensors are refused: this package
has no CPU path.  numpy arrays -- the reference's functions carry a numpy branch each (detection/tools/IOU.py:60-66,130-146;
metrics/map.py feeds it) -- are served by the SAME device kernels: uploaded, computed in fp32 on the GPU, returned as a numpy
array of the input's dtype (INTEGRATION.md notes the one arithmetic difference: the reference's numpy branch of the pairwise
xyxy / xywh IoU has no eps in the height factor, a 1e-7 relative effect).
"""
import ctypes as C

import torch

from ... import _lib
from ...ops import _p, _stream, require_gpu

__all__ = ['cal_iou', 'cal_iou_batch', 'xyxy_iou', 'xywh_iou', 'wh_iou', 'xyxy_iou_batch', 'xywh_iou_batch', 'wh_iou_batch',
           'GIOU', 'GIOU_batch', 'DIOU', 'DIOU_batch', 'CIOU', 'CIOU_batch']

_MODE = {'xyxy': 0, 'xywh': 1, 'wh': 2}
VARIANT = 0     # 0 = library semantics; the demo package sets 1 (demos/yolov3_u/utils/iou.py)


def _prep(t, who):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'fastvision_amd.{who}: expected a CUDA torch.Tensor (numpy/CPU inputs have no path here)')
    require_gpu(t, who)
    return t.detach().float().contiguous()


class _PairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, kind, mode, variant, eps):
        fa, fb = _prep(a, 'iou'), _prep(b, 'iou')
        n = fa.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=fa.device)
        grad = torch.empty_like(fa) if ctx.needs_input_grad[0] else None
        if n:
            _lib.call('fva_iou_pairwise', kind, mode, variant, _p(fa), _p(fb), _p(out), _p(grad), n, eps, _stream())
        ctx.grad_a = grad
        ctx.a_dtype = a.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('fastvision_amd IoU functions are differentiable w.r.t. their first argument only')
        ga = (ctx.grad_a * g.reshape(-1, 1)).to(ctx.a_dtype) if ctx.grad_a is not None else None
        return ga, None, None, None, None, None


def _from_numpy(a, b):
    """(a, b as CUDA tensors, dtype to hand back) when the caller passed numpy arrays (or lists), else (a, b, None)."""
    import numpy as np
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        return a, b, None
    if not torch.cuda.is_available():
        raise RuntimeError('fastvision_amd IoU: numpy inputs are computed on the GPU -- no device is available (there is no CPU path)')
    na = a if isinstance(a, torch.Tensor) else np.asarray(a)
    nb = b if isinstance(b, torch.Tensor) else np.asarray(b)
    back = na.dtype if isinstance(na, np.ndarray) else nb.dtype
    dev = next((t.device for t in (a, b) if isinstance(t, torch.Tensor) and t.is_cuda), torch.device('cuda', torch.cuda.current_device()))
    up = lambda v: v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev)
    return up(na), up(nb), (back if np.issubdtype(back, np.floating) else np.dtype(np.float32))


def _pair(a, b, kind, mode, eps, variant=None):
    a, b, back = _from_numpy(a, b)
    out = _PairFn.apply(a, b, kind, _MODE[mode], VARIANT if variant is None else variant, eps)
    return out if back is None else _NumpyOut(out, back)


class _NumpyOut:
    """Result for numpy callers: converted on the first reshape (every public wrapper reshapes or returns it as is)."""

    def __init__(self, t, dtype):
        self.t, self.dtype = t, dtype

    def reshape(self, *shape):
        return self.t.detach().cpu().numpy().astype(self.dtype).reshape(*shape)


def _batch(a, b, kind, mode, eps, variant=None):
    a, b, back = _from_numpy(a, b)
    out = _batch_dev(a, b, kind, mode, eps, variant)
    return out if back is None else out.cpu().numpy().astype(back)


def _batch_dev(a, b, kind, mode, eps, variant=None):
    fa, fb = _prep(a, 'iou_batch'), _prep(b, 'iou_batch')
    out = torch.empty((fa.shape[0], fb.shape[0]), dtype=torch.float32, device=fa.device)
    _lib.call('fva_iou_batch', kind, _MODE[mode], VARIANT if variant is None else variant, _p(fa), _p(fb), _p(out),
              fa.shape[0], fb.shape[0], eps, _stream())
    return out


def xyxy_iou(xyxy1, xyxy2, eps=1e-7):
    return _pair(xyxy1, xyxy2, 0, 'xyxy', eps).reshape(-1, 1)


def xywh_iou(xywh1, xywh2, eps=1e-7):
    return _pair(xywh1, xywh2, 0, 'xywh', eps).reshape(-1, 1)


def wh_iou(wh1, wh2, eps=1e-7):
    return _pair(wh1, wh2, 0, 'wh', eps).reshape(-1, 1)


def xyxy_iou_batch(xyxy1, xyxy2, eps=1e-7):
    return _batch(xyxy1, xyxy2, 0, 'xyxy', eps)


def xywh_iou_batch(xywh1, xywh2, eps=1e-7):
    return _batch(xywh1, xywh2, 0, 'xywh', eps)


def wh_iou_batch(wh1, wh2, eps=1e-7):
    return _batch(wh1, wh2, 0, 'wh', eps)


def cal_iou(box1, box2, mode='xyxy', eps=1e-7):
    if mode not in _MODE:
        raise Exception('mode must be xyxy or xywh or wh')
    return _pair(box1, box2, 0, mode, eps).reshape(-1, 1)


def cal_iou_batch(box1, box2, mode='xyxy', eps=1e-7):
    if mode not in _MODE:
        raise Exception('mode must be xyxy or xywh or wh')
    return _batch(box1, box2, 0, mode, eps)


def GIOU(box1, box2, mode='xyxy', eps=1e-7):
    return _pair(box1, box2, 1, mode, eps).reshape(-1)          # [N] (the reference returns a flat vector here)


def GIOU_batch(box1, box2, mode='xyxy', eps=1e-7):
    return _batch(box1, box2, 1, mode, eps)


def DIOU(box1, box2, mode='xyxy', eps=1e-7):
    return _pair(box1, box2, 2, mode, eps).reshape(-1, 1)


def DIOU_batch(box1, box2, mode='xyxy', eps=1e-7):
    return _batch(box1, box2, 2, mode, eps)


def CIOU(box1, box2, mode='xyxy', eps=1e-7):
    return _pair(box1, box2, 3, mode, eps).reshape(-1, 1)


def CIOU_batch(box1, box2, mode='xyxy', eps=1e-7):
    return _batch(box1, box2, 3, mode, eps)
