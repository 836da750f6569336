"""Fully connected layers and the small losses of the two-stage head on the library's kernels.

The reference's Fast head runs RoI features through the VGG classifier ``Linear(25088, 4096) -> ReLU -> Dropout -> Linear(4096, 4096)
-> ReLU -> Dropout`` and two output layers (demos/faster_rcnn/models/vgg.py:41-47, fast.py:47-52), then ``F.cross_entropy`` /
``F.smooth_l1_loss`` (fast.py:192,197); the RPN uses its FocalLoss and ``F.smooth_l1_loss`` (rpn.py:8-64,303-312).

    linear_relu(x, lin)        relu(x @ W^T + b): a 1x1 convolution over R "pixels" on the implicit-GEMM kernel with the bias + ReLU
                               epilogue (fva_conv_fwd_bias_act); backward: ReLU mask + bias gradient in one pass
                               (fva_rows_relu_bwd + fva_colsum), then fva_conv_dgrad / fva_conv_wgrad
    linear(x, lin)             x @ W^T + b without activation, fp32 output: the detection head's biased 1x1 convolution (ops.HeadFn)
    cross_entropy_mean / focal_mean / smooth_l1_mean      value and gradient in one launch (fva_row_loss / fva_smooth_l1)

Rows are the RoIs (a few hundred), so these GEMMs stream their weights once: HBM-bound on the 205 MB bf16 copy of the first layer.
Activations in the compute dtype (bf16 or fp32), parameters and their gradients fp32.  No CPU path.
"""
import ctypes as C

import torch

from . import _lib
from .ops import HeadFn, _code, _p, _stream, get_compute_dtype, packed_weights, require_gpu

__all__ = ['linear_relu', 'linear', 'cross_entropy_mean', 'focal_mean', 'smooth_l1_mean']


def _packed_linear(weight, d, dtype):
    """(w_fwd, w_dgrad) of an nn.Linear weight [N, K] seen as a 1x1 filter; cached on the parameter until it changes and re-packed
    with every other layer in the one launch of the registry (the 103 M weights of the first VGG classifier layer took 1.1 ms per
    step in the single-layer pack kernel)."""
    if not weight.is_contiguous():
        raise RuntimeError('linear: the weight must be contiguous')
    return packed_weights(weight.detach().view(weight.shape[0], weight.shape[1], 1, 1), d, dtype, owner=weight)


class LinearReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dtype):
        require_gpu(x, 'linear_relu')
        R, K = x.shape
        N = weight.shape[0]
        xs = x.detach()
        xs = xs if (xs.dtype == dtype and xs.is_contiguous()) else xs.to(dtype).contiguous()
        d = _lib.ConvDesc(_code(dtype), 1, R, 1, K, N, 1, 1, 0, 1)            # x: [1][R][1][K] NHWC without halo; dY gets a border of 1
        wf, wd = _packed_linear(weight, d, dtype)
        z = torch.empty((R, N), dtype=dtype, device=x.device)
        _lib.call('fva_conv_fwd_bias_act', C.byref(d), _p(xs), _p(wf), _p(bias.detach().float().contiguous()), 1, _p(z), 0, _stream())
        ctx.saved = (xs, z, d, wd, dtype, tuple(weight.shape))
        ctx.x_dtype, ctx.weight = x.dtype, weight
        return z

    @staticmethod
    def backward(ctx, dz):
        xs, z, d, wd, dtype, wshape = ctx.saved
        lib = _lib.load()
        R, N, dev, code = d.H, d.Cout, z.device, _code(dtype)
        g = dz if (dz.dtype == dtype and dz.is_contiguous()) else dz.to(dtype).contiguous()
        dy = torch.zeros((1, R + 2, 3, N), dtype=dtype, device=dev)           # halo buffer (W = 1): the kernels want a zero border
        rows = lib.fva_rows_relu_bwd_rows(R)
        part = torch.empty((rows, N), dtype=torch.float32, device=dev)
        _lib.call('fva_rows_relu_bwd', code, _p(g), _p(z), C.c_void_p(dy[0, 1, 1].data_ptr()), 3 * N, _p(part), R, N, 1, _stream())
        dbias = torch.empty(N, dtype=torch.float32, device=dev)
        srows = lib.fva_colsum_scratch_rows(rows)
        scratch = torch.empty((srows, N), dtype=torch.float32, device=dev) if srows else None
        _lib.call('fva_colsum', _p(part), rows, N, _p(dbias), _p(scratch) if srows else None, _stream())
        dw = torch.empty((wshape[0], wshape[1], 1, 1), dtype=torch.float32, device=dev)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        # on the launch stream, not the side stream: a fully connected layer may be applied several times in one forward pass (the
        # reference runs positives and negatives through the head separately), and autograd then SUMS the weight gradients of the
        # calls on the launch stream -- it must find them finished
        _lib.call('fva_conv_wgrad', C.byref(d), _p(xs), _p(dy), _p(dw), 0, _p(ws), wsb, _stream())
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, d.Cin), dtype=dtype, device=dev)
            _lib.call('fva_conv_dgrad', C.byref(d), _p(dy), _p(wd), _p(dx), C.c_void_p(0), _stream())
            dx = dx if dx.dtype == ctx.x_dtype else dx.to(ctx.x_dtype)
        return dx, dw.view(wshape), dbias, None


def linear_relu(x, lin, dtype=None):
    """``relu(lin(x))`` for an ``nn.Linear`` with bias; x [R, in_features]."""
    dtype = dtype or get_compute_dtype()
    bk = 64 if dtype == torch.bfloat16 else 32
    if lin.bias is None or lin.in_features % bk or lin.out_features % 8:
        raise RuntimeError(f'linear_relu: in_features must be a multiple of {bk}, out_features of 8, with a bias')
    return LinearReLUFn.apply(x, lin.weight, lin.bias, dtype)


def linear(x, lin, dtype=None):
    """``lin(x)`` (no activation), fp32 result [R, out_features] -- the biased 1x1 head convolution over R rows."""
    dtype = dtype or get_compute_dtype()
    R, K = x.shape
    xs = x if (x.dtype == dtype and x.is_contiguous()) else x.to(dtype).contiguous()
    x4 = xs.view(1, R, 1, K).permute(0, 3, 1, 2)                              # logical [1, K, R, 1] view of the NHWC rows: zero-copy
    out = HeadFn.apply(x4, lin.weight.view(lin.out_features, K, 1, 1), lin.bias, dtype)
    return out.view(R, lin.out_features)


class _RowLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, mode, gamma):
        require_gpu(logits, 'row loss')
        z = logits.detach().float().contiguous()
        R, Cc = z.shape
        grad = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        out = torch.empty(1, dtype=torch.float32, device=z.device)
        ws = torch.empty(R, dtype=torch.float32, device=z.device)
        _lib.call('fva_row_loss', _p(z), _p(labels.detach().long().contiguous()), R, Cc, mode, float(gamma), _p(out), _p(grad), _p(ws), _stream())
        ctx.grad, ctx.dtype = grad, logits.dtype
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad * gout
        return (g if g.dtype == ctx.dtype else g.to(ctx.dtype)), None, None, None


class _SmoothL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        require_gpu(pred, 'smooth_l1')
        a, b = pred.detach().float().contiguous(), target.detach().float().contiguous()
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        ws = torch.empty(1024, dtype=torch.float32, device=a.device)
        _lib.call('fva_smooth_l1', _p(a), _p(b), a.numel(), _p(out), _p(grad), _p(ws), _stream())
        ctx.grad, ctx.dtype, ctx.shape = grad, pred.dtype, pred.shape
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        g = (ctx.grad * gout).view(ctx.shape)
        return (g if g.dtype == ctx.dtype else g.to(ctx.dtype)), None


def cross_entropy_mean(logits, labels):
    """F.cross_entropy(logits [R, C], labels [R], reduction='mean')"""
    return _RowLossFn.apply(logits, labels, 0, 0.0)


def focal_mean(logits, labels, gamma=2.0):
    """mean over rows of -(1 - p_t)^gamma * log p_t, p = softmax(logits) (the RPN's FocalLoss with alpha = 1, rpn.py:8-64)"""
    return _RowLossFn.apply(logits, labels, 1, gamma)


def smooth_l1_mean(pred, target):
    """F.smooth_l1_loss(pred, target, reduction='mean') (beta = 1)"""
    if pred.numel() == 0:
        return pred.sum() * 0.0
    return _SmoothL1Fn.apply(pred, target)
