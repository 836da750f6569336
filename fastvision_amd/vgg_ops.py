"""Backbone blocks of the two-stage head on the HIP kernels: the reference's Faster R-CNN backbone is a plain
VGG16 (demos/faster_rcnn/models/vgg.py) -- ``nn.Conv2d(3x3, padding 1, bias) -> nn.ReLU`` blocks and ``nn.MaxPool2d(2, 2)``.

    conv_bias_relu(x, conv)      one Conv2d + ReLU block: implicit-GEMM conv with the bias + ReLU epilogue; backward = ReLU mask
                                 and bias gradient in one pass, then the library's dgrad / wgrad (wgrad on the side stream)
    max_pool2(x)                 MaxPool2d(2, 2)

Activations stay in the package's halo NHWC layout between blocks (the returned tensors are [B,C,H,W] views of it), in the
compute dtype (ops.get_compute_dtype(): bf16 or fp32); parameters and their gradients are fp32.  No CPU path.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib
from .ops import (_code, _grad_like, _p, _stream, get_compute_dtype, halo_alloc, halo_info, packed_weights, require_gpu, to_dense, to_halo,
                  wgrad_stream)

__all__ = ['conv_bias_relu', 'max_pool2']

_CIN_ALIGN = 32        # the implicit-GEMM kernels reduce over whole 64-byte channel slices


class ConvBiasReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dtype):
        require_gpu(x, 'conv_bias_relu')
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        keep, x_ptr, x_pad = to_halo(x.detach(), dtype, 1)
        d = _lib.ConvDesc(_code(dtype), B, H, W, Cin, Cout, 3, 1, x_pad, 1)
        wf, wd = packed_weights(weight, d, dtype, cache=weight.is_leaf)     # the channel-padded first-block filter is a fresh tensor every call
        zbuf, z = halo_alloc(B, Cout, H, W, dtype, x.device, 1)
        b32 = bias.detach().float().contiguous()
        _lib.call('fva_conv_fwd_bias_act', C.byref(d), C.c_void_p(x_ptr), _p(wf), _p(b32), 1, _p(zbuf), 1, _stream())
        ctx.saved = (keep, x_ptr, zbuf, d, wd, dtype, tuple(weight.shape))
        ctx.x_like, ctx.weight = x, weight
        return z

    @staticmethod
    def backward(ctx, dz):
        keep, x_ptr, zbuf, d, wd, dtype, wshape = ctx.saved
        lib = _lib.load()
        B, H, W, Cout, dev, code = d.B, d.H, d.W, d.Cout, zbuf.device, _code(dtype)
        keep_dz, dz_ptr = to_dense(dz, dtype)
        dy = torch.empty((B, H + 2, W + 2, Cout), dtype=dtype, device=dev)
        rows = lib.fva_bias_relu_bwd_rows(B, H, 1)
        part = torch.empty((rows, Cout), dtype=torch.float32, device=dev)
        _lib.call('fva_bias_relu_bwd', code, C.c_void_p(dz_ptr), _p(zbuf), 1, _p(dy), 1, _p(part), B, H, W, Cout, _stream())
        dbias = torch.empty(Cout, dtype=torch.float32, device=dev)
        srows = lib.fva_colsum_scratch_rows(rows)
        scratch = torch.empty((srows, Cout), dtype=torch.float32, device=dev) if srows else None
        _lib.call('fva_colsum', _p(part), rows, Cout, _p(dbias), _p(scratch) if srows else None, _stream())
        dw = torch.empty(wshape, dtype=torch.float32, device=dev)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(x_ptr), _p(dy), _p(dw), 0, _p(ws), wsb, wgrad_stream((keep, dy, ws), ctx.weight))
        dx = None
        if ctx.needs_input_grad[0]:
            dxb = torch.empty((B, H, W, d.Cin), dtype=dtype, device=dev)
            _lib.call('fva_conv_dgrad', C.byref(d), _p(dy), _p(wd), _p(dxb), C.c_void_p(0), _stream())
            dx = _grad_like(dxb, ctx.x_like)
        return dx, dw, dbias, None


class MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        require_gpu(x, 'max_pool2')
        B, Cc, H, W = x.shape
        keep, x_ptr, x_pad = to_halo(x.detach(), dtype, 0)
        obuf, out = halo_alloc(B, Cc, H // 2, W // 2, dtype, x.device, 1)
        _lib.call('fva_maxpool2_fwd', _code(dtype), C.c_void_p(x_ptr), x_pad, _p(obuf), 1, B, H, W, Cc, _stream())
        ctx.saved = (keep, x_ptr, x_pad, (B, Cc, H, W), dtype)
        ctx.x_like = x
        return out

    @staticmethod
    def backward(ctx, dz):
        keep, x_ptr, x_pad, (B, Cc, H, W), dtype = ctx.saved
        keep_dz, dz_ptr = to_dense(dz, dtype)
        dxb = torch.empty((B, H, W, Cc), dtype=dtype, device=dz.device)
        _lib.call('fva_maxpool2_bwd', _code(dtype), C.c_void_p(dz_ptr), C.c_void_p(x_ptr), x_pad, _p(dxb), B, H, W, Cc, _stream())
        return _grad_like(dxb, ctx.x_like), None


def conv_bias_relu(x, conv, dtype=None):
    """``relu(conv(x))`` for an ``nn.Conv2d(Cin, Cout, 3, stride 1, padding 1, bias=True)`` (vgg.py's block)."""
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or conv.bias is None:
        raise RuntimeError('conv_bias_relu: only the VGG block (3x3, stride 1, padding 1, bias) is on this path')
    dtype = dtype or get_compute_dtype()
    weight, cin = conv.weight, conv.weight.shape[1]
    if cin % _CIN_ALIGN:                     # the RGB input of the first block: zero channels (and zero filter taps) up to 32
        extra = _CIN_ALIGN - cin % _CIN_ALIGN
        weight = F.pad(weight, (0, 0, 0, 0, 0, extra))
        if x.requires_grad:
            x = F.pad(x, (0, 0, 0, 0, 0, extra))
        else:                                # an image batch: written straight into a zeroed halo buffer, no padded fp32 copy
            B, _, H, W = x.shape
            buf = torch.zeros((B, H + 2, W + 2, cin + extra), dtype=dtype, device=x.device)
            buf[:, 1:-1, 1:-1, :cin] = x.permute(0, 2, 3, 1)
            x = buf[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)
    return ConvBiasReLUFn.apply(x, weight, conv.bias, dtype)


def max_pool2(x, dtype=None):
    """``nn.MaxPool2d(kernel_size=2, stride=2)``"""
    return MaxPool2Fn.apply(x, dtype or get_compute_dtype())
