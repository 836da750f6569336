#!/usr/bin/env python3
"""HBM rate of the two BatchNorm / SiLU apply passes on the layer shapes of YOLOv3 at B = 32 (bf16), against a device copy of the
same number of bytes.  usage: python tools/bench_bn.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvision_amd import _lib, ops

SHAPES = [(32, 64, 320), (32, 128, 160), (32, 256, 80), (32, 128, 80), (32, 512, 40), (32, 256, 40), (32, 1024, 20), (32, 512, 20)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev, dt = 'cuda:0', torch.bfloat16
    code = ops._code(dt)
    tot = {'fwd': 0.0, 'fwd_res': 0.0, 'bwd': 0.0}
    for B, Cc, H in SHAPES:
        M = B * H * H
        y = torch.randn(M, Cc, device=dev).to(dt)
        dz = torch.randn(M, Cc, device=dev).to(dt)
        res = torch.randn(B, H + 2, H + 2, Cc, device=dev).to(dt)
        z = torch.empty(B, H + 2, H + 2, Cc, device=dev, dtype=dt)
        v = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
        coef = torch.rand(3, Cc, device=dev)
        st = ops._stream()
        dense, halo = M * Cc * 2, B * (H + 2) * (H + 2) * Cc * 2
        f = lambda: _lib.call('fva_bn_silu_apply', code, ops._p(y), ops._p(v[0]), ops._p(v[1]), C.c_void_p(0), 0, ops._p(z), 1, B, H, H, Cc, st)
        fr = lambda: _lib.call('fva_bn_silu_apply', code, ops._p(y), ops._p(v[0]), ops._p(v[1]), ops._p(res), 1, ops._p(z), 1, B, H, H, Cc, st)
        b = lambda: _lib.call('fva_bn_silu_bwd_apply', code, ops._p(dz), ops._p(y), ops._p(v[0]), ops._p(v[1]), ops._p(v[2]), ops._p(v[3]),
                              ops._p(coef), ops._p(z), 1, B, H, H, Cc, st)
        src = torch.empty((dense + halo) // 2, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        cp = lambda: dst.copy_(src)
        tf, tfr, tb, tc = timed(f), timed(fr), timed(b), timed(cp)
        tot['fwd'] += tf; tot['fwd_res'] += tfr; tot['bwd'] += tb
        print(f'C={Cc:4d} @{H:3d}: fwd {tf:6.1f} us {(dense + halo) / tf / 1e6:.2f} TB/s | fwd+res {tfr:6.1f} us {(dense + 2 * halo) / tfr / 1e6:.2f} TB/s | '
              f'bwd {tb:6.1f} us {(2 * dense + halo) / tb / 1e6:.2f} TB/s | copy of {(dense + halo) / 1e6:.0f} MB: {(dense + halo) / tc / 1e6:.2f} TB/s', flush=True)
    print('sum us', {k: round(v, 1) for k, v in tot.items()})


if __name__ == '__main__':
    main()
