#!/usr/bin/env python3
"""Weight-gradient launches (split-K kernel + reduce) of YOLOv3's layer shapes at B = 32, bf16: us per call and TFLOP/s.
usage: tools/bench_wgrad.py [iters]   (FVA_LIB_PATH selects another build for A/Bs)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvision_amd import _lib, ops

# (Cin, Cout, H, k, stride, launches per step)
SHAPES = [(128, 256, 80, 3, 1, 11), (256, 512, 40, 3, 1, 11), (512, 1024, 20, 3, 1, 7), (256, 128, 80, 1, 1, 10), (512, 256, 40, 1, 1, 11),
          (1024, 512, 20, 1, 1, 7), (32, 64, 320, 3, 1, 1), (32, 64, 640, 3, 2, 1), (64, 128, 160, 3, 1, 2), (64, 128, 320, 3, 2, 1),
          (128, 64, 160, 1, 1, 2), (64, 32, 320, 1, 1, 1), (128, 256, 160, 3, 2, 1), (256, 512, 80, 3, 2, 1), (512, 1024, 40, 3, 2, 1),
          (384, 128, 80, 1, 1, 1), (768, 256, 40, 1, 1, 1), (256, 256, 80, 1, 1, 1)]


def main(iters=10):
    dev, dtype, B = 'cuda:0', torch.bfloat16, 32
    lib = _lib.load()
    st = ops._stream()
    total = 0.0
    for Cin, Cout, H, k, s, n in SHAPES:
        OH = (H - 1) // s + 1
        x = torch.randn(B, H + 2, H + 2, Cin, device=dev).to(dtype)
        dy = torch.randn(B, OH + 2, OH + 2, Cout, device=dev).to(dtype)
        d = _lib.ConvDesc(ops._code(dtype), B, H, H, Cin, Cout, k, s, 1, 1)
        dw = torch.empty(Cout, Cin, k, k, device=dev)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        fn = lambda: _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, st)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        flop = 2.0 * B * OH * OH * Cout * Cin * k * k
        total += us * n
        print(f'{Cin:5d}->{Cout:5d} k{k} s{s} @{H:3d} x{n:2d}: {us:7.1f} us {flop / us / 1e6:6.0f} TF  slabs {wsb / 1e6:6.1f} MB', flush=True)
    print(f'weighted sum {total / 1e3:.3f} ms per step (listed layers)')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
