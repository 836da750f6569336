#!/usr/bin/env python3
"""Micro-benchmark of the conv entry points through the C ABI:  tools/bench_conv.py B Cin Cout H W k stride [iters]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvision_amd import _lib, ops

def run(B, Cin, Cout, H, W, k, s, iters=20, dtype=torch.bfloat16):
    dev = 'cuda:0'
    lib = _lib.load()
    x = torch.randn(B, H + 2, W + 2, Cin, device=dev).to(dtype)
    x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    dy = torch.randn(B, OH + 2, OW + 2, Cout, device=dev).to(dtype)
    dy[:, 0], dy[:, -1], dy[:, :, 0], dy[:, :, -1] = 0, 0, 0, 0
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, 1, 1)
    wf, wd = ops.packed_weights(w, d, dtype, cache=False)
    M = B * OH * OW
    y = torch.empty(M, Cout, device=dev, dtype=dtype)
    nblk = lib.fva_conv_stat_blocks(C.byref(d))
    stats = torch.empty(nblk, 2, Cout, device=dev)
    dx = torch.empty(B, H, W, Cin, device=dev, dtype=dtype)
    dw = torch.empty_like(w)
    wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = ops._stream()
    calls = {'fwd': lambda: _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), st),
             'dgrad': lambda: _lib.call('fva_conv_dgrad', C.byref(d), ops._p(dy), ops._p(wd), ops._p(dx), C.c_void_p(0), st),
             'wgrad': lambda: _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, st)}
    flop = 2.0 * M * Cout * Cin * k * k
    out = {}
    for name, fn in calls.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out[name] = (ms * 1e3, flop / ms / 1e9)
    return out

if __name__ == '__main__':
    a = [int(v) for v in sys.argv[1:8]]
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
    r = run(*a, iters=iters)
    print(' '.join(map(str, a)), ' | '.join(f'{k}: {v[0]:.1f} us {v[1]:.0f} TF' for k, v in r.items()))
