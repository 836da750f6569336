#!/usr/bin/env python3
"""Forward convolution launches with the statistics stored in the table against added to the accumulator (isolated launches, back to back):
tools/sweep_acc.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from fastvision_amd import _lib, ops
from sweep_conv import LAYERS

dev, dt = 'cuda:0', torch.bfloat16
lib = _lib.load()
tot = [0.0, 0.0]
for n, ci, co, h, k, s in LAYERS:
    B = 32
    x = torch.randn(B, h + 2, h + 2, ci, device=dev).to(dt)
    oh = (h - 1) // s + 1
    w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
    d = _lib.ConvDesc(ops._code(dt), B, h, h, ci, co, k, s, 1, 1)
    wf, _ = ops.packed_weights(w, d, dt, cache=False)
    M = B * oh * oh
    y = torch.empty(M, co, device=dev, dtype=dt)
    nblk = lib.fva_conv_stat_blocks(C.byref(d))
    stats = torch.empty(lib.fva_bn_partial_rows(nblk), 2, co, device=dev)
    R = ops._replicas(M)
    acc = torch.zeros(R * 5 * co, dtype=torch.int64, device=dev)
    st = ops._stream()
    fns = [lambda: _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), st),
           lambda: _lib.call('fva_conv_fwd_acc', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(acc), R, st)]
    us = []
    for fn in fns:
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 100)
    tot[0] += n * us[0]; tot[1] += n * us[1]
    print(f'{n:>2} {ci:>5} {co:>5} {h:>4} {k} {s} | tiles(stat rows) {nblk:>6} replicas {R:>2} | table {us[0]:8.1f} us | accumulator {us[1]:8.1f} us | {us[1] - us[0]:+7.1f}', flush=True)
print('total per step: table %.3f ms, accumulator %.3f ms' % (tot[0] / 1e3, tot[1] / 1e3))
