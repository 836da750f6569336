#!/usr/bin/env python3
"""Where does a block of the 8-phase convolution kernel spend its time?  Per-block wall_clock64 stamps (100 MHz) written by
the kernel itself (fva_conv_debug_stamps): entry -> first k-tile ready (prologue) -> k-loop done -> exit (epilogue).
usage: python tools/tile_timing.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastvision_amd import _lib, ops

SHAPES = [(32, 128, 256, 80, 3), (32, 256, 512, 40, 3), (32, 512, 1024, 20, 3), (32, 512, 256, 40, 3)]      # (B, Cin, Cout, H, k); the last one is the 256->512 dgrad's shape run as a forward


def main():
    lib = _lib.load()
    dev, dtype = 'cuda:0', torch.bfloat16
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for (B, Cin, Cout, H, k) in SHAPES:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(dev).to(dtype)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
        d = _lib.ConvDesc(ops._code(dtype), B, H, H, Cin, Cout, k, 1, 1, 1)
        wf, wd = ops.packed_weights(w, d, dtype, cache=False)
        M = B * H * H
        y = torch.empty(M, Cout, device=dev, dtype=dtype)
        nblk = lib.fva_conv_stat_blocks(C.byref(d))
        stats = torch.zeros(lib.fva_bn_partial_rows(nblk), 2, Cout, device=dev)
        st = ops._stream()
        for with_stats in (True, False):
            sp = ops._p(stats) if with_stats else C.c_void_p(0)
            fwd = lambda: _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), sp, st)
            for _ in range(3):
                fwd()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fwd()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            stamps.zero_()
            _lib.call('fva_conv_debug_stamps', ops._p(stamps), 4096)
            fwd()
            torch.cuda.synchronize()
            _lib.call('fva_conv_debug_stamps', C.c_void_p(0), 0)
            raw = stamps.cpu().numpy().reshape(-1, 8)
            raw = raw[raw[:, 3] > 0]
            s, cyc = raw[:, :4], raw[:, 4:]
            mhz = np.median((cyc[:, 2] - cyc[:, 1]) / ((s[:, 2] - s[:, 1]) / 100.0))
            t0 = s[:, 0].min()
            rel = (s - t0) / 100.0                                   # microseconds since the first block started
            pro, loop, epi = rel[:, 1] - rel[:, 0], rel[:, 2] - rel[:, 1], rel[:, 3] - rel[:, 2]
            first = rel[:, 0] < 5.0                                  # blocks of the first round
            tiles, kt = len(s), k * k * Cin // 64
            print(f'{Cin}->{Cout} @{H} k{k} stats={int(with_stats)}: {tiles} tiles x {kt} k-tiles, launch {us:.1f} us, shader clock in the k-loop {mhz:.0f} MHz | '
                  f'last exit {rel[:, 3].max():.1f} us | prologue {np.median(pro):.2f} loop {np.median(loop):.2f} '
                  f'({np.median(loop) / kt * 1e3:.0f} ns/k-tile) epilogue {np.median(epi):.2f} us (medians) | '
                  f'first-round blocks: {first.sum()}, start spread {rel[first, 0].max():.2f} us; '
                  f'second-round starts {np.sort(rel[~first, 0])[:3].round(1).tolist() if (~first).any() else "-"}')


def wgrad():
    lib = _lib.load()
    dev, dtype = 'cuda:0', torch.bfloat16
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for (B, Cin, Cout, H, k) in [(32, 128, 256, 80, 3), (32, 256, 512, 40, 3), (32, 512, 1024, 20, 3)]:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(dev).to(dtype)
        dy = torch.randn(B, H + 2, H + 2, Cout, generator=g).to(dev).to(dtype)
        d = _lib.ConvDesc(ops._code(dtype), B, H, H, Cin, Cout, k, 1, 1, 1)
        dw = torch.empty(Cout, Cin, k, k, device=dev)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st = ops._stream()
        run = lambda: _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, st)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        stamps.zero_()
        _lib.call('fva_conv_debug_stamps', ops._p(stamps), 4096)
        run()
        torch.cuda.synchronize()
        _lib.call('fva_conv_debug_stamps', C.c_void_p(0), 0)
        raw = stamps.cpu().numpy().reshape(-1, 8)
        raw = raw[raw[:, 3] > 0]
        s, cyc = raw[:, :4], raw[:, 4:]
        rel = (s - s[:, 0].min()) / 100.0
        pro, loop, epi = rel[:, 1] - rel[:, 0], rel[:, 2] - rel[:, 1], rel[:, 3] - rel[:, 2]
        mhz = np.median((cyc[:, 2] - cyc[:, 1]) / np.maximum(s[:, 2] - s[:, 1], 1) * 100.0)
        print(f'wgrad {Cin}->{Cout} @{H}: {len(s)} blocks, call {us:.1f} us (incl. reduce), slab {wsb / 1e6:.0f} MB, clock {mhz:.0f} MHz | '
              f'last exit {rel[:, 3].max():.1f} us | prologue {np.median(pro):.2f} loop {np.median(loop):.2f} epilogue {np.median(epi):.2f} '
              f'(max {epi.max():.2f}) us | start spread {rel[:, 0].max():.2f} us')


def pw():
    """1x1 layers on igemm_kernel (FVA_STAMP_IGEMM=1): forward with statistics; dgrad plain, with addend, with addend + fused
    BatchNorm-backward statistics.  Phases per block: setup (pointers, prefetch issue) / first k-tile wait / rest of the k loop /
    accumulators -> LDS (+ forward statistics) / store loop."""
    lib = _lib.load()
    dev, dtype = 'cuda:0', torch.bfloat16
    NS = 16384
    stamps = torch.zeros(NS * 8, dtype=torch.int64, device=dev)
    shapes = [(32, 256, 128, 80, 1), (32, 512, 256, 40, 1), (32, 1024, 512, 20, 1), (32, 128, 64, 160, 1)]
    if len(sys.argv) > 2 and sys.argv[2] == 'thin':
        shapes = [(32, 32, 64, 320, 3), (32, 64, 128, 160, 3), (32, 64, 32, 320, 1), (32, 128, 256, 80, 3)]
    for (B, Cin, Cout, H, ks) in shapes:
        g = torch.Generator().manual_seed(0)
        M = B * H * H
        x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(dev).to(dtype)
        dyh = torch.randn(B, H + 2, H + 2, Cout, generator=g).to(dev).to(dtype)
        w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).to(dev)
        d = _lib.ConvDesc(ops._code(dtype), B, H, H, Cin, Cout, ks, 1, 1, 1)
        wf, wd = ops.packed_weights(w, d, dtype, cache=False)
        y = torch.empty(M, Cout, device=dev, dtype=dtype)
        nblk = lib.fva_conv_stat_blocks(C.byref(d))
        stats = torch.zeros(lib.fva_bn_partial_rows(nblk), 2, Cout, device=dev)
        dx = torch.empty(M, Cin, device=dev, dtype=dtype)
        add = torch.randn(M, Cin, generator=g).to(dev).to(dtype)
        yprod = torch.randn(M, Cin, generator=g).to(dev).to(dtype)
        coef = [torch.rand(Cin, device=dev) + 0.5 for _ in range(4)]
        rows = lib.fva_conv_dgrad_stat_rows(C.byref(d))
        part = torch.empty(lib.fva_bn_partial_rows(rows), 2, Cin, device=dev)
        fs = _lib.BnBwdFuse(yprod.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr())
        st = ops._stream()
        cases = [
            ('fwd+stats', (Cin + Cout) * 2, lambda: _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), st)),
            ('dgrad plain', (Cin + Cout) * 2, lambda: _lib.call('fva_conv_dgrad', C.byref(d), ops._p(dyh), ops._p(wd), ops._p(dx), C.c_void_p(0), st)),
            ('dgrad +addend', (2 * Cin + Cout) * 2, lambda: _lib.call('fva_conv_dgrad', C.byref(d), ops._p(dyh), ops._p(wd), ops._p(dx), ops._p(add), st)),
            ('dgrad +addend +bnb', (3 * Cin + Cout) * 2, lambda: _lib.call('fva_conv_dgrad_bnstats', C.byref(d), ops._p(dyh), ops._p(wd), ops._p(dx), ops._p(add), C.byref(fs), st)),
        ]
        for name, bpp, run in cases:
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            outbuf = y if name.startswith('fwd') else dx
            chk = f'checksum {outbuf.float().sum().item():.6e} {outbuf.float().abs().sum().item():.6e}' + (f' part {part[:rows].double().sum().item():.9e}' if 'bnb' in name else '')
            stamps.zero_()
            _lib.call('fva_conv_debug_stamps', ops._p(stamps), NS)
            run()
            torch.cuda.synchronize()
            _lib.call('fva_conv_debug_stamps', C.c_void_p(0), 0)
            s8 = stamps.cpu().numpy().reshape(-1, 8)
            s8 = s8[s8[:, 5] > 0]
            s = s8[:, :6]
            if not len(s):
                print(f'{Cin}->{Cout} @{H} {name}: launch {us:.1f} us = {M * bpp / us / 1e6:.2f} TB/s algorithmic, {2.0 * M * Cin * Cout * ks * ks / us / 1e6:.0f} TF | {chk}', flush=True)
                continue
            rel = (s - s[:, 0].min()) / 100.0
            ph = np.diff(rel, axis=1)
            med = np.median(ph, axis=0)
            starts = np.sort(rel[:, 0])
            print(f'{Cin}->{Cout} k{ks} @{H} {name}: launch {us:.1f} us = {M * bpp / us / 1e6:.2f} TB/s algorithmic, {2.0 * M * Cin * Cout * ks * ks / us / 1e6:.0f} TF | {len(s)} blocks stamped, last exit {rel[:, 5].max():.1f} us | '
                  f'medians: setup {med[0]:.2f} first-tile wait {med[1]:.2f} k-loop {med[2]:.2f} to-LDS {med[3]:.2f} store {med[4]:.2f} = {np.median(rel[:, 5] - rel[:, 0]):.2f} us/block | '
                  f'setup split: rows+ktab {np.median(s8[:, 6] - s8[:, 0]) / 100:.2f} operand requests {np.median(s8[:, 7] - s8[:, 6]) / 100:.2f} DMA issue {np.median(s8[:, 1] - s8[:, 7]) / 100:.2f} | '
                  f'block starts at 25/50/75 %: {starts[len(starts) // 4]:.1f} {starts[len(starts) // 2]:.1f} {starts[3 * len(starts) // 4]:.1f}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pw':
        pw()
    else:
        main()
        wgrad()
