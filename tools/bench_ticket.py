#!/usr/bin/env python3
"""Tail cost of the BatchNorm finalisation, per layer shape at B = 32 (interleaved A/B in one process):
   A: fva_conv_fwd + fva_bn_finalize + apply      B: fva_conv_fwd_bn (tickets, csrc/bn_ticket.h) + apply      C: conv + apply only
and the same for the data gradient with the fused statistics.  The apply pass is in the chain because it is what waits for the
coefficients (back-to-back launches on one stream serialise, so the fold's latency shows).  tools/bench_ticket.py [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvision_amd import _lib, ops

SHAPES = [(32, 128, 256, 80, 80, 3, 1), (32, 256, 128, 80, 80, 1, 1), (32, 256, 512, 40, 40, 3, 1), (32, 512, 256, 40, 40, 1, 1),
          (32, 512, 1024, 20, 20, 3, 1), (32, 1024, 512, 20, 20, 1, 1), (32, 64, 128, 160, 160, 3, 1), (32, 128, 64, 160, 160, 1, 1),
          (32, 32, 64, 320, 320, 3, 1), (32, 64, 32, 320, 320, 1, 1)]


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main(rounds=5, iters=10):
    dev, dtype = 'cuda:0', torch.bfloat16
    lib = _lib.load()
    st = ops._stream()
    for (B, Cin, Cout, H, W, k, s) in SHAPES:
        OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
        M = B * OH * OW
        x = torch.randn(B, H + 2, W + 2, Cin, device=dev).to(dtype)
        w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
        d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, 1, 1)
        wf, wd = ops.packed_weights(w, d, dtype, cache=False)
        y = torch.empty(M, Cout, device=dev, dtype=dtype)
        z = torch.empty(B, OH + 2, OW + 2, Cout, device=dev, dtype=dtype)
        nblk = lib.fva_conv_stat_blocks(C.byref(d))
        stats = torch.empty(lib.fva_bn_partial_rows(nblk), 2, Cout, device=dev)
        gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        rm, rv, nbt = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        mean, rstd, scale, shift = (torch.ones(Cout, device=dev) for _ in range(4))
        cnt = torch.zeros(lib.fva_bn_ticket_counters(nblk, Cout), dtype=torch.int32, device=dev)
        gs = torch.empty(lib.fva_bn_ticket_groups(nblk), 2, Cout, dtype=torch.float64, device=dev)
        fin = _lib.BnFwdFin(cnt.data_ptr(), gs.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5,
                            mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
        code = ops._code(dtype)

        def apply():
            _lib.call('fva_bn_silu_apply', code, ops._p(y), ops._p(scale), ops._p(shift), C.c_void_p(0), 0, ops._p(z), 1, B, OH, OW, Cout, st)

        def a():
            _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), st)
            _lib.call('fva_bn_finalize', ops._p(stats), nblk, stats.shape[0], M, Cout, ops._p(gamma), ops._p(beta), ops._p(rm), ops._p(rv), ops._p(nbt), 0.1, 1e-5,
                      ops._p(mean), ops._p(rstd), ops._p(scale), ops._p(shift), st)
            apply()

        def b():
            _lib.call('fva_conv_fwd_bn', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), C.byref(fin), st)
            apply()

        def c():
            _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), st)
            apply()
        # backward: dgrad of THIS layer with the statistics of a producer of its input (Cin channels)
        dy = torch.randn(B, OH + 2, OW + 2, Cout, device=dev).to(dtype)
        dx = torch.empty(B, H, W, Cin, device=dev, dtype=dtype)
        yp = torch.randn(B * H * W, Cin, device=dev).to(dtype)
        gP = torch.ones(Cin, device=dev)
        v4 = [torch.ones(Cin, device=dev) for _ in range(4)]
        rows = lib.fva_conv_dgrad_stat_rows(C.byref(d))
        part = torch.empty(lib.fva_bn_partial_rows(rows), 2, Cin, device=dev)
        fs = _lib.BnBwdFuse(yp.data_ptr(), v4[0].data_ptr(), v4[1].data_ptr(), v4[2].data_ptr(), v4[3].data_ptr(), part.data_ptr())
        dg, db, coef = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev), torch.empty(3, Cin, device=dev)
        cntb = torch.zeros(lib.fva_bn_ticket_counters(rows, Cin), dtype=torch.int32, device=dev)
        gsb = torch.empty(lib.fva_bn_ticket_groups(rows), 2, Cin, dtype=torch.float64, device=dev)
        finb = _lib.BnBwdFin(cntb.data_ptr(), gsb.data_ptr(), gP.data_ptr(), dg.data_ptr(), db.data_ptr(), coef.data_ptr(), 0)
        dyp = torch.empty(B, H + 2, W + 2, Cin, device=dev, dtype=dtype)

        def bapply():
            _lib.call('fva_bn_silu_bwd_apply', code, ops._p(dx), ops._p(yp), ops._p(v4[0]), ops._p(v4[1]), ops._p(v4[2]), ops._p(v4[3]), ops._p(coef),
                      ops._p(dyp), 1, B, H, W, Cin, st)

        def ba():
            _lib.call('fva_conv_dgrad_bnstats', C.byref(d), ops._p(dy), ops._p(wd), ops._p(dx), C.c_void_p(0), C.byref(fs), st)
            _lib.call('fva_bn_bwd_finalize', ops._p(part), rows, part.shape[0], B * H * W, Cin, ops._p(gP), ops._p(v4[3]), ops._p(dg), ops._p(db), 0, ops._p(coef), st)
            bapply()

        def bb():
            _lib.call('fva_conv_dgrad_bn', C.byref(d), ops._p(dy), ops._p(wd), ops._p(dx), C.c_void_p(0), C.byref(fs), C.byref(finb), st)
            bapply()

        def bc():
            _lib.call('fva_conv_dgrad_bnstats', C.byref(d), ops._p(dy), ops._p(wd), ops._p(dx), C.c_void_p(0), C.byref(fs), st)
            bapply()
        res = {n: [] for n in ('a', 'b', 'c', 'ba', 'bb', 'bc')}
        fns = {'a': a, 'b': b, 'c': c, 'ba': ba, 'bb': bb, 'bc': bc}
        for f in fns.values():
            f()
        for _ in range(rounds):
            for n, f in fns.items():
                res[n].append(timed(f, iters))
        med = {n: sorted(v)[len(v) // 2] for n, v in res.items()}
        print(f'{Cin:5d}->{Cout:5d} k{k} @{H:3d} rows {nblk:5d}/{rows:5d} | fwd: launch+finalize {med["a"]:7.1f}  ticket {med["b"]:7.1f}  none {med["c"]:7.1f} us'
              f' | dgrad: {med["ba"]:7.1f}  {med["bb"]:7.1f}  {med["bc"]:7.1f} us', flush=True)


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
