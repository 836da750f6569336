#!/usr/bin/env python3
"""The apply rider (fva_conv_wgrad_ride): a BatchNorm-backward apply pass of one layer carried by the weight-gradient launch of
another.  For every (host layer, job) pair of YOLOv3's backward pass at B = 32 this checks that the ridden pass writes the SAME
BITS as fva_bn_silu_bwd_apply and that the host's dW equals the plain launch's bit for bit, and times: host alone, pass alone,
the two back to back (today's chain), host + rider.   usage: python tools/ride_check.py [small]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvision_amd import _lib, ops

# (host: B, Cin, Cout, H, stride), (job: C, H)
PAIRS = [((32, 128, 256, 80, 1), (256, 80)), ((32, 128, 256, 80, 1), (128, 80)), ((32, 256, 512, 40, 1), (512, 40)),
         ((32, 256, 512, 40, 1), (256, 40)), ((32, 512, 1024, 20, 1), (1024, 20)), ((32, 512, 1024, 20, 1), (512, 20)),
         ((32, 128, 256, 80, 1), (64, 160)), ((32, 256, 512, 80, 2), (128, 80))]
SMALL = [((14, 256, 512, 40, 1), (256, 24)), ((6, 512, 1024, 33, 1), (64, 37)), ((16, 256, 512, 80, 2), (1024, 7))]


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
    lib = _lib.load()
    bad = 0
    pairs = SMALL if 'small' in sys.argv[1:] else PAIRS
    for (B, Cin, Cout, H, s), (Cj, Hj) in pairs:
        g = torch.Generator().manual_seed(Cin + Hj)
        x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(dev).to(dt)
        OH = (H - 1) // s + 1
        dyh = torch.randn(B, OH + 2, OH + 2, Cout, generator=g).to(dev).to(dt)
        for t in (x, dyh):
            t[:, 0], t[:, -1], t[:, :, 0], t[:, :, -1] = 0, 0, 0, 0
        d = _lib.ConvDesc(code, B, H, H, Cin, Cout, 3, s, 1, 1)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        dw0 = torch.empty(Cout, Cin, 3, 3, device=dev)
        dw1 = torch.empty_like(dw0)
        cap = lib.fva_conv_wgrad_ride_capacity(C.byref(d), Cj)
        # the job
        Mj = B * Hj * Hj
        y = torch.randn(Mj, Cj, generator=g).to(dev).to(dt)
        dz = torch.randn(Mj, Cj, generator=g).to(dev).to(dt)
        sc, sh, mu, rs = [(torch.rand(Cj, generator=g) + 0.5).to(dev) for _ in range(4)]
        coef = torch.rand(3, Cj, generator=g).to(dev)
        coef[0] = sc                       # what fva_bn_bwd_finalize writes: coef[0] = gamma * rstd = the forward scale
        ref = torch.empty(B, Hj + 2, Hj + 2, Cj, device=dev, dtype=dt)
        out = torch.full_like(ref, float('nan'))
        total = B * (Hj + 2) * (Hj + 2) * (Cj // 8)
        st = ops._stream()
        job = _lib.BnBwdJob(dz.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), coef.data_ptr(),
                            out.data_ptr(), 1, B, Hj, Hj, Cj, 0, min(total, cap))
        rest = _lib.BnBwdJob(dz.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), coef.data_ptr(),
                             out.data_ptr(), 1, B, Hj, Hj, Cj, min(total, cap), total)
        f_host = lambda: _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dyh), ops._p(dw0), 0, ops._p(ws), wsb, st)
        f_apply = lambda: _lib.call('fva_bn_silu_bwd_apply', code, ops._p(dz), ops._p(y), ops._p(sc), ops._p(sh), ops._p(mu), ops._p(rs),
                                    ops._p(coef), ops._p(ref), 1, B, Hj, Hj, Cj, st)

        def f_ride():
            _lib.call('fva_conv_wgrad_ride', C.byref(d), ops._p(x), ops._p(dyh), ops._p(dw1), 0, ops._p(ws), wsb, C.byref(job), st)
            if rest.chunk_end > rest.chunk_begin:
                _lib.call('fva_bn_silu_bwd_apply_range', code, C.byref(rest), st)

        def f_chain():
            f_apply()
            f_host()
        f_host(); f_apply(); f_ride()
        torch.cuda.synchronize()
        same_dy = torch.equal(out.view(torch.int16), ref.view(torch.int16))
        same_dw = torch.equal(dw0, dw1)
        if not (same_dy and same_dw):
            bad += 1
            nd = (out.view(torch.int16) != ref.view(torch.int16)).sum().item()
            print(f'   MISMATCH: dY differs in {nd} of {ref.numel()} elements (nan: {torch.isnan(out.float()).sum().item()}), dW equal: {same_dw}')
        th, ta, tc, tr = timed(f_host), timed(f_apply), timed(f_chain), timed(f_ride)
        # run to run
        f_ride(); torch.cuda.synchronize()
        again = torch.equal(out.view(torch.int16), ref.view(torch.int16))
        print(f'host {Cin}->{Cout}@{OH}^2 s{s} + job C={Cj}@{Hj}^2 ({total / max(cap, 1):.2f} of capacity): host {th:6.1f} us, pass {ta:5.1f}, '
              f'chain {tc:6.1f}, ridden {tr:6.1f} us  ({tr - th:+.1f} over the host, {tc - tr:+.1f} saved); dY bits equal: {same_dy and again}, dW equal: {same_dw}', flush=True)
    print('ride_check:', 'all equal' if not bad else f'{bad} mismatching pairs')
    return bad


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
