#!/usr/bin/env python3
"""A/B of the 8-phase igemm kernel against the default kernels: run once per setting of FVA_IGEMM8, dump strided samples
and checksums of y / stats / dx, then compare.   tools/check_igemm8.py run <out.npz> | compare <a.npz> <b.npz>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if os.environ.get('CHECK_SHAPES') == 'small':     # the test suite's subset: every code path, seconds to run
    SHAPES = [(32, 256, 512, 40, 3, 1), (9, 512, 1024, 40, 3, 1), (32, 512, 256, 40, 1, 1), (33, 256, 512, 80, 3, 2)]
else:
  SHAPES = [(64, 256, 512, 32, 3, 1), (32, 128, 256, 80, 3, 1), (32, 256, 512, 40, 3, 1), (32, 512, 1024, 20, 3, 1), (32, 512, 256, 40, 1, 1),
          (32, 256, 512, 80, 3, 2), (33, 256, 512, 40, 3, 1), (41, 256, 512, 40, 3, 1)]


def run(path):
    import torch
    from fastvision_amd import _lib, ops
    lib = _lib.load()
    out = {}
    dev = 'cuda:0'
    dtype = torch.bfloat16
    for si, (B, Cin, Cout, H, k, s) in enumerate(SHAPES):
        g = torch.Generator().manual_seed(si)
        W = H
        x = torch.randn(B, H + 2, W + 2, Cin, generator=g).to(dev).to(dtype)
        x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0
        OH = (H - 1) // s + 1
        dy = torch.randn(B, OH + 2, OH + 2, Cout, generator=g).to(dev).to(dtype)
        dy[:, 0], dy[:, -1], dy[:, :, 0], dy[:, :, -1] = 0, 0, 0, 0
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
        d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, 1, 1)
        wf, wd = ops.packed_weights(w, d, dtype, cache=False)
        M = B * OH * OH
        y = torch.empty(M, Cout, device=dev, dtype=dtype)
        nblk = lib.fva_conv_stat_blocks(C.byref(d))
        stats = torch.zeros(lib.fva_bn_partial_rows(nblk), 2, Cout, device=dev)
        dx = torch.empty(B, H, W, Cin, device=dev, dtype=dtype)
        add = torch.randn(B, H, W, Cin, generator=g).to(dev).to(dtype)
        st = ops._stream()
        fwd = lambda: _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(stats), st)
        dgr = lambda: _lib.call('fva_conv_dgrad', C.byref(d), ops._p(dy), ops._p(wd), ops._p(dx), ops._p(add), st)
        res = {}
        for name, fn in (('fwd', fwd), ('dgrad', dgr)):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            res[name] = (ms * 1e3, 2.0 * M * Cout * Cin * k * k / ms / 1e9)
        # race screen: the same launch repeated must give identical bits
        y0, dx0 = y.clone(), dx.clone()
        for _ in range(5):
            fwd(); dgr()
            assert torch.equal(y, y0) and torch.equal(dx, dx0), 'non-deterministic output'
        out[f's{si}_y'] = y.float().cpu().numpy()[::97, ::3]
        out[f's{si}_ysum'] = np.array([y.double().sum().item(), (y.double() ** 2).sum().item()])
        out[f's{si}_stats'] = stats[:nblk].double().sum(0).cpu().numpy()
        out[f's{si}_dx'] = dx.float().cpu().numpy().reshape(-1, Cin)[::101, ::3]
        out[f's{si}_dxsum'] = np.array([dx.double().sum().item(), (dx.double() ** 2).sum().item()])
        print(SHAPES[si], 'stat rows', nblk, ' | '.join(f'{k_}: {v[0]:.1f} us {v[1]:.0f} TF' for k_, v in res.items()), flush=True)
    np.savez(path, **out)


def compare(a, b):
    za, zb = np.load(a), np.load(b)
    bad = 0
    for k in za.files:
        if k.endswith('_stats'):
            ok = np.allclose(za[k], zb[k], rtol=1e-5, atol=1e-2)     # partial rows are grouped differently (128 vs 256 rows)
        elif os.environ.get('CHECK_TOL'):
            ok = np.allclose(za[k], zb[k], rtol=float(os.environ['CHECK_TOL']), atol=float(os.environ['CHECK_TOL']) * np.abs(za[k]).max())
        else:
            ok = np.array_equal(za[k], zb[k])
        if not ok:
            bad += 1
            print('MISMATCH', k, np.abs(za[k] - zb[k]).max())
    print('compare:', 'all equal' if not bad else f'{bad} mismatching arrays')
    return bad


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2])
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
