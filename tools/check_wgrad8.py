#!/usr/bin/env python3
"""A/B of the 8-phase wgrad kernel against the 128x128 kernel: run once per setting of FVA_WGRAD8, dump dW, compare.
tools/check_wgrad8.py run <out.npz> | compare <a.npz> <b.npz>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if os.environ.get('CHECK_SHAPES') == 'small':
    SHAPES = [(32, 128, 256, 80, 3, 1), (16, 256, 512, 40, 3, 1), (32, 512, 1024, 40, 3, 2), (33, 256, 512, 37, 3, 1)]
else:
    SHAPES = [(32, 128, 256, 80, 3, 1), (32, 256, 512, 40, 3, 1), (32, 512, 1024, 20, 3, 1), (32, 128, 256, 160, 3, 2),
              (32, 256, 512, 80, 3, 2), (32, 512, 1024, 40, 3, 2), (7, 256, 512, 33, 3, 1)]


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
        d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, 1, 1)
        dw = torch.empty(Cout, Cin, k, k, device=dev)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st = ops._stream()
        fn = lambda: _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        dw0 = dw.clone()
        for _ in range(5):
            fn()
            assert torch.equal(dw, dw0), 'non-deterministic output'
        out[f's{si}_dw'] = dw.cpu().numpy()
        M = B * OH * OH
        print(SHAPES[si], f'workspace {wsb / 2**20:.0f} MiB  wgrad: {ms * 1e3:.1f} us {2.0 * M * Cout * Cin * k * k / ms / 1e9:.0f} TF', flush=True)
    np.savez(path, **out)


def compare(a, b):
    za, zb = np.load(a), np.load(b)
    bad = 0
    for k in za.files:
        scale = np.abs(za[k]).max()
        err = np.abs(za[k] - zb[k]).max() / scale
        if not err < 2e-5:
            bad += 1
        print(k, 'max abs err / scale', err)
    print('compare:', 'all within 2e-5' if not bad else f'{bad} mismatching arrays')
    return bad


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2])
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
