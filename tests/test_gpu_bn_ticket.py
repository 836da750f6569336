"""BatchNorm finalisation inside the producing convolution launch (fva_conv_fwd_bn / fva_conv_dgrad_bn, csrc/bn_ticket.h) against the
stand-alone finalize launches it replaces (fva_bn_finalize / fva_bn_bwd_finalize) -- reference op: nn.BatchNorm2d in training mode
(classfication/models/darknet53.py:11-12) and its backward.

Kernel level: every tile variant the dispatcher can pick (128x128, thin 256x64, 8-phase 256x256, stride-2 parity and paired
launches), one- and two-level tables, both dtypes: the partial table is bit-identical with the plain launch's, the finalised
vectors agree with the stand-alone finalize to fp32 rounding of the different (more precise) summation order and with a float64
fold of the table itself, the running statistics and num_batches_tracked advance, the ticket counters are back at zero, and a second
launch reproduces the first bit for bit (the fold order is fixed, whichever wave happens to arrive last).
Model level: a training step with the tickets on equals the step with the stand-alone launches (loss 1e-6, gradients 1e-5 in fp32),
and no finalize launch is left for the fused layers.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


FWD_CASES = [
    # B, Cin, Cout, H, W, k, stride
    (2, 64, 128, 12, 10, 3, 1),       # 128x128 tile, 2 row blocks: one level
    (4, 64, 64, 48, 40, 1, 1),        # thin 256x64 tile, 30 row blocks
    (8, 128, 256, 40, 40, 1, 1),      # 128x128 tile, 100 row blocks x 2 column blocks: two levels (G = 16)
    (2, 32, 64, 64, 64, 3, 1),        # half-row k-tiles (Cin = 32), thin tile, 32 rows
    (3, 64, 96, 20, 20, 3, 2),        # stride 2, Cout = 96: three slices, column tail of the 128 tile
    (8, 256, 512, 64, 64, 3, 1),      # 8-phase 256x256 kernel (bf16): 128 row blocks x 2 column blocks
    (2, 64, 128, 72, 64, 3, 1),       # patch kernel (bf16): 36 patches
]


@pytest.mark.parametrize('key', ['f32', 'bf16'])
@pytest.mark.parametrize('case', FWD_CASES)
def test_forward_statistics_finalised_in_the_launch(case, key):
    from fastvision_amd import _lib, ops
    B, Cin, Cout, H, W, k, s = case
    if key == 'f32' and B * H * W * Cin > 4e6:
        pytest.skip('the large case exists for the bf16 8-phase kernel')
    dtype = DT[key]
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * OH * OW
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    gamma = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(Cout, generator=g) * 0.2).to(DEV)
    keep, xptr, xpad = ops.to_halo(x, dtype, k // 2)
    d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, xpad, 1)
    wf, _ = ops.packed_weights(w, d, dtype, cache=False)
    nblk = lib.fva_conv_stat_blocks(C.byref(d))

    def vecs():
        return [torch.full((Cout,), float('nan'), device=DEV) for _ in range(4)]
    # stand-alone: conv + finalize launch
    y0 = torch.empty((M, Cout), dtype=dtype, device=DEV)
    st0 = torch.full((lib.fva_bn_partial_rows(nblk), 2, Cout), float('nan'), device=DEV)
    rm0, rv0, nbt0 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    m0, r0, sc0, sh0 = vecs()
    _lib.call('fva_conv_fwd', C.byref(d), C.c_void_p(xptr), ops._p(wf), ops._p(y0), ops._p(st0), ops._stream())
    _lib.call('fva_bn_finalize', ops._p(st0), nblk, st0.shape[0], M, Cout, ops._p(gamma), ops._p(beta), ops._p(rm0), ops._p(rv0), ops._p(nbt0),
              0.1, 1e-5, ops._p(m0), ops._p(r0), ops._p(sc0), ops._p(sh0), ops._stream())
    # in the launch
    ng = lib.fva_bn_ticket_groups(nblk)
    cnt = torch.zeros(lib.fva_bn_ticket_counters(nblk, Cout), dtype=torch.int32, device=DEV)
    gs = torch.full((ng, 2, Cout), float('nan'), dtype=torch.float64, device=DEV)
    rm1, rv1, nbt1 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    runs = []
    for rep in range(2):
        y1 = torch.empty((M, Cout), dtype=dtype, device=DEV)
        st1 = torch.full((nblk, 2, Cout), float('nan'), device=DEV)
        m1, r1, sc1, sh1 = vecs()
        fin = _lib.BnFwdFin(cnt.data_ptr(), gs.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm1.data_ptr(), rv1.data_ptr(), nbt1.data_ptr(),
                            0.1, 1e-5, m1.data_ptr(), r1.data_ptr(), sc1.data_ptr(), sh1.data_ptr())
        _lib.call('fva_conv_fwd_bn', C.byref(d), C.c_void_p(xptr), ops._p(wf), ops._p(y1), ops._p(st1), C.byref(fin), ops._stream())
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0, 'the last arrivers put every counter back to zero'
        runs.append((y1, st1, m1, r1, sc1, sh1))
    y1, st1, m1, r1, sc1, sh1 = runs[0]
    assert torch.equal(y0, y1) and torch.equal(st0[:nblk], st1), 'same tile, same partial table as the plain launch'
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b), 'fixed fold order: bit-identical from launch to launch'
    assert int(nbt1) == 2 and int(nbt0) == 1
    # float64 fold of the table itself
    t = st1.double().sum(0)
    mean = t[0] / M
    var = (t[1] / M - mean * mean).clamp_min(0)
    assert torch.allclose(m1.double(), mean, rtol=1e-6, atol=1e-7) and torch.allclose(r1.double(), 1 / torch.sqrt(var + 1e-5), rtol=2e-6)
    assert rel(m1, m0) < 2e-6 and rel(r1, r0) < 2e-6 and rel(sc1, sc0) < 2e-6 and rel(sh1, sh0) < 5e-6
    # running statistics after two updates from (0, 1) with the same batch statistics
    unb = var * M / (M - 1)
    want_rm = 0.9 * (0.1 * mean) + 0.1 * mean
    want_rv = 0.9 * (0.9 * 1.0 + 0.1 * unb) + 0.1 * unb
    assert torch.allclose(rm1.double(), want_rm, rtol=1e-5, atol=1e-7) and torch.allclose(rv1.double(), want_rv, rtol=1e-5)


BWD_CASES = [
    # B, Cin, Cout, H, W, k, stride            (dgrad problem: M = B*H*W rows, N = Cin columns, reduction over Cout)
    (2, 128, 64, 12, 10, 3, 1),      # 128x128 tile, row tail, one level
    (2, 64, 128, 16, 16, 3, 1),      # thin 256x64 tile
    (4, 256, 128, 40, 40, 1, 1),     # 1x1, two column blocks, 50 row blocks... and
    (8, 128, 256, 40, 40, 1, 1),     # 100 row blocks: two levels
    (3, 128, 128, 8, 12, 3, 2),      # stride 2: four parity launches share the table and the counters
    (2, 64, 128, 16, 16, 3, 2),      # stride 2, thin: bf16 the stride-2 patch kernel; fp32 two paired launches (N' = 2 * Cin: two table rows per block)
    (2, 32, 64, 32, 32, 3, 2),       # the same with Cin = 32: a slice is one pixel parity
    (8, 256, 512, 64, 64, 3, 1),     # 8-phase 256x256 kernel (bf16)
    (2, 64, 128, 72, 64, 3, 1),      # patch kernel (bf16): reduction over 128 channels in two slices, N = 64
]


@pytest.mark.parametrize('key', ['f32', 'bf16'])
@pytest.mark.parametrize('with_addend', [False, True])
@pytest.mark.parametrize('case', BWD_CASES)
def test_backward_statistics_finalised_in_the_launch(case, key, with_addend):
    from fastvision_amd import _lib, ops
    B, Cin, Cout, H, W, k, s = case
    if key == 'f32' and B * H * W * Cin > 4e6:
        pytest.skip('the large case exists for the bf16 8-phase kernel')
    dtype = DT[key]
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * H * W
    gy = torch.randn(B, Cout, OH, OW, generator=g)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5).to(DEV)
    y = (torch.randn(M, Cin, generator=g) * 1.5 + 0.3).to(DEV).to(dtype)
    gamma = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(Cin, generator=g) * 0.2).to(DEV)
    yf = y.float()
    mean = yf.mean(0)
    rstd = 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5)
    scale = gamma * rstd
    shift = beta - mean * scale
    add = torch.randn(B, H, W, Cin, generator=g).to(DEV).to(dtype) if with_addend else None
    keep, dyptr, dypad = ops.to_halo(gy.to(DEV), dtype, 1)
    d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, 1, 1)
    _, wd = ops.packed_weights(w, d, dtype, cache=False)
    rows = lib.fva_conv_dgrad_stat_rows(C.byref(d))
    assert rows > 0
    # stand-alone
    part0 = torch.full((lib.fva_bn_partial_rows(rows), 2, Cin), float('nan'), device=DEV)
    fs0 = _lib.BnBwdFuse(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), part0.data_ptr())
    dx0 = torch.empty((B, H, W, Cin), dtype=dtype, device=DEV)
    _lib.call('fva_conv_dgrad_bnstats', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx0), ops._p(add), C.byref(fs0), ops._stream())
    dg0, db0, coef0 = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV), torch.empty((3, Cin), device=DEV)
    _lib.call('fva_bn_bwd_finalize', ops._p(part0), rows, part0.shape[0], M, Cin, ops._p(gamma), ops._p(rstd), ops._p(dg0), ops._p(db0), 0, ops._p(coef0),
              ops._stream())
    # in the launch(es)
    ng = lib.fva_bn_ticket_groups(rows)
    cnt = torch.zeros(lib.fva_bn_ticket_counters(rows, Cin), dtype=torch.int32, device=DEV)
    gs = torch.full((ng, 2, Cin), float('nan'), dtype=torch.float64, device=DEV)
    runs = []
    for rep in range(2):
        part1 = torch.full((rows, 2, Cin), float('nan'), device=DEV)
        fs1 = _lib.BnBwdFuse(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), part1.data_ptr())
        dg1, db1, coef1 = torch.full((Cin,), float('nan'), device=DEV), torch.full((Cin,), float('nan'), device=DEV), torch.full((3, Cin), float('nan'), device=DEV)
        fin = _lib.BnBwdFin(cnt.data_ptr(), gs.data_ptr(), gamma.data_ptr(), dg1.data_ptr(), db1.data_ptr(), coef1.data_ptr(), 0)
        dx1 = torch.empty_like(dx0)
        _lib.call('fva_conv_dgrad_bn', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx1), ops._p(add), C.byref(fs1), C.byref(fin), ops._stream())
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0
        runs.append((dx1, part1, dg1, db1, coef1))
    dx1, part1, dg1, db1, coef1 = runs[0]
    assert torch.equal(dx0, dx1) and torch.equal(part0[:rows], part1)
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    t = part1.double().sum(0)
    assert torch.allclose(db1.double(), t[0], rtol=1e-6, atol=1e-6 * t[0].abs().max().item())
    assert torch.allclose(dg1.double(), t[1], rtol=1e-6, atol=1e-6 * t[1].abs().max().item())
    sc_b, sc_g = db0.abs().max().item(), dg0.abs().max().item()
    assert (db1 - db0).abs().max().item() < 1e-5 * sc_b and (dg1 - dg0).abs().max().item() < 1e-5 * sc_g
    assert rel(coef1, coef0) < 1e-5


def test_descriptor_is_checked():
    from fastvision_amd import _lib, ops
    lib = _lib.load()
    d = _lib.ConvDesc(_lib.BF16, 1, 8, 8, 64, 72, 1, 1, 0, 1)       # Cout = 72: not a multiple of 32
    t = torch.zeros(4096, device=DEV)
    fin = _lib.BnFwdFin(*([t.data_ptr()] * 7), 0.1, 1e-5, *([t.data_ptr()] * 4))
    rc = lib.fva_conv_fwd_bn(C.byref(d), ops._p(t), ops._p(t), ops._p(t), ops._p(t), C.byref(fin), None)
    assert rc != 0 and b'C % 32' in lib.fva_last_error()
    # a table that is too short for the two-level stand-alone finalize is refused, not written past
    rc = lib.fva_bn_bwd_finalize(ops._p(t), 2000, 2000, 100, 8, ops._p(t), ops._p(t), ops._p(t), ops._p(t), 0, ops._p(t), None)
    assert rc != 0 and b'fva_bn_partial_rows' in lib.fva_last_error()


class _Count:
    def __init__(self):
        self.n = {}

    def __call__(self, name, args):
        self.n[name] = self.n.get(name, 0) + 1
        return None


@pytest.mark.parametrize('key', ['f32', 'bf16'])
def test_training_step_same_with_and_without_tickets(key):
    import fastvision_amd
    from fastvision_amd import _lib, ops
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.synthetic import synthetic_batch
    from test_gpu_bn_fusion import lib_model
    images, tg = synthetic_batch(4, 160)
    images, tg = images.to(DEV), tg.to(DEV)
    out = {}
    for on in (False, True):
        prev = ops.set_bn_ticket_finalize(on)
        try:
            with fastvision_amd.compute_dtype(DT[key]):
                net = lib_model()
                crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
                cnt = _Count()
                _lib.tracer = cnt
                try:
                    loss = crit(net(images), tg)
                    loss.backward()
                finally:
                    _lib.tracer = None
                torch.cuda.synchronize()
                bufs = {k: v.detach().clone() for k, v in net.named_buffers()}
                out[on] = (float(loss), [p.grad.detach().clone() for p in net.parameters()], cnt.n, bufs)
        finally:
            ops.set_bn_ticket_finalize(prev)
    (l0, g0, n0, b0), (l1, g1, n1, b1) = out[False], out[True]
    devs = np.array([rel(a, b) for a, b in zip(g1, g0)])
    print(f'{key}: loss {l0} vs {l1}; finalize launches {n0.get("fva_bn_finalize", 0)} + {n0.get("fva_bn_bwd_finalize", 0)} -> '
          f'{n1.get("fva_bn_finalize", 0)} + {n1.get("fva_bn_bwd_finalize", 0)}; gradient deviation median {np.median(devs):.2e}, largest {devs.max():.2e}')
    assert abs(l0 - l1) <= (1e-6 if key == 'f32' else 2e-3) * abs(l0)
    for k in b0:
        assert rel(b1[k].float(), b0[k].float()) < 1e-5, k                       # running statistics, num_batches_tracked
    if key == 'f32':
        assert devs.max() < 2e-5
    else:
        norms = np.array([abs(a.double().norm().item() - b.double().norm().item()) / max(b.double().norm().item(), 1e-30) for a, b in zip(g1, g0)])
        assert np.median(norms) < 5e-3 and norms.max() < 6e-2
    # what is left: the stem (its statistics come from stem_fused_kernel) and the layers whose backward statistics are not fused
    assert n0.get('fva_conv_fwd_bn', 0) == 0 and n0.get('fva_conv_dgrad_bn', 0) == 0
    assert n1.get('fva_bn_finalize', 0) <= 1 and n1.get('fva_bn_bwd_finalize', 0) <= 10
    assert n1.get('fva_conv_fwd_bn', 0) >= 70 and n1.get('fva_conv_dgrad_bn', 0) >= 60
