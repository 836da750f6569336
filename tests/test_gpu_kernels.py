"""GPU parity tests, kernel level: every C-ABI compute entry point against a plain PyTorch fp32 reference of
the same op (conv / BN / SiLU arithmetic is PyTorch's in the reference).  All calls go through the
C ABI (fastvision_amd._lib / fastvision_amd.ops).

Tolerances: fp32 path 1e-4 relative to the tensor scale (north_star allows 1e-3); bf16 path is checked against
the same fp32 reference fed with bf16-rounded operands, 1e-2 of the tensor scale (bf16 output rounding 2^-9).
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {'f32': torch.float32, 'bf16': torch.bfloat16}
TOL = {'f32': 1e-4, 'bf16': 1e-2}


def dev():
    return torch.device('cuda:0')


def rel_err(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-12)).item()


def halo(x_nchw, dtype, pad=1):
    """NCHW fp32 CPU tensor -> (halo buffer on GPU, logical view) via the product's own packer."""
    from fastvision_amd import ops
    keep, ptr, p = ops.to_halo(x_nchw.to(dev()), dtype, pad)
    return keep, ptr, p


def rounded(t, key):
    return t.to(DT[key]).float()


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride
    (2, 64, 128, 12, 10, 3, 1),      # M tail (240 rows), wide tile
    (2, 32, 64, 16, 16, 3, 2),       # Cin=32 (bf16 half-row k-tiles), stride 2, narrow tile
    (1, 128, 64, 9, 7, 1, 1),        # 1x1, odd sizes
    (2, 64, 32, 8, 8, 1, 1),         # Cout=32 masked in the 64-wide tile
    (1, 256, 256, 13, 13, 3, 1),     # two column blocks, several k-tiles per tap
    (3, 64, 128, 8, 12, 3, 2),       # stride 2, non-square
    (1, 384, 128, 6, 6, 1, 1),       # concat-style channel count
    # thin 3x3 stride-1 layers on maps >= 64 x 64: the patch kernel (bf16; fp32 stays on the implicit-GEMM kernels)
    (2, 32, 64, 64, 96, 3, 1),       # forward C = 32 / N = 64 (resident weights), dgrad C = 64 / N = 32; whole 8 x 32 patches
    (1, 64, 128, 70, 72, 3, 1),      # forward C = 64 / N = 128 (weight ring), dgrad C = 128 / N = 64 (two channel slices); patches overhang both ways
    (1, 128, 64, 66, 64, 3, 1),      # forward two channel slices, dgrad the weight ring
    (1, 64, 64, 64, 65, 3, 1),       # C = N = 64
    # thin 3x3 stride-2 layers: the data gradient runs on the stride-2 patch kernel (bf16; the small stride-2 cases above too)
    (1, 32, 64, 72, 136, 3, 2),      # dgrad C = 64 / N = 32, resident weights; 8 x 64 dx tiles overhang in x
    (1, 64, 128, 40, 72, 3, 2),      # dgrad C = 128 in two slices / N = 64, weight ring; overhang both ways
]


@pytest.mark.parametrize('key', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, key):
    from fastvision_amd import _lib, ops
    B, Cin, Cout, H, W, k, s = case
    dtype = DT[key]
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    gy = torch.randn(B, Cout, OH, OW, generator=g)
    xr, wr, gyr = rounded(x, key), rounded(w, key), rounded(gy, key)
    want_y = F.conv2d(xr, wr, stride=s, padding=k // 2)
    want_dx = torch.nn.grad.conv2d_input(x.shape, wr, gyr, stride=s, padding=k // 2)
    want_dw = torch.nn.grad.conv2d_weight(xr, w.shape, gyr, stride=s, padding=k // 2)

    lib = _lib.load()
    keep, xptr, xpad = halo(x, dtype)
    d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, xpad, 1)
    wg = w.to(dev())
    wf, wd = ops.packed_weights(wg, d, dtype, cache=False)
    M = B * OH * OW
    y = torch.empty((M, Cout), dtype=dtype, device=dev())
    nblk = lib.fva_conv_stat_blocks(C.byref(d))
    stats = torch.full((nblk, 2, Cout), float('nan'), device=dev())
    _lib.call('fva_conv_fwd', C.byref(d), C.c_void_p(xptr), ops._p(wf), ops._p(y), ops._p(stats), ops._stream())
    got_y = y.float().view(B, OH, OW, Cout).permute(0, 3, 1, 2)
    assert rel_err(got_y, want_y) < TOL[key], f'conv fwd {case} {key}: {rel_err(got_y, want_y)}'
    # BatchNorm partial statistics: per-channel sum / sum of squares of the fp32 conv result (taken from the fp32
    # accumulators, i.e. before the bf16 rounding of the stored tile)
    ref = want_y.permute(0, 2, 3, 1).reshape(M, Cout)
    ysum, ysq = ref.sum(0), (ref ** 2).sum(0)
    assert torch.allclose(stats[:, 0].sum(0).cpu(), ysum, rtol=1e-3, atol=1e-3 * ysq.max().sqrt().item())
    assert torch.allclose(stats[:, 1].sum(0).cpu(), ysq, rtol=1e-3, atol=1e-3)

    # dgrad (with and without the fused addend) and wgrad from a halo dY buffer
    dyk, dyptr, dypad = halo(gy, dtype)
    dx = torch.empty((B, H, W, Cin), dtype=dtype, device=dev())
    _lib.call('fva_conv_dgrad', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx), C.c_void_p(0), ops._stream())
    got_dx = dx.float().permute(0, 3, 1, 2)
    assert rel_err(got_dx, want_dx) < TOL[key], f'conv dgrad {case} {key}: {rel_err(got_dx, want_dx)}'
    add = torch.randn(B, H, W, Cin, generator=g).to(dev()).to(dtype)
    dx2 = torch.empty_like(dx)
    _lib.call('fva_conv_dgrad', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx2), ops._p(add), ops._stream())
    assert rel_err(dx2.float(), dx.float() + add.float()) < TOL[key]

    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=dev())
    wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(xptr), C.c_void_p(dyptr), ops._p(dw), 0, ops._p(ws), wsb, ops._stream())
    assert rel_err(dw, want_dw) < TOL[key], f'conv wgrad {case} {key}: {rel_err(dw, want_dw)}'
    # accumulate form
    _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(xptr), C.c_void_p(dyptr), ops._p(dw), 1, ops._p(ws), wsb, ops._stream())
    assert rel_err(dw, 2 * want_dw) < TOL[key]


@pytest.mark.parametrize('size', [(2, 20, 28), (3, 24, 48)])        # W = 48: the bf16 MFMA forward; 28: the direct kernel
@pytest.mark.parametrize('key', ['f32', 'bf16'])
def test_stem_fwd_wgrad(key, size):
    from fastvision_amd import _lib, ops
    dtype = DT[key]
    g = torch.Generator().manual_seed(5)
    B, H, W = size
    img = torch.rand(B, 3, H, W, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    gy = torch.randn(B, 32, H, W, generator=g)
    want_y = F.conv2d(img, w, padding=1)
    want_dw = torch.nn.grad.conv2d_weight(img, w.shape, rounded(gy, key), padding=1)
    lib = _lib.load()
    imgd, wdv = img.to(dev()), w.to(dev())
    M = B * H * W
    y = torch.empty((M, 32), dtype=dtype, device=dev())
    code = ops._code(dtype)
    nblk = lib.fva_stem_stat_blocks(code, B, H, W)
    stats = torch.empty((nblk, 2, 32), device=dev())
    wsb = lib.fva_stem_fwd_workspace(code, B, H, W)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev())
    _lib.call('fva_stem_fwd', code, ops._p(imgd), ops._p(wdv), ops._p(y), ops._p(stats), ops._p(ws), wsb, B, 3, H, W, 32, ops._stream())
    got = y.float().view(B, H, W, 32).permute(0, 3, 1, 2)
    assert rel_err(got, want_y) < TOL[key]
    # the direct kernel sums the stored (rounded) values, the MFMA kernel its fp32 accumulators: equal up to the bf16 rounding
    # of M values per channel
    assert torch.allclose(stats[:, 0].sum(0).cpu(), y.float().sum(0).cpu(), rtol=2e-3, atol=0.5)
    assert torch.allclose(stats[:, 1].sum(0).cpu(), (y.float() ** 2).sum(0).cpu(), rtol=2e-3, atol=0.5)
    dy = gy.permute(0, 2, 3, 1).contiguous().to(dev()).to(dtype)
    dw = torch.empty((32, 3, 3, 3), device=dev())
    wsb = lib.fva_stem_wgrad_workspace(B, 3, H, W, 32)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    _lib.call('fva_stem_wgrad', ops._code(dtype), ops._p(imgd), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, B, 3, H, W, 32, ops._stream())
    assert rel_err(dw, want_dw) < 1e-4


@pytest.mark.parametrize('key', ['f32', 'bf16'])
def test_conv_block_module_matches_oracle(key):
    """ConvBlock3x3 / ConvBlock1x1 / ResidualBlock modules (fwd, dx, dW, dgamma, dbeta, running stats) vs the oracle."""
    import fastvision_amd
    from fastvision_amd.classfication.models.darknet53 import ConvBlock1x1, ConvBlock3x3, ResidualBlock
    from oracle import model as om
    dtype = DT[key]
    tol = 2e-4 if key == 'f32' else 4e-2
    g = torch.Generator().manual_seed(7)
    cases = [(lambda: ConvBlock3x3(32, 64), lambda: om.ConvUnit(32, 64, 3), (2, 32, 10, 12)),
             (lambda: ConvBlock3x3(32, 64, stride=(2, 2)), lambda: om.ConvUnit(32, 64, 3, 2), (2, 32, 10, 12)),
             (lambda: ConvBlock1x1(64, 32), lambda: om.ConvUnit(64, 32, 1), (2, 64, 6, 6)),
             (lambda: ResidualBlock(64, 32), lambda: om.Residual(64), (2, 64, 8, 8))]
    with fastvision_amd.compute_dtype(dtype):
        for mk, mko, shape in cases:
            torch.manual_seed(11)
            mod = mk().to(dev()).train()
            torch.manual_seed(11)
            ref = mko().train()
            assert [k for k, _ in mod.state_dict().items()] == [k for k, _ in ref.state_dict().items()]
            x = torch.randn(shape, generator=g)
            gy_shape = ref(x).shape
            ref.zero_grad()
            for m_ in ref.modules():
                if isinstance(m_, torch.nn.BatchNorm2d):
                    m_.reset_running_stats()
            gy = torch.randn(gy_shape, generator=g)
            xr = x.clone().requires_grad_(True)
            yr = ref(xr)
            (yr * gy).sum().backward()
            xg = x.to(dev()).requires_grad_(True)
            yg = mod(xg)
            assert tuple(yg.shape) == tuple(yr.shape)
            (yg.float() * gy.to(dev())).sum().backward()
            assert rel_err(yg, yr) < tol, f'{key} fwd {rel_err(yg, yr)}'
            assert rel_err(xg.grad, xr.grad) < tol, f'{key} dx {rel_err(xg.grad, xr.grad)}'
            for (kn, pg), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
                assert rel_err(pg.grad, pr.grad) < tol, f'{key} grad {kn} {rel_err(pg.grad, pr.grad)}'
            for (kn, bg), (_, br) in zip(mod.named_buffers(), ref.named_buffers()):
                assert rel_err(bg.float(), br.float()) < tol, f'{key} buffer {kn}'


@pytest.mark.parametrize('key', ['f32', 'bf16'])
@pytest.mark.parametrize('up_first', [True, False])
def test_upsample_concat(key, up_first):
    import fastvision_amd
    from fastvision_amd import ops
    dtype = DT[key]
    g = torch.Generator().manual_seed(3)
    up = torch.randn(2, 64, 4, 5, generator=g)
    skip = torch.randn(2, 32, 8, 10, generator=g)
    upr, skr = rounded(up, key).clone().requires_grad_(True), rounded(skip, key).clone().requires_grad_(True)
    u = F.interpolate(upr, scale_factor=2, mode='nearest')
    want = torch.cat([u, skr] if up_first else [skr, u], dim=1)
    gy = torch.randn(want.shape, generator=g)
    (want * rounded(gy, key)).sum().backward()
    ug, sg = up.detach().clone().to(dev()).requires_grad_(True), skip.detach().clone().to(dev()).requires_grad_(True)
    with fastvision_amd.compute_dtype(dtype):
        got = ops.upsample2_concat(ug, sg, up_first)
    assert rel_err(got, want) < 1e-6
    (got.float() * rounded(gy, key).to(dev())).sum().backward()
    assert rel_err(ug.grad, upr.grad) < TOL[key]
    assert rel_err(sg.grad, skr.grad) < TOL[key]


def test_adam_matches_torch():
    from fastvision_amd import FusedAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(64, 32, 3, 3), (255,), (1000, 33), (7,)]
    ps_ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    ps_gpu = [p.detach().clone().to(dev()).requires_grad_(True) for p in ps_ref]
    o_ref = torch.optim.Adam(ps_ref, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
    o_gpu = FusedAdam(ps_gpu, lr=1e-3, betas=(0.937, 0.999), weight_decay=5e-4)
    for step in range(5):
        for pr, pg in zip(ps_ref, ps_gpu):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad = gr.clone()
            pg.grad = gr.to(dev())
        v0 = ps_gpu[0]._version
        o_ref.step()
        o_gpu.step()
        assert ps_gpu[0]._version > v0
    for pr, pg in zip(ps_ref, ps_gpu):
        assert torch.allclose(pg.cpu(), pr, rtol=1e-5, atol=1e-6)
    sd = o_gpu.state_dict()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}


@pytest.mark.parametrize('key', ['f32', 'bf16'])
def test_multi_tensor_weight_pack_matches_single(key):
    """fva_conv_pack_weights_multi (one launch for all layers) must produce exactly the per-layer packing."""
    from fastvision_amd import _lib, ops
    dtype = DT[key]
    lib = _lib.load()
    shapes = [(64, 32, 3), (32, 64, 1), (255, 128, 1), (128, 64, 3), (48, 40, 3), (256, 128, 3), (64, 128, 3)]
    g = torch.Generator().manual_seed(9)
    ws = [torch.randn(co, ci, k, k, generator=g).to(dev()) for co, ci, k in shapes]
    arr = (_lib.PackEntry * len(ws))()
    singles, multis, mx = [], [], 0
    for i, (w, (co, ci, k)) in enumerate(zip(ws, shapes)):
        d = _lib.ConvDesc(ops._code(dtype), 1, 8, 8, ci, co, k, 1, 1, 1)
        singles.append(ops.packed_weights(w, d, dtype, cache=False))
        wf = torch.full((lib.fva_conv_packed_elems(C.byref(d), 0),), float('nan'), dtype=dtype, device=dev())
        wd = torch.full((lib.fva_conv_packed_elems(C.byref(d), 1),), float('nan'), dtype=dtype, device=dev())
        multis.append((wf, wd))
        e = arr[i]
        e.w, e.w_fwd, e.w_dgrad = w.data_ptr(), wf.data_ptr(), wd.data_ptr()
        e.Cout, e.Cin, e.ksize, e.dtype = co, ci, k, ops._code(dtype)
        e.taps_fwd, e.taps_dgrad = wf.numel() // (co * ci), wd.numel() // (co * ci)
        mx = max(mx, wf.numel(), wd.numel())
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev())
    _lib.call('fva_conv_pack_weights_multi', ops._p(table), len(ws), mx, ops._stream())
    for (sf, sd), (mf, md) in zip(singles, multis):
        assert torch.equal(sf.float(), mf.float()) and torch.equal(sd.float(), md.float())
    # ... and the tiled launch (one block per 32 x 32 weight tile of any layer: what the pack registry uses)
    for mf, md in multis:
        mf.fill_(float('nan')); md.fill_(float('nan'))
    tiles = 0
    for i, (co, ci, k) in enumerate(shapes):
        arr[i].tile_start = tiles
        tiles += ((co + 31) // 32) * ((ci + 31) // 32)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev())
    _lib.call('fva_conv_pack_weights_tiled', ops._p(table), len(ws), tiles, ops._stream())
    for (sf, sd), (mf, md) in zip(singles, multis):
        assert torch.equal(sf.float(), mf.float()) and torch.equal(sd.float(), md.float())


@pytest.mark.parametrize('nblocks', [7, 1023, 1024, 3001, 51200])
def test_bn_finalize_large_tables(nblocks):
    """partial tables of >= 1024 rows go through the parallel pre-reduce (rows behind nblocks are its scratch)"""
    from fastvision_amd import _lib, ops
    lib = _lib.load()
    Cc = 32
    g = torch.Generator().manual_seed(nblocks)
    rows = lib.fva_bn_partial_rows(nblocks)
    assert rows >= nblocks
    part = torch.full((rows, 2, Cc), float('nan'), device=dev())
    vals = torch.rand(nblocks, 2, Cc, generator=g) * 50
    vals[:, 1] += 3000                                        # sum of squares dominates the squared mean
    part[:nblocks] = vals.to(dev())
    count = nblocks * 256
    gamma, beta = torch.rand(Cc, generator=g).to(dev()) + 0.5, torch.randn(Cc, generator=g).to(dev())
    rm, rv = torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())
    nbt = torch.zeros(1, dtype=torch.int64, device=dev())
    mean, rstd, scale, shift = (torch.empty(Cc, device=dev()) for _ in range(4))
    _lib.call('fva_bn_finalize', ops._p(part), nblocks, rows, count, Cc, ops._p(gamma), ops._p(beta), ops._p(rm), ops._p(rv), ops._p(nbt),
              0.1, 1e-5, ops._p(mean), ops._p(rstd), ops._p(scale), ops._p(shift), ops._stream())
    s = vals.double().sum(0)
    m = s[0] / count
    var = s[1] / count - m * m
    assert torch.allclose(mean.cpu().double(), m, rtol=2e-6, atol=1e-7)
    assert torch.allclose(rstd.cpu().double(), 1 / torch.sqrt(var + 1e-5), rtol=2e-5)
    assert torch.allclose(rm.cpu().double(), 0.1 * m, rtol=2e-6, atol=1e-7)
    assert int(nbt) == 1
    assert torch.equal(part[:nblocks].cpu(), vals)            # the inputs themselves are not touched


def test_igemm_8phase_kernel_equals_128_tile_kernel(tmp_path):
    """the 256x256 8-phase kernel (default for the long-reduction bf16 layers) against the 128x128 kernel (FVA_IGEMM8=0) on
    the same inputs: forward + BN partials and dgrad with addend, stride 1 and 2, partial last tile -- bit-identical
    (same accumulation order), and identical across repeated launches (race screen).  Two child processes: the switch is
    read once per process."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'check_igemm8.py')
    outs = []
    for flag in ('1', '0'):
        out = str(tmp_path / f'ig{flag}.npz')
        env = dict(os.environ, FVA_IGEMM8=flag, CHECK_SHAPES='small')
        r = subprocess.run([sys.executable, tool, 'run', out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out)
        if flag == '1':
            assert 'stat rows 200' in r.stdout            # 256-row tiles were really used
    r = subprocess.run([sys.executable, tool, 'compare'] + outs, capture_output=True, text=True)
    assert r.returncode == 0 and 'all equal' in r.stdout, r.stdout + r.stderr


def test_wgrad_8phase_kernel_matches_128_tile_kernel(tmp_path):
    """the 256x256 8-phase weight-gradient kernel (default for the large 3x3 bf16 layers) against the 128x128 kernel
    (FVA_WGRAD8=0) on the same inputs, stride 1 and 2, two-taps-per-tile packing (Cin = 128), ragged pixel counts: equal to
    fp32 rounding of the different split-K order, and bit-identical across repeated launches.  Two child processes."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'check_wgrad8.py')
    outs = []
    for flag in ('1', '0'):
        out = str(tmp_path / f'wg{flag}.npz')
        env = dict(os.environ, FVA_WGRAD8=flag, CHECK_SHAPES='small')
        r = subprocess.run([sys.executable, tool, 'run', out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out)
    r = subprocess.run([sys.executable, tool, 'compare'] + outs, capture_output=True, text=True)
    assert r.returncode == 0 and 'all within' in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize('key', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 32, 64, 24, 3, 2), (2, 128, 256, 20, 3, 1), (3, 64, 32, 18, 1, 1), (32, 256, 512, 40, 3, 1)])
def test_conv_fwd_bnact_equals_conv_then_apply(key, shape):
    """the inference epilogue (affine + SiLU + residual inside the conv kernel, border by a second launch) against the
    two-kernel path conv -> bn_silu_apply: fp32 to 1e-5, bf16 to the rounding of the unfused path's bf16 conv output"""
    from fastvision_amd import _lib, ops
    lib = _lib.load()
    B, Cin, Cout, H, k, s = shape
    dtype = torch.float32 if key == 'f32' else torch.bfloat16
    if key == 'f32' and B == 32:
        pytest.skip('large case is for the bf16 8-phase kernel')
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(dev()).to(dtype)
    x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev())
    OH = (H - 1) // s + 1
    d = _lib.ConvDesc(ops._code(dtype), B, H, H, Cin, Cout, k, s, 1, 1)
    wf, _ = ops.packed_weights(w, d, dtype, cache=False)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev())
    shift = torch.randn(Cout, generator=g).to(dev())
    res = torch.randn(B, OH + 2, OH + 2, Cout, generator=g).to(dev()).to(dtype)
    st = ops._stream()
    y = torch.empty(B * OH * OH, Cout, dtype=dtype, device=dev())
    _lib.call('fva_conv_fwd', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), C.c_void_p(0), st)
    for residual in (None, res):
        ref = torch.full((B, OH + 2, OH + 2, Cout), float('nan'), dtype=dtype, device=dev())
        _lib.call('fva_bn_silu_apply', ops._code(dtype), ops._p(y), ops._p(scale), ops._p(shift), ops._p(residual), 1, ops._p(ref), 1,
                  B, OH, OH, Cout, st)
        got = torch.full((B, OH + 2, OH + 2, Cout), float('nan'), dtype=dtype, device=dev())
        _lib.call('fva_conv_fwd_bnact', C.byref(d), ops._p(x), ops._p(wf), ops._p(scale), ops._p(shift), ops._p(residual), ops._p(got), 1, st)
        assert torch.isfinite(got).all()
        assert (got[:, 0] == 0).all() and (got[:, -1] == 0).all() and (got[:, :, 0] == 0).all() and (got[:, :, -1] == 0).all()
        tol = 1e-5 if key == 'f32' else 2e-2
        assert rel_err(got.float(), ref.float()) < tol, (key, shape, residual is not None, rel_err(got.float(), ref.float()))


def test_stem_wgrad_mfma_matches_direct_kernel():
    """conv0's weight gradient through the MFMA wgrad kernel (3 vertical taps over 4-pixel windows of the packed NHWC4 image)
    against the register-blocked direct kernel, on the bf16-rounded operands both see"""
    from fastvision_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    B, H, W = 3, 40, 64
    img = torch.rand(B, 3, H, W, generator=g).to(dev())
    w = (torch.randn(32, 3, 3, 3, generator=g) * 0.2).to(dev())
    gy = torch.randn(B, H, W, 32, generator=g).to(dev()).to(torch.bfloat16)
    y = torch.empty(B * H * W, 32, dtype=torch.bfloat16, device=dev())
    wsb = lib.fva_stem_fwd_workspace(1, B, H, W)
    img4 = torch.empty(wsb, dtype=torch.uint8, device=dev())
    st = ops._stream()
    _lib.call('fva_stem_fwd', 1, ops._p(img), ops._p(w), ops._p(y), C.c_void_p(0), ops._p(img4), wsb, B, 3, H, W, 32, st)
    dyh = torch.zeros(B, H + 2, W + 2, 32, dtype=torch.bfloat16, device=dev())
    dyh[:, 1:-1, 1:-1] = gy
    raw = torch.full((32, 4, 4, 3), float('nan'), device=dev())
    wb = lib.fva_stem_wgrad_mfma_workspace()
    ws = torch.empty(wb, dtype=torch.uint8, device=dev())
    _lib.call('fva_stem_wgrad_mfma', ops._p(img4), ops._p(dyh), ops._p(raw), ops._p(ws), wb, B, H, W, st)
    got = raw[:, :3, :3, :].permute(0, 2, 3, 1)
    want = torch.nn.grad.conv2d_weight(img.bfloat16().float(), w.shape, gy.float().permute(0, 3, 1, 2), padding=1)
    assert rel_err(got, want) < 1e-4, rel_err(got, want)


def test_stem_fused_passes_equal_unfused_path(monkeypatch):
    """bf16 stem block (conv0 + BN + SiLU, forward and backward) with conv0 recomputed inside the four BatchNorm passes
    against the path that stores the pre-BN output: same z, dW, dgamma, dbeta and running statistics up to the bf16
    rounding of the stored intermediate (the fused path keeps it in fp32 registers)."""
    import fastvision_amd
    from fastvision_amd.classfication.models.darknet53 import ConvBlock3x3
    g = torch.Generator().manual_seed(2)
    img = torch.rand(4, 3, 32, 48, generator=g).to(dev())
    gz = torch.randn(4, 32, 32, 48, generator=g).to(dev())
    res = {}
    with fastvision_amd.compute_dtype(torch.bfloat16):
        for flag in ('1', '0'):
            monkeypatch.setenv('FVA_STEM_FUSED', flag)
            torch.manual_seed(4)
            blk = ConvBlock3x3(3, 32, stride=(1, 1)).to(dev()).train()
            z = blk(img)
            z.backward(gz.to(z.dtype))
            res[flag] = (z.detach().float(), blk.conv.weight.grad.clone(), blk.bn.weight.grad.clone(), blk.bn.bias.grad.clone(),
                         blk.bn.running_mean.clone(), blk.bn.running_var.clone())
    names = ('z', 'dW', 'dgamma', 'dbeta', 'running_mean', 'running_var')
    for name, a, b in zip(names, res['1'], res['0']):
        assert rel_err(a, b) < 2e-2, (name, rel_err(a, b))
    assert rel_err(res['1'][4], res['0'][4]) < 1e-4 and rel_err(res['1'][5], res['0'][5]) < 1e-4


@pytest.mark.parametrize('wire', [torch.bfloat16, torch.float32])
def test_gather_cast_fills_a_gradient_bucket(wire):
    """fva_gather_cast (parallel.GradientReducer's bucket fill; the reference gathers gradients tensor by tensor inside
    nn.DataParallel, demos/yolov3_u/train.py:85): n fp32 tensors -> one flat buffer at given element offsets, narrowed to the wire
    dtype with round-to-nearest-even; odd sizes / offsets take the scalar path, a null pointer or a zero count skips the tensor."""
    import ctypes as C
    from fastvision_amd import _lib, ops
    DEV = dev()
    g = torch.Generator().manual_seed(5)
    sizes = [4096, 255, 1, 32 * 3 * 3 * 3, 1024 * 512, 7, 64]
    srcs = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += n
    dst = torch.full((off,), float('nan'), dtype=wire, device=DEV)
    ptrs = [t.data_ptr() for t in srcs]
    ptrs[2] = 0                                   # skipped by pointer
    cnt = list(sizes)
    cnt[5] = 0                                    # skipped by count
    table = torch.tensor(ptrs + cnt + offs, dtype=torch.int64).to(DEV)
    _lib.call('fva_gather_cast', C.c_void_p(table.data_ptr()), len(sizes), max(sizes), C.c_void_p(dst.data_ptr()), ops._code(wire), ops._stream())
    torch.cuda.synchronize()
    for i, (t, o, n) in enumerate(zip(srcs, offs, sizes)):
        got = dst[o:o + n]
        if i in (2, 5):
            assert torch.isnan(got.float()).all(), i
        else:
            assert torch.equal(got, t.to(wire)), i


@pytest.mark.parametrize('case', [(2, 32, 64, 64, 96), (1, 64, 128, 70, 72), (1, 128, 64, 66, 64), (1, 64, 64, 64, 65)])
def test_patch_kernel_against_the_implicit_gemm_kernels(case):
    """pconv_kernel (thin 3x3 stride-1 bf16 layers: the patch of an 8 x 32 output tile staged once for the nine taps) against the
    implicit-GEMM kernels it replaces on the same inputs, switched in-process (fva_conv_patch_kernel): forward tile + BatchNorm
    statistics, dgrad with the residual addend, dgrad with the fused BatchNorm-backward statistics.  Same products, another summation
    order: outputs agree to bf16 rounding of the stored value (one ulp = 2^-8 relative), column sums of the statistics to 1e-4."""
    from fastvision_amd import _lib, ops
    B, Cin, Cout, H, W = case
    lib = _lib.load()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev())
    gy = torch.randn(B, Cout, H, W, generator=g)
    keep, xptr, xpad = halo(x, dtype)
    dyk, dyptr, dypad = halo(gy, dtype)
    d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, 3, 1, xpad, 1)
    wf, wd = ops.packed_weights(w, d, dtype, cache=False)
    M = B * H * W
    add = torch.randn(B, H, W, Cin, generator=g).to(dev()).to(dtype)
    yprod = (torch.randn(M, Cin, generator=g) * 1.5 + 0.3).to(dev()).to(dtype)
    v4 = [(torch.rand(Cin, generator=g) + 0.5).to(dev()) for _ in range(4)]
    out = {}
    prev = lib.fva_conv_patch_kernel(1)
    try:
        for on in (0, 1):
            lib.fva_conv_patch_kernel(on)
            nblk = lib.fva_conv_stat_blocks(C.byref(d))
            y = torch.empty((M, Cout), dtype=dtype, device=dev())
            stats = torch.full((nblk, 2, Cout), float('nan'), device=dev())
            _lib.call('fva_conv_fwd', C.byref(d), C.c_void_p(xptr), ops._p(wf), ops._p(y), ops._p(stats), ops._stream())
            dx = torch.empty((B, H, W, Cin), dtype=dtype, device=dev())
            _lib.call('fva_conv_dgrad', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx), ops._p(add), ops._stream())
            rows = lib.fva_conv_dgrad_stat_rows(C.byref(d))
            part = torch.full((lib.fva_bn_partial_rows(rows), 2, Cin), float('nan'), device=dev())
            fs = _lib.BnBwdFuse(yprod.data_ptr(), v4[0].data_ptr(), v4[1].data_ptr(), v4[2].data_ptr(), v4[3].data_ptr(), part.data_ptr())
            dx2 = torch.empty_like(dx)
            _lib.call('fva_conv_dgrad_bnstats', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx2), ops._p(add), C.byref(fs), ops._stream())
            torch.cuda.synchronize()
            assert torch.isfinite(stats).all() and torch.isfinite(part[:rows]).all()
            out[on] = (y.float(), stats.double().sum(0), dx.float(), dx2.float(), part[:rows].double().sum(0), nblk, rows)
    finally:
        lib.fva_conv_patch_kernel(prev)
    (y0, s0, dx0, dxb0, p0, n0, r0), (y1, s1, dx1, dxb1, p1, n1, r1) = out[0], out[1]
    tiles = B * ((H + 7) // 8) * ((W + 31) // 32)
    assert n1 == tiles and r1 == tiles                                       # the patch kernel's tables: one row per 8 x 32 patch
    for a, b, what in ((y1, y0, 'forward'), (dx1, dx0, 'dgrad + addend'), (dxb1, dxb0, 'dgrad + statistics')):
        # the fp32 sums differ in their last bits, so some stored values round to the neighbouring bf16 (2^-8 relative; the addend
        # path rounds twice): at most a bf16 ulp or two of the tensor's scale anywhere, a small fraction of an ulp on average
        assert rel_err(a, b) < 1e-2, f'{what}: {rel_err(a, b)}'
        assert ((a - b).abs().mean() / b.abs().mean()).item() < 2e-3, what
    assert torch.equal(dx1, dxb1)
    assert rel_err(s1, s0) < 1e-4 and rel_err(p1, p0) < 2e-3


# ---- the kernels the BENCHMARK dispatches, element by element against a non-HIP reference (round 4) -------------------------------
# CONV_CASES above are small: they exercise igemm_kernel (128 x 128 / 256 x 64), the patch kernels and pwgrad_kernel, but the 8-phase
# kernels (igemm8_kernel, wgrad8_kernel) only dispatch from 128 tiles of 256 x 256 on and were compared with other HIP kernels only.
# Here shapes that select them run through the C ABI against ATen's fp32 convolutions on the CPU fed the same bf16-rounded operands,
# 1e-2 of the tensor scale as above, and the library says which kernel ran (fva_conv_last_kernel).
BIG_CASES = [
    # B, Cin, Cout, H, k, stride,  forward,   dgrad,      wgrad
    (32, 128, 256, 80, 3, 1, 'igemm8', 'igemm128', 'wgrad8'),      # 800 tiles; dgrad N = 128 stays on the 128 x 128 kernel; wgrad: Cin = 128, two taps per column tile
    (12, 512, 1024, 40, 3, 1, 'igemm8', 'igemm8', 'wgrad8'),       # 72 k-tiles forward, 144 dgrad; ragged last row block (19200 = 75 x 256)
    (32, 256, 512, 80, 3, 2, 'igemm8', 'igemm8', 'wgrad8'),        # stride 2: forward gathers every other pixel, dgrad = four parity launches
    (7, 256, 512, 72, 3, 1, 'igemm8', 'igemm8', 'wgrad8'),         # M = 36288: not a multiple of 256 (row tail in the 8-phase epilogue and in wgrad's last k-step)
    (32, 256, 128, 80, 1, 1, 'igemm128', 'igemm128', 'wgrad128'),  # the residual blocks' 1x1 at its benchmark size (two k-tiles: below the 8-phase kernel's floor)
]


def _kernel():
    from fastvision_amd import _lib
    return _lib.load().fva_conv_last_kernel().decode()


@pytest.mark.parametrize('case', BIG_CASES)
def test_benchmark_kernels_elementwise_vs_aten(case):
    from fastvision_amd import _lib, ops
    B, Cin, Cout, H, k, s, k_fwd, k_dgrad, k_wgrad = case
    W = H
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    gy = torch.randn(B, Cout, OH, OW, generator=g)
    xr, wr, gyr = rounded(x, 'bf16'), rounded(w, 'bf16'), rounded(gy, 'bf16')
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    want_y = F.conv2d(xr, wr, stride=s, padding=k // 2)
    want_dx = torch.nn.grad.conv2d_input(x.shape, wr, gyr, stride=s, padding=k // 2)
    want_dw = torch.nn.grad.conv2d_weight(xr, w.shape, gyr, stride=s, padding=k // 2)

    lib = _lib.load()
    keep, xptr, xpad = halo(x, dtype)
    d = _lib.ConvDesc(ops._code(dtype), B, H, W, Cin, Cout, k, s, xpad, 1)
    wf, wd = ops.packed_weights(w.to(dev()), d, dtype, cache=False)
    M = B * OH * OW
    y = torch.empty((M, Cout), dtype=dtype, device=dev())
    nblk = lib.fva_conv_stat_blocks(C.byref(d))
    stats = torch.full((lib.fva_bn_partial_rows(nblk), 2, Cout), float('nan'), device=dev())
    _lib.call('fva_conv_fwd', C.byref(d), C.c_void_p(xptr), ops._p(wf), ops._p(y), ops._p(stats), ops._stream())
    ran = {'fwd': _kernel()}
    got_y = y.float().view(B, OH, OW, Cout).permute(0, 3, 1, 2)
    e_y = rel_err(got_y, want_y)
    ref = want_y.permute(0, 2, 3, 1).reshape(M, Cout)
    assert torch.allclose(stats[:nblk, 0].sum(0).cpu(), ref.sum(0), rtol=2e-3, atol=2e-3 * (ref ** 2).sum(0).max().sqrt().item())
    assert torch.allclose(stats[:nblk, 1].sum(0).cpu(), (ref ** 2).sum(0), rtol=2e-3)

    dyk, dyptr, dypad = halo(gy, dtype)
    dx = torch.empty((B, H, W, Cin), dtype=dtype, device=dev())
    _lib.call('fva_conv_dgrad', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx), C.c_void_p(0), ops._stream())
    ran['dgrad'] = _kernel()
    e_dx = rel_err(dx.float().permute(0, 3, 1, 2), want_dx)

    # the same data gradient with the fused BatchNorm-backward statistics of a producer block (fva_conv_dgrad_bnstats): dx must not
    # change by a bit, and the two sums must equal what the formulas give on the STORED (bf16) dx and the producer's y
    yp = torch.randn(B * H * W, Cin, generator=g).to(dev()).to(dtype)
    sc, sh, mu, rs = [(torch.rand(Cin, generator=g) + 0.5).to(dev()) for _ in range(4)]
    rows = lib.fva_conv_dgrad_stat_rows(C.byref(d))
    assert rows > 0
    part = torch.full((lib.fva_bn_partial_rows(rows), 2, Cin), float('nan'), device=dev())
    fs = _lib.BnBwdFuse(yp.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr())
    dx2 = torch.empty_like(dx)
    _lib.call('fva_conv_dgrad_bnstats', C.byref(d), C.c_void_p(dyptr), ops._p(wd), ops._p(dx2), C.c_void_p(0), C.byref(fs), ops._stream())
    ran['dgrad_bnstats'] = _kernel()
    assert torch.equal(dx2.view(torch.int16), dx.view(torch.int16)), 'the fused-statistics launch changed dx'
    dzf, yf = dx.float().view(-1, Cin).double(), yp.double()
    u = yf * sc.double() + sh.double()
    sg = torch.sigmoid(u)
    du = dzf * (sg * (1 + u * (1 - sg)))
    want_s1, want_s2 = du.sum(0), (du * (yf - mu.double()) * rs.double()).sum(0)
    got = part[:rows].double().sum(0)
    scale1 = du.abs().sum(0).max().item()
    assert (got[0] - want_s1).abs().max().item() < 2e-3 * scale1 and (got[1] - want_s2).abs().max().item() < 2e-3 * scale1 * 4

    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=dev())
    wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    # under both split-K plans (fva_conv_wgrad_plan: "alone" = a launch that has the chip to itself, "beside" = what the weight
    # gradients use on the side stream): two summation orders, the same gradient
    prev = lib.fva_conv_wgrad_plan(0)
    try:
        _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(xptr), C.c_void_p(dyptr), ops._p(dw), 0, ops._p(ws), wsb, ops._stream())
        ran['wgrad'] = _kernel()
        lib.fva_conv_wgrad_plan(1)
        dw_b = torch.empty_like(dw)
        _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(xptr), C.c_void_p(dyptr), ops._p(dw_b), 0, ops._p(ws), wsb, ops._stream())
    finally:
        lib.fva_conv_wgrad_plan(prev)
    e_dw, e_dw_b = rel_err(dw, want_dw), rel_err(dw_b, want_dw)
    print(f'{case[:6]}: kernels {ran}; max err / scale: y {e_y:.2e}, dx {e_dx:.2e}, dW {e_dw:.2e} (plan "beside" {e_dw_b:.2e}, '
          f'bit-equal to "alone": {torch.equal(dw, dw_b)})')
    assert (ran['fwd'], ran['wgrad']) == (k_fwd, k_wgrad), ran
    assert ran['dgrad'] == k_dgrad and ran['dgrad_bnstats'] == k_dgrad, ran
    assert e_y < TOL['bf16'] and e_dx < TOL['bf16'] and e_dw < TOL['bf16'] and e_dw_b < TOL['bf16']


def test_wgrad_plan_is_a_process_setting_not_a_stream_property():
    """the split-K plan (and with it the fp32 summation order of dW) used to follow the stream a launch was given.
    It is now a sticky process-wide setting: the same plan gives the same bits on the launch stream and on the library's side stream."""
    from fastvision_amd import _lib, ops
    lib = _lib.load()
    B, Cin, Cout, H = 32, 128, 256, 80
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(dev()).to(torch.bfloat16)
    dy = torch.randn(B, H + 2, H + 2, Cout, generator=g).to(dev()).to(torch.bfloat16)
    for t in (x, dy):
        t[:, 0], t[:, -1], t[:, :, 0], t[:, :, -1] = 0, 0, 0, 0
    d = _lib.ConvDesc(ops._code(torch.bfloat16), B, H, H, Cin, Cout, 3, 1, 1, 1)
    wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    out = {}
    prev = lib.fva_conv_wgrad_plan(-1)
    try:
        for plan in (0, 1):
            lib.fva_conv_wgrad_plan(plan)
            for where in ('main', 'side'):
                dw = torch.empty(Cout, Cin, 3, 3, device=dev())
                if where == 'side':
                    side = C.c_void_p()
                    _lib.call('fva_side_stream_fork', ops._stream(), C.byref(side))
                    _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, side)
                    _lib.call('fva_side_stream_join', ops._stream())
                else:
                    _lib.call('fva_conv_wgrad', C.byref(d), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), wsb, ops._stream())
                torch.cuda.synchronize()
                out[plan, where] = dw
    finally:
        lib.fva_conv_wgrad_plan(prev)
    assert torch.equal(out[0, 'main'], out[0, 'side']) and torch.equal(out[1, 'main'], out[1, 'side'])
    assert not torch.equal(out[0, 'main'], out[1, 'main'])         # this shape splits 14 ways alone and 7 ways beside: another order ...
    assert rel_err(out[0, 'main'], out[1, 'main']) < 1e-5          # ... of the same sum


# ---- the forward apply pass fused into the 1x1 convolution that consumes it (fva_conv1x1_fwd_apply, round 4) ---------------------------
@pytest.mark.parametrize('case', [
    # B, C (= Cin of the 1x1 layer), N (= its Cout), H, W, residual
    (2, 64, 32, 20, 24, True),        # thin tile (256 x 64), one k-tile
    (3, 128, 64, 17, 9, True),        # thin tile, two k-tiles, M = 459: a row tail
    (2, 256, 128, 16, 16, True),      # 128 x 128 tile, four k-tiles
    (1, 256, 128, 13, 11, False),     # no identity (the down-sampling convolution feeding a stage's first block)
    (32, 256, 128, 80, 80, True),     # the benchmark's res3 shape
])
def test_conv1x1_fused_apply_is_bit_identical_with_the_two_launches(case):
    """z (incl. its zero border), the 1x1 layer's output and its BatchNorm partial statistics from ONE fused launch must equal, bit for
    bit, what fva_bn_silu_apply followed by fva_conv_fwd write."""
    from fastvision_amd import _lib, ops
    B, Cc, N, H, W, with_res = case
    lib = _lib.load()
    dt = torch.bfloat16
    code = ops._code(dt)
    g = torch.Generator().manual_seed(Cc + H)
    M = B * H * W
    y_prev = torch.randn(M, Cc, generator=g).to(dev()).to(dt)
    sc = (torch.rand(Cc, generator=g) + 0.5).to(dev())
    sh = (torch.rand(Cc, generator=g) - 0.5).to(dev())
    res = torch.randn(B, H + 2, W + 2, Cc, generator=g).to(dev()).to(dt) if with_res else None
    w = (torch.randn(N, Cc, 1, 1, generator=g) / Cc ** 0.5).to(dev())
    d = _lib.ConvDesc(code, B, H, W, Cc, N, 1, 1, 1, 1)
    wf, _ = ops.packed_weights(w, d, dt, cache=False)
    nblk = lib.fva_conv_stat_blocks(C.byref(d))
    rows = lib.fva_bn_partial_rows(nblk)
    st = ops._stream()
    rp = ops._p(res) if with_res else C.c_void_p(0)
    # reference: two launches
    z0 = torch.full((B, H + 2, W + 2, Cc), float('nan'), device=dev(), dtype=dt)
    y0 = torch.empty(M, N, device=dev(), dtype=dt)
    s0 = torch.zeros(rows, 2, N, device=dev())
    _lib.call('fva_bn_silu_apply', code, ops._p(y_prev), ops._p(sc), ops._p(sh), rp, 1, ops._p(z0), 1, B, H, W, Cc, st)
    _lib.call('fva_conv_fwd', C.byref(d), ops._p(z0), ops._p(wf), ops._p(y0), ops._p(s0), st)
    k_plain = lib.fva_conv_last_kernel().decode()
    # fused
    z1 = torch.full_like(z0, float('nan'))
    y1 = torch.empty_like(y0)
    s1 = torch.zeros_like(s0)
    _lib.call('fva_conv1x1_fwd_apply', C.byref(d), ops._p(y_prev), ops._p(sc), ops._p(sh), rp, 1, ops._p(z1), ops._p(wf), ops._p(y1), ops._p(s1), st)
    k_fused = lib.fva_conv_last_kernel().decode()
    torch.cuda.synchronize()
    assert k_fused == k_plain + 'ax', (k_plain, k_fused)
    assert not torch.isnan(z1.float()).any(), 'part of z (or of its border) was not written'
    assert torch.equal(z1.view(torch.int16), z0.view(torch.int16))
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16))
    assert torch.equal(s1[:nblk], s0[:nblk])


def test_deferred_apply_leaves_the_training_step_bit_identical():
    """A whole YOLOv3 train step (B = 2, 128 px, bf16) with the deferred / fused forward apply on and off: identical loss, head outputs
    and gradients, and the fused launch really ran (11 of Darknet-53's residual conv1 layers and two 1x1 layers of the 80 x 80 neck
    block have Cout <= 128)."""
    import fastvision_amd
    from fastvision_amd import ops
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    images, tg = synthetic_batch(2, 128)
    out = {}
    with fastvision_amd.compute_dtype(torch.bfloat16):
        for on in (False, True):
            prev = ops.set_apply_fusion(on)
            try:
                torch.manual_seed(5)
                net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                             training=True).to(dev()).train()
                crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
                n0 = ops._DEFER['fused']
                pred = net(images.to(dev()))
                fused = ops._DEFER['fused'] - n0
                loss = crit(pred, tg.to(dev()))
                loss.backward()
                torch.cuda.synchronize()
                out[on] = (loss.detach().clone(), [p.detach().clone() for p in pred], [p.grad.clone() for p in net.parameters()],
                           [b.clone() for b in net.buffers()], fused)
            finally:
                ops.set_apply_fusion(prev)
    assert out[False][4] == 0 and out[True][4] == 13, (out[False][4], out[True][4])
    assert torch.equal(out[False][0], out[True][0])
    for a, b in zip(out[False][1] + out[False][2] + out[False][3], out[True][1] + out[True][2] + out[True][3]):
        assert torch.equal(a, b)


def test_saved_buffers_are_released_by_backward():
    """Round 4 (found by the stall watchdog): the autograd nodes used to keep their saved activations for as long as anything referenced
    the graph -- a caller holding the attached loss of step k while issuing step k + 1 kept two steps' activations alive.  They are
    released by the backward pass that used them, like torch's own saved tensors; a second pass through the same graph raises."""
    import fastvision_amd
    from fastvision_amd.classfication.models.darknet53 import ResidualBlock
    with fastvision_amd.compute_dtype(torch.bfloat16):
        blk = ResidualBlock(64, 32).to(dev()).train()
        x = torch.randn(2, 64, 16, 16, device=dev(), requires_grad=True)
        y = blk(x)
        gy = torch.randn_like(y.float())
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        torch.autograd.grad(y, x, gy.to(y.dtype), retain_graph=True)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() < before, 'the node still holds its saved buffers after backward'
        with pytest.raises(RuntimeError, match='already been released'):
            torch.autograd.grad(y, x, gy.to(y.dtype))
