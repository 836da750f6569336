"""VGG blocks of the two-stage head's backbone (scope row f-4: Conv2d + bias -> ReLU, MaxPool2d(2, 2)) on the HIP kernels against
plain PyTorch fp32 on the CPU (the numerics reference for a floating-point kernel), through the C ABI (fastvision_amd.vgg_ops)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


class Mini(nn.Module):
    """first two VGG16 stages and the start of the third: conv-conv-pool-conv-conv-pool-conv"""

    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(3, 64, 3, 1, 1), nn.Conv2d(64, 64, 3, 1, 1), nn.Conv2d(64, 128, 3, 1, 1), nn.Conv2d(128, 128, 3, 1, 1),
                                    nn.Conv2d(128, 256, 3, 1, 1)])

    def forward(self, x, conv_relu, pool):
        x = conv_relu(x, self.convs[0])
        x = pool(conv_relu(x, self.convs[1]))
        x = conv_relu(x, self.convs[2])
        x = pool(conv_relu(x, self.convs[3]))
        return conv_relu(x, self.convs[4])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_vgg_blocks_forward_backward_vs_torch(dtype):
    import fastvision_amd
    from fastvision_amd.vgg_ops import conv_bias_relu, max_pool2
    torch.manual_seed(3)
    ref = Mini()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 38, 50, generator=g)                 # odd sizes after the second pool (9 x 12): floor mode
    if dtype == torch.float32:
        conv_ref = lambda t, c: F.relu(c(t))
    else:
        # bf16 path: the same fp32 arithmetic with the operands the kernels see -- filters and activations rounded to bf16 where
        # they are stored (the casts round the gradients at the same points on the way back)
        r = lambda t: t.bfloat16().float()
        conv_ref = lambda t, c: r(F.relu(F.conv2d(r(t), r(c.weight), c.bias, 1, 1)))
    out_ref = ref(x, conv_ref, lambda t: F.max_pool2d(t, 2, 2))
    gout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gout)
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    dev_net = Mini().to(DEV)
    dev_net.load_state_dict(ref.state_dict())
    with fastvision_amd.compute_dtype(dtype):
        out = dev_net(x.to(DEV), conv_bias_relu, max_pool2)
        assert tuple(out.shape) == tuple(out_ref.shape)
        out.backward(gout.to(DEV).to(out.dtype))
    # fp32: max-norm agreement with plain fp32.  bf16: against the bf16-operand emulation above, in the L2 norm (a value that
    # rounds the other way flips a ReLU mask or a pooling arg-max and reroutes single gradient entries)
    def rel2(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-12))
    err = rel if dtype == torch.float32 else rel2
    tol_f, tol_g = (1e-4, 2e-4) if dtype == torch.float32 else (5e-3, 1e-2)
    assert err(out.float().cpu(), out_ref.detach()) < tol_f
    errs = {}
    for k, p in dev_net.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        errs[k] = err(p.grad.cpu(), want[k])
    print(dtype, {k: round(v, 5) for k, v in errs.items()})
    assert max(errs.values()) < tol_g, errs


def test_maxpool_ties_and_odd_sizes_match_torch():
    """gradient routing on ties (first maximum in scan order) and floor-mode sizes, exactly"""
    import fastvision_amd
    from fastvision_amd.vgg_ops import max_pool2
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 3, (2, 8, 7, 9), generator=g).float()   # many ties; H, W odd: last row / column dropped
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    with fastvision_amd.compute_dtype(torch.float32):
        xd = x.to(DEV).requires_grad_(True)
        yd = max_pool2(xd)
        yd.backward(gy.to(DEV))
    assert torch.equal(yd.cpu(), yr.detach()) and torch.equal(xd.grad.cpu(), xr.grad)


def test_bias_relu_backward_kernel_exact_fp32():
    from fastvision_amd import _lib, ops
    import ctypes as C
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    B, H, W, Cc = 3, 5, 7, 64
    z = torch.randn(B, H + 2, W + 2, Cc, generator=g).to(DEV)
    z[:, 0], z[:, -1], z[:, :, 0], z[:, :, -1] = 0, 0, 0, 0
    dz = torch.randn(B, H, W, Cc, generator=g).to(DEV)
    dy = torch.full((B, H + 2, W + 2, Cc), float('nan'), device=DEV)
    rows = lib.fva_bias_relu_bwd_rows(B, H, 1)
    part = torch.empty(rows, Cc, device=DEV)
    _lib.call('fva_bias_relu_bwd', 0, ops._p(dz), ops._p(z), 1, ops._p(dy), 1, ops._p(part), B, H, W, Cc, ops._stream())
    db = torch.empty(Cc, device=DEV)
    _lib.call('fva_colsum', ops._p(part), rows, Cc, ops._p(db), None, ops._stream())
    big = torch.randn(1000, Cc, device=DEV)                       # the two-pass form
    scratch = torch.empty(lib.fva_colsum_scratch_rows(1000), Cc, device=DEV)
    db2 = torch.empty(Cc, device=DEV)
    _lib.call('fva_colsum', ops._p(big), 1000, Cc, ops._p(db2), ops._p(scratch), ops._stream())
    assert torch.allclose(db2, big.double().sum(0).float(), rtol=1e-5, atol=1e-4)
    want = dz * (z[:, 1:-1, 1:-1] > 0)
    assert torch.equal(dy[:, 1:-1, 1:-1], want)
    assert (dy[:, 0] == 0).all() and (dy[:, -1] == 0).all() and (dy[:, :, 0] == 0).all() and (dy[:, :, -1] == 0).all()
    assert torch.allclose(db, want.sum((0, 1, 2)), rtol=1e-5, atol=1e-5)


def test_vgg16_backbone_keys_numerics_and_timing():
    """The demo's backbone (fastvision_amd.demos.faster_rcnn.models.vgg16): reference state_dict keys, fp32 forward/backward vs a
    plain torch stack with the same weights at a small size, and the bf16 forward+backward time at the reference's training shape."""
    import time
    import fastvision_amd
    from fastvision_amd.demos.faster_rcnn.models import vgg16
    torch.manual_seed(0)
    net = vgg16().to(DEV)
    convs = [f'vgg{s}.{2 * i}' for s, n in zip(range(1, 6), (2, 2, 3, 3, 3)) for i in range(n)]
    want_keys = [f'{c}.{p}' for c in convs for p in ('weight', 'bias')] + [f'classifier.{i}.{p}' for i in (0, 3) for p in ('weight', 'bias')]
    assert list(net.state_dict().keys()) == want_keys
    feats = [m for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(1))

    def torch_stack(t):
        k = 0
        for stage, n in enumerate((2, 2, 3, 3, 3)):
            for _ in range(n):
                t = F.relu(F.conv2d(t, feats[k].weight.detach().cpu(), feats[k].bias.detach().cpu(), 1, 1))
                k += 1
            if stage < 4:
                t = F.max_pool2d(t, 2, 2)
        return t
    want = torch_stack(x)
    with fastvision_amd.compute_dtype(torch.float32):
        got = net(x.to(DEV))
        assert tuple(got.shape) == (2, 512, 4, 6)
        assert rel(got.float().cpu(), want) < 2e-4
        got.square().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in net.named_parameters() if n.startswith('vgg'))
    with fastvision_amd.compute_dtype(torch.bfloat16):
        xb = torch.randn(4, 3, 608, 800, device=DEV)
        for i in range(4):
            if i == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            net.zero_grad(set_to_none=True)
            out = net(xb)
            out.float().square().mean().backward()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
    assert tuple(out.shape) == (4, 512, 38, 50)
    flop = 2 * 3 * 4 * sum(m.weight.numel() * hw for m, hw in zip(feats, [608 * 800] * 2 + [304 * 400] * 2 + [152 * 200] * 3 + [76 * 100] * 3 + [38 * 50] * 3))
    print(f'VGG16 backbone 4x3x608x800 bf16 forward+backward: {ms:.1f} ms ({flop / ms / 1e9:.0f} TFLOP/s of conv work, fwd + dgrad + wgrad)')
