"""RoIAlign HIP kernel (scope row f-4) against the CPU restatement, through the C ABI (fastvision_amd.roi_ops)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _boxes(rng, K, B, H, W):
    x1 = rng.uniform(-3, W, K)
    y1 = rng.uniform(-3, H, K)
    w = np.exp(rng.uniform(np.log(0.05), np.log(W * 1.2), K))
    h = np.exp(rng.uniform(np.log(0.05), np.log(H * 1.2), K))
    b = rng.integers(0, B, K)
    return np.stack([b, x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 70, 13, 17, (7, 7)), (3, 64, 9, 9, (2, 3)), (1, 130, 25, 19, (8, 8))])
def test_forward_and_backward_match_oracle(dtype, shape):
    from fastvision_amd.roi_ops import roi_align
    from oracle import roi_align as R
    B, C, H, W, osz = shape
    rng = np.random.default_rng(B * 100 + C)
    feat = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).to(dtype)
    boxes = _boxes(rng, 37, B, H, W)
    boxes[0, 1:] = [2.0, 2.0, 2.0, 2.0]                       # degenerate: widened to 1 x 1
    boxes[1, 1:] = [W + 2.0, H + 2.0, W + 5.0, H + 4.0]       # fully outside: zeros
    boxes[2, 1:] = [-1.0, -1.0, W + 1.0, H + 1.0]             # larger than the map
    f_dev = feat.to(DEV).requires_grad_(True)
    out = roi_align(f_dev, torch.from_numpy(boxes).to(DEV), osz)
    want = R.roi_align(feat.float().numpy(), boxes, osz)
    assert out.shape == want.shape and out.dtype == torch.float32
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    assert torch.all(out[1] == 0)
    g = torch.from_numpy(rng.standard_normal(want.shape).astype(np.float32))
    out.backward(g.to(DEV))
    dwant = R.roi_align_backward(g.numpy(), boxes, (B, C, H, W))
    got = f_dev.grad.float().cpu().numpy()
    tol = 1e-4 if dtype == torch.float32 else 2e-2            # bf16 features: the gradient is rounded to bf16 on return
    assert np.abs(got - dwant).max() <= tol * max(1.0, np.abs(dwant).max())


def test_halo_view_input_is_read_in_place_and_empty_box_list():
    from fastvision_amd import ops
    from fastvision_amd.roi_ops import roi_align
    from oracle import roi_align as R
    rng = np.random.default_rng(5)
    B, C, H, W = 2, 64, 10, 12
    buf, view = ops.halo_alloc(B, C, H, W, torch.float32, torch.device(DEV), 1)
    buf.normal_(generator=torch.Generator(device=DEV).manual_seed(1))
    assert ops.halo_info(view, torch.float32) is not None
    boxes = _boxes(rng, 11, B, H, W)
    out = roi_align(view, torch.from_numpy(boxes).to(DEV), (7, 7))
    want = R.roi_align(view.cpu().numpy(), boxes, (7, 7))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    empty = roi_align(view, torch.zeros((0, 5), device=DEV), (7, 7))
    assert tuple(empty.shape) == (0, C, 7, 7)
    with pytest.raises(RuntimeError):
        roi_align(view.cpu(), torch.zeros((1, 5)), (7, 7))                    # no CPU path


def test_reference_size_adjoint_property_and_timing():
    """VGG16 stride-16 map of an 800x608 image, 512 channels, 4 images x 64 sampled boxes (fast.py: 16 + 48 per image):
    too big for the Python oracle, so the size-independent property is checked -- <roi_align(f), g> == <f, roi_align^T(g)>."""
    from fastvision_amd.roi_ops import roi_align
    rng = np.random.default_rng(9)
    B, C, H, W, K = 4, 512, 38, 50, 256
    feat = torch.randn(B, C, H, W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).requires_grad_(True)
    boxes = torch.from_numpy(_boxes(rng, K, B, H, W)).to(DEV)
    out = roi_align(feat, boxes, (7, 7))
    g = torch.randn(out.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    out.backward(g)
    lhs, rhs = (out.double() * g.double()).sum().item(), (feat.grad.double() * feat.detach().double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.no_grad():
        roi_align(feat, boxes, (7, 7))
        ev[0].record()
        for _ in range(10):
            roi_align(feat, boxes, (7, 7))
        ev[1].record()
    from fastvision_amd import ops
    buf, view = ops.halo_alloc(B, C, H, W, torch.float32, torch.device(DEV), 1)
    view.copy_(feat.detach())
    with torch.no_grad():
        ref = roi_align(feat, boxes, (7, 7))
        assert torch.equal(roi_align(view, boxes, (7, 7)), ref)                # halo view read in place: same bits
        ev[2].record()
        for _ in range(10):
            roi_align(view, boxes, (7, 7))
        ev[3].record()
    torch.cuda.synchronize()
    print(f'roi_align fwd {B}x{C}x{H}x{W}, K={K}: {ev[0].elapsed_time(ev[1]) * 100:.1f} us per call incl. the NCHW -> NHWC pack, '
          f'{ev[2].elapsed_time(ev[3]) * 100:.1f} us on a halo NHWC view (25.7 MB written)')


def test_non_finite_and_huge_boxes_terminate():
    """A box with an infinite or absurd extent (user input; the RPN clamps its own proposals) must neither spin nor fault: the sampling
    grid is bounded at 64 x 64 per bin.  Finite rows of the same call are unaffected."""
    from fastvision_amd.roi_ops import roi_align
    from oracle import roi_align as R
    rng = np.random.default_rng(9)
    feat = torch.from_numpy(rng.standard_normal((1, 64, 12, 12)).astype(np.float32))
    boxes = np.array([[0, 1.0, 1.0, 8.0, 9.0], [0, 0.0, 0.0, float('inf'), 5.0], [0, 2.0, 2.0, 1e30, 1e30], [0, float('nan'), 0.0, 4.0, 4.0]], np.float32)
    out = roi_align(feat.to(DEV), torch.from_numpy(boxes).to(DEV), (7, 7))
    torch.cuda.synchronize()
    want = R.roi_align(feat.numpy(), boxes[:1], (7, 7))
    np.testing.assert_allclose(out[0].cpu().numpy(), want[0], rtol=1e-5, atol=2e-5)
    assert out.shape == (4, 64, 7, 7)
    with pytest.raises(ValueError):
        roi_align(feat.to(DEV), torch.from_numpy(boxes[:1]).to(DEV), (7, 7), sampling_ratio=100)
