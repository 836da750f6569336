"""Fully connected layers and small losses of the two-stage head on the library's kernels (fc_ops) against plain
PyTorch fp32 statements of the same ops (the reference uses nn.Linear / F.cross_entropy / F.smooth_l1_loss and its FocalLoss:
demos/faster_rcnn/models/vgg.py:41-47, fast.py:47-52,192,197, rpn.py:8-64), and the inference post-processing of
demos/faster_rcnn/inference.py:87-118 against a restatement on the CPU with the oracle's NMS."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('shape', [(70, 512, 256), (256, 25088, 4096), (3, 4096, 4096)])
def test_linear_relu_forward_backward(shape, dtype, tol):
    from fastvision_amd.fc_ops import linear_relu
    R, K, N = shape
    g = torch.Generator().manual_seed(R)
    lin = torch.nn.Linear(K, N)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g) / K ** 0.5)
        lin.bias.copy_(torch.randn(N, generator=g) * 0.1)
    x = torch.randn(R, K, generator=g)
    gy = torch.randn(R, N, generator=g)
    q = (lambda t: t.to(dtype).float()) if dtype == torch.bfloat16 else (lambda t: t)
    xr = q(x).to(DEV).requires_grad_(True)
    wr = q(lin.weight.detach()).to(DEV).requires_grad_(True)
    br = lin.bias.detach().to(DEV).requires_grad_(True)
    want = F.relu(F.linear(xr, wr, br))
    want.backward(q(gy).to(DEV))
    lin = lin.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    got = linear_relu(xd, lin, dtype)
    assert got.dtype == dtype and tuple(got.shape) == (R, N)
    got.backward(gy.to(DEV).to(dtype))
    torch.cuda.synchronize()
    assert rel(got.float(), want) < tol
    assert rel(xd.grad, xr.grad) < tol and rel(lin.weight.grad, wr.grad) < tol and rel(lin.bias.grad, br.grad) < tol


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_linear_head_forward_backward(dtype, tol):
    from fastvision_amd.fc_ops import linear
    R, K, N = 200, 4096, 21
    g = torch.Generator().manual_seed(1)
    lin = torch.nn.Linear(K, N)
    x = torch.randn(R, K, generator=g)
    gy = torch.randn(R, N, generator=g)
    q = (lambda t: t.to(dtype).float()) if dtype == torch.bfloat16 else (lambda t: t)
    xr = q(x).to(DEV).requires_grad_(True)
    wr = q(lin.weight.detach()).to(DEV).requires_grad_(True)
    br = lin.bias.detach().to(DEV).requires_grad_(True)
    want = F.linear(xr, wr, br)
    want.backward(gy.to(DEV))
    lin = lin.to(DEV)
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    got = linear(xd, lin, dtype)
    assert got.dtype == torch.float32 and tuple(got.shape) == (R, N)
    got.backward(gy.to(DEV))
    assert rel(got, want) < tol and rel(xd.grad.float(), xr.grad) < tol
    assert rel(lin.weight.grad, wr.grad) < tol and rel(lin.bias.grad, br.grad) < 1e-4


def test_row_losses_and_smooth_l1_vs_torch():
    from fastvision_amd.fc_ops import cross_entropy_mean, focal_mean, smooth_l1_mean
    g = torch.Generator().manual_seed(3)
    for R, Cc in ((300, 21), (512, 2), (1, 5)):
        z = torch.randn(R, Cc, generator=g) * 3
        y = torch.randint(0, Cc, (R,), generator=g)
        zr = z.clone().requires_grad_(True)
        F.cross_entropy(zr, y).backward()
        zd = z.to(DEV).requires_grad_(True)
        got = cross_entropy_mean(zd, y.to(DEV))
        got.backward()
        assert abs(got.item() - F.cross_entropy(z, y).item()) < 1e-5 * max(1.0, abs(got.item())) and rel(zd.grad, zr.grad) < 1e-4
        zr = z.clone().requires_grad_(True)
        p = torch.softmax(zr, 1).gather(1, y.view(-1, 1))
        want = (-torch.pow(1 - p, 2) * p.log()).mean()
        want.backward()
        zd = z.to(DEV).requires_grad_(True)
        got = focal_mean(zd, y.to(DEV), 2.0)
        got.backward()
        assert abs(got.item() - want.item()) < 1e-5 * max(1.0, abs(want.item())) and rel(zd.grad, zr.grad) < 1e-4
    a, b = torch.randn(77, 4, generator=g) * 2, torch.randn(77, 4, generator=g)
    ar = a.clone().requires_grad_(True)
    F.smooth_l1_loss(ar, b).backward()
    ad = a.to(DEV).requires_grad_(True)
    got = smooth_l1_mean(ad, b.to(DEV))
    (got * 3).backward()
    assert abs(got.item() - F.smooth_l1_loss(a, b).item()) < 1e-6 and rel(ad.grad, 3 * ar.grad) < 1e-5
    assert float(smooth_l1_mean(torch.zeros(0, 4, device=DEV), torch.zeros(0, 4, device=DEV))) == 0.0


def test_faster_rcnn_inference_post_processing_vs_restatement():
    from fastvision_amd.demos.faster_rcnn.inference import anchor_fn, postProcess, preProcess
    from oracle import detect as OD, pipeline as OP
    g = torch.Generator().manual_seed(5)
    n = 400
    xy = torch.rand(n, 2, generator=g) * torch.tensor([40.0, 30.0])
    wh = torch.exp(torch.rand(n, 2, generator=g) * 2.5) * 0.6
    prop = torch.cat([xy, wh, torch.randint(0, 20, (n, 1), generator=g).float(), torch.rand(n, 1, generator=g)], 1)
    args = types.SimpleNamespace(backbone_stride=16, inference_conf_thres=0.3, inference_iou_thres=0.5)
    rr, pl, pt, ow, oh = 0.8, 12, 40, 700, 500
    scores, cats, boxes = postProcess(prop.to(DEV), args, rr, pl, pt, ow, oh)
    # the reference's steps (inference.py:87-118) on the CPU, NMS from the oracle
    p = prop.clone()
    p[:, 0:4] *= 16
    p[:, 0] = ((p[:, 0] - pl) / rr).clamp(0, ow - 1); p[:, 1] = ((p[:, 1] - pt) / rr).clamp(0, oh - 1)
    p[:, 2] = (p[:, 2] / rr).clamp(0, ow); p[:, 3] = (p[:, 3] / rr).clamp(0, oh)
    p = p[(p[:, 2] > 5) & (p[:, 3] > 5)]
    b = torch.cat([p[:, :2] - p[:, 2:4] / 2, p[:, :2] + p[:, 2:4] / 2], 1)
    b[:, [0, 2]] = b[:, [0, 2]].clamp(0, ow - 1); b[:, [1, 3]] = b[:, [1, 3]].clamp(0, oh - 1)
    p[:, :4] = b
    p = p[p[:, 5] > 0.3]
    keep = OD.nms(p[:, :4] + p[:, 4:5] * 4096, p[:, 5], 0.5)[:300]
    want = p[keep]
    assert boxes.size(0) == want.size(0) > 10
    np.testing.assert_allclose(boxes.cpu().numpy(), want[:, :4].numpy(), rtol=1e-6, atol=1e-4)
    assert torch.equal(cats.cpu().view(-1), want[:, 4]) and torch.allclose(scores.cpu().view(-1), want[:, 5])
    # preProcess: ResizeByMax + Padding(128) + / 255 on the device against the oracle's pipeline
    img = np.random.RandomState(1).randint(0, 256, (90, 140, 3)).astype(np.uint8)
    image, ori, ratio, left, top, h0, w0 = preProcess(img, 128, DEV)
    small, _ = OP.demo_resize_by_max(img, np.zeros((0, 4), np.float32), 128)
    canvas, _ = OP.demo_padding(small, np.zeros((0, 4), np.float32), 128, fill_value=128)
    want_img = np.transpose(canvas, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    assert (ratio, left, top, h0, w0) == (128 / 140, 0, (128 - int(90 * 128 / 140)) // 2, 90, 140)
    assert np.array_equal(image[0].cpu().numpy(), want_img)
    assert tuple(anchor_fn([128, 256, 512], [0.5, 1, 2]).shape) == (9, 2)
