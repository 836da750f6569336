"""Pin the CPU oracle (oracle/) against golden vectors captured from the reference itself.

The fixtures in tests/golden/{lib,demo}.npz were produced by oracle/make_golden.py, which imports and
runs the real reference in the build container.  Integer outputs must be bit-exact; fp32 values are
compared at 1e-6 relative (same torch CPU kernels underneath) unless stated otherwise.
"""
import numpy as np
import pytest
import torch

from fastvision_amd.synthetic import synthetic_batch
from oracle import boxes, losses, model as omodel, train as otrain

RT = dict(rtol=1e-6, atol=1e-7)


def T(a):
    return torch.from_numpy(np.asarray(a))


def stats(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item()] + t[:4].tolist() + [0.0] * max(0, 4 - t.numel()))


def assert_stats_close(got, want, rel=1e-4):
    """Rows are (sum, abs-sum, first four values).  The plain sum cancels, so it is judged against abs-sum."""
    scale = np.maximum(want[:, 1:2], 1e-30)
    np.testing.assert_array_less(np.abs(got[:, 0:1] - want[:, 0:1]) / scale, rel)
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=rel, atol=1e-12)
    np.testing.assert_allclose(got[:, 2:], want[:, 2:], rtol=10 * rel, atol=1e-6)


# ------------------------------------------------------------------------------------------- G1 matcher
def _check_match(gold, prefix, grids_hw, batch):
    tg = T(gold[f'{prefix}_targets'])
    shapes = [(batch, 3, h, w, 85) for h, w in grids_hw]
    anc = [a for a in omodel.coco_anchors_px().view(3, 3, 1, 1, 2)]
    locs, cats, xywh, matched = losses.build_target(shapes, tg, anc, omodel.LEVEL_STRIDES)
    for l in range(3):
        assert np.array_equal(locs[l][0].numpy(), gold[f'{prefix}_l{l}_b'])
        assert np.array_equal(locs[l][1].numpy(), gold[f'{prefix}_l{l}_gxy'])
        assert np.array_equal(locs[l][2].numpy(), gold[f'{prefix}_l{l}_a'])
        assert np.array_equal(cats[l].numpy(), gold[f'{prefix}_l{l}_cls'])
        assert np.array_equal(xywh[l].numpy(), gold[f'{prefix}_l{l}_xywh'])       # same fp32 op order: exact
        assert np.array_equal(matched[l].numpy(), gold[f'{prefix}_l{l}_anc'])


def test_g1_build_target_bit_exact(gold_lib):
    n = int(gold_lib['g1_cases'])
    assert n == 45
    total = 0
    for c in range(n):
        grids = gold_lib[f'g1_{c}_grids']
        _check_match(gold_lib, f'g1_{c}', [(int(s), int(s)) for s in grids], 8)
        total += sum(len(gold_lib[f'g1_{c}_l{l}_b']) for l in range(3))
    assert total > 1000          # the cases do exercise matches


def test_g1_build_target_non_square(gold_lib):
    _check_match(gold_lib, 'g1ns', [(20, 15), (40, 30), (80, 60)], 2)


# ------------------------------------------------------------------------------------------- G2 IoU family
def test_g2_iou_family(gold_lib):
    a, b = T(gold_lib['g2_a']), T(gold_lib['g2_b'])
    wa, wb = a[:, 2:] - a[:, :2], b[:, 2:] - b[:, :2]
    xa, xb = boxes.xyxy2xywh(a), boxes.xyxy2xywh(b)
    chk = lambda got, key: np.testing.assert_allclose(got.numpy(), gold_lib[key], **RT)
    chk(boxes.xyxy_iou(a, b), 'g2_xyxy_iou')
    chk(boxes.xywh_iou(xa, xb), 'g2_xywh_iou')
    chk(boxes.wh_iou(wa, wb), 'g2_wh_iou')
    chk(boxes.xyxy_iou_batch(a[:40], b[:24]), 'g2_xyxy_iou_batch')
    chk(boxes.xywh_iou_batch(xa[:40], xb[:24]), 'g2_xywh_iou_batch')
    chk(boxes.wh_iou_batch(wa[:40], wb[:24]), 'g2_wh_iou_batch')
    chk(boxes.GIOU(a, b), 'g2_giou')
    chk(boxes.DIOU(a, b), 'g2_diou')
    chk(boxes.CIOU(a, b), 'g2_ciou')
    chk(boxes.CIOU(xa, xb, mode='xywh'), 'g2_ciou_xywh')
    chk(boxes.xywh2xyxy(xa), 'g2_xywh2xyxy')
    chk(boxes.xyxy2xywhn(a, 480, 640), 'g2_xyxy2xywhn')
    assert np.array_equal(boxes.grid(3, 5, 'xy').numpy(), gold_lib['g2_grid_xy'])
    assert np.array_equal(boxes.grid(3, 5, 'yx').numpy(), gold_lib['g2_grid_yx'])
    # quirks really are reproduced: DIoU > IoU for displaced boxes (App. B-2)
    assert (boxes.DIOU(a, b) >= boxes.xyxy_iou(a, b)).all()


def test_g2_ciou_loss_and_bce(gold_lib):
    a, b = T(gold_lib['g2_a']).clone().requires_grad_(True), T(gold_lib['g2_b'])
    l = losses.ciou_loss(a, b)
    np.testing.assert_allclose(l.detach().numpy().reshape(1), gold_lib['g2_cioul'], **RT)
    l.backward()
    np.testing.assert_allclose(a.grad.numpy(), gold_lib['g2_cioul_grad'], rtol=1e-5, atol=1e-7)
    p, lab = T(gold_lib['g2_bce_p']), T(gold_lib['g2_bce_lab'])
    np.testing.assert_allclose(losses.bce_probs(p, lab).numpy().reshape(1), gold_lib['g2_bce_mean'], **RT)
    np.testing.assert_allclose(losses.bce_probs((p * 4 - 2).sigmoid(), lab, 'sum').numpy().reshape(1),
                               gold_lib['g2_bce_logits_sum'], **RT)


def test_g2_demo_iou_variants(gold_demo):
    a, b = T(gold_demo['g2_a']), T(gold_demo['g2_b'])
    np.testing.assert_allclose(boxes.DIOU(a, b, demo=True).numpy(), gold_demo['g2_diou'], **RT)
    np.testing.assert_allclose(boxes.CIOU(a, b, demo=True).numpy(), gold_demo['g2_ciou'], **RT)


# ------------------------------------------------------------------------------------------- G3 losses
@pytest.mark.parametrize('tag', ['rand', 'empty', 'dup', 'syn'])
def test_g3_library_loss_and_head_grads(gold_lib, tag):
    tg = T(gold_lib[f'g3_{tag}_targets'])
    heads = [T(gold_lib[f'g3_{tag}_head{l}']).clone().requires_grad_(True) for l in range(3)]
    anc = [a for a in omodel.coco_anchors_px().view(3, 3, 1, 1, 2)]
    loss = losses.yolov3_loss(heads, tg, anc, omodel.LEVEL_STRIDES, 0.05, 1.0, 0.5)
    np.testing.assert_allclose(loss.detach().numpy(), gold_lib[f'g3_{tag}_loss'], **RT)
    loss.backward()
    for l in range(3):
        np.testing.assert_allclose(heads[l].grad.numpy(), gold_lib[f'g3_{tag}_grad{l}'], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('tag', ['syn', 'syn4', 'dup'])
def test_g3_demo_loss_and_head_grads(gold_demo, tag):
    tg = T(gold_demo[f'g3_{tag}_targets'])
    heads = [T(gold_demo[f'g3_{tag}_head{l}']).clone().requires_grad_(True) for l in range(3)]
    loss, parts = losses.demo_loss(heads, tg, omodel.coco_anchors_feature(), parts=True)
    np.testing.assert_allclose(loss.detach().numpy(), gold_demo[f'g3_{tag}_loss'], **RT)
    np.testing.assert_allclose(np.array([p.item() for p in parts]), gold_demo[f'g3_{tag}_parts'], rtol=1e-6)
    loss.backward()
    for l in range(3):
        np.testing.assert_allclose(heads[l].grad.numpy(), gold_demo[f'g3_{tag}_grad{l}'], rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------------------------------- G4 blocks
def _run_block(gold, tag, mod, extra=None):
    mod.train()
    sd = {k[len(f'g4_{tag}_sd_'):]: T(gold[k]) for k in gold.files if k.startswith(f'g4_{tag}_sd_')}
    x = T(gold[f'g4_{tag}_x']).clone().requires_grad_(True)
    y = mod(x) if extra is None else extra(mod, x)
    np.testing.assert_allclose(y.detach().numpy(), gold[f'g4_{tag}_y'], rtol=1e-5, atol=1e-6)
    (y * T(gold[f'g4_{tag}_gy'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), gold[f'g4_{tag}_dx'], rtol=1e-4, atol=1e-6)
    for k, v in mod.named_parameters():
        np.testing.assert_allclose(v.grad.numpy(), gold[f'g4_{tag}_gr_{k}'], rtol=1e-4, atol=1e-5)
    for k, v in mod.state_dict().items():          # running stats / num_batches_tracked after the forward
        np.testing.assert_allclose(v.numpy(), sd[k].numpy(), rtol=1e-5, atol=1e-6)


def test_g4_blocks_seeded_init_and_values(gold_lib):
    torch.manual_seed(41); m = omodel.ConvUnit(32, 64, 3)
    assert np.array_equal(m.conv.weight.detach().numpy(), gold_lib['g4_cb3_w0'])     # same RNG consumption
    _run_block(gold_lib, 'cb3', m)
    torch.manual_seed(42); m = omodel.ConvUnit(32, 64, 3, stride=2)
    assert np.array_equal(m.conv.weight.detach().numpy(), gold_lib['g4_cb3s2_w0'])
    _run_block(gold_lib, 'cb3s2', m)
    torch.manual_seed(43); m = omodel.ConvUnit(64, 32, 1)
    assert np.array_equal(m.conv.weight.detach().numpy(), gold_lib['g4_cb1_w0'])
    _run_block(gold_lib, 'cb1', m)
    torch.manual_seed(44); m = omodel.Residual(64)
    assert np.array_equal(m.conv1.conv.weight.detach().numpy(), gold_lib['g4_res_w1'])
    assert np.array_equal(m.conv2.conv.weight.detach().numpy(), gold_lib['g4_res_w2'])
    _run_block(gold_lib, 'res', m)
    torch.manual_seed(45); sq = omodel.ConvUnit(64, 32, 1)
    assert np.array_equal(sq.conv.weight.detach().numpy(), gold_lib['g4_up_w0'])
    up = omodel._Named([('squeeze', sq)])
    skip = T(gold_lib['g4_up_skip'])
    _run_block(gold_lib, 'up', up, extra=lambda mod, x: torch.cat(
        [torch.nn.functional.interpolate(mod(x), scale_factor=2, mode='nearest'), skip], dim=1))


# ------------------------------------------------------------------------------------------- G5 whole model
def test_g5_library_model(gold_lib):
    net, crit = otrain.make_library(20220504)
    sd = net.state_dict()
    assert list(sd.keys()) == list(gold_lib['g5_keys'])
    assert len(sd) == 438 and sum(p.numel() for p in net.parameters()) == 61949149
    init = np.stack([stats(v.float()) for v in sd.values()])
    assert np.array_equal(init, gold_lib['g5_init'])            # bit-identical seeded init, all 438 tensors
    images, tg = synthetic_batch(2, 64)
    assert np.array_equal(tg.numpy(), gold_lib['g5_targets'])
    pred = net(images)
    for l, h in enumerate(pred):
        np.testing.assert_allclose(h.detach().numpy(), gold_lib[f'g5_head{l}'], rtol=1e-5, atol=1e-6)
    loss = crit(pred, tg)
    np.testing.assert_allclose(loss.detach().numpy(), gold_lib['g5_loss'], rtol=1e-6)
    loss.backward()
    assert [k for k, _ in net.named_parameters()] == list(gold_lib['g5_gradkeys'])
    grads = np.stack([stats(p.grad) for _, p in net.named_parameters()])
    assert_stats_close(grads, gold_lib['g5_grads'])
    after = np.stack([stats(v.float()) for v in net.state_dict().values()])
    assert_stats_close(after, gold_lib['g5_after'], rel=1e-5)
    net.eval()
    with torch.no_grad():
        _, dec = net(images, val=True)
    np.testing.assert_allclose(dec.numpy(), gold_lib['g5_decode'], rtol=1e-5, atol=1e-5)


def test_g5_demo_model(gold_demo):
    net, crit = otrain.make_demo(20220504)
    sd = net.state_dict()
    assert list(sd.keys()) == list(gold_demo['g5_keys'])
    init = np.stack([stats(v.float()) for v in sd.values()])
    assert np.array_equal(init, gold_demo['g5_init'])
    images, tg = synthetic_batch(2, 64)
    pred = net(images)
    for l, h in enumerate(pred):
        np.testing.assert_allclose(h.detach().numpy(), gold_demo[f'g5_head{l}'], rtol=1e-5, atol=1e-6)
    loss = crit(pred, tg)
    np.testing.assert_allclose(loss.detach().numpy(), gold_demo['g5_loss'], rtol=1e-6)
    loss.backward()
    assert [k for k, _ in net.named_parameters()] == list(gold_demo['g5_gradkeys'])
    grads = np.stack([stats(p.grad) for _, p in net.named_parameters()])
    assert_stats_close(grads, gold_demo['g5_grads'])


# ------------------------------------------------------------------------------------------- G6 curve (first steps)
def test_g6_loss_curve_prefix(gold_lib, gold_demo):
    """The full 100-step curves are the GPU path's gate; here the oracle reproduces their first 3 steps."""
    images, tg = synthetic_batch(2, 128)
    for gold, make in ((gold_lib, otrain.make_library), (gold_demo, otrain.make_demo)):
        assert gold['g6_curve'].shape == (100,)
        net, crit = make(20220504)
        got, _ = otrain.train_steps(net, crit, otrain.make_adam(net), images, tg, 3)
        np.testing.assert_allclose(np.array(got), gold['g6_curve'][:3], rtol=2e-5)
