"""GPU parity tests, path level: matcher / losses / whole model / 100-step curve against the golden vectors
captured from the reference (tests/golden) and against the CPU oracle on the same seeded inputs.

Bars (north_star): anchor/target indexing bit-exact; fp32 loss and gradients within 1e-3 relative; the
100-step loss curve within 1e-3 of the CPU reference.  bf16 results are reported with their own (looser)
tolerance, stated in each test.
"""
import os

import numpy as np
import pytest
import torch

from fastvision_amd.synthetic import coco_anchors_feature, coco_anchors_px, synthetic_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.from_numpy(np.asarray(a))


def stats(t):
    t = t.detach().double().flatten().cpu()
    return np.array([t.sum().item(), t.abs().sum().item()] + t[:4].tolist() + [0.0] * max(0, 4 - t.numel()))


def assert_stats_close(got, want, rel, what=''):
    scale = np.maximum(want[:, 1:2], 1e-30)
    bad = np.abs(got[:, 0:1] - want[:, 0:1]) / scale
    assert bad.max() < rel, f'{what} sum mismatch {bad.max()} at row {bad.argmax()}'
    r = np.abs(got[:, 1] - want[:, 1]) / np.maximum(want[:, 1], 1e-12)
    assert r.max() < rel, f'{what} abs-sum mismatch {r.max()} at row {r.argmax()}'


def lib_model(seed=20220504, training=True):
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    torch.manual_seed(seed)
    m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
               in_channels=3, num_classes=80, training=training)
    return m.to(DEV).train(training)


class _Shell:
    def __init__(self):
        self.anchors_per_level = [a for a in coco_anchors_px().view(3, 3, 1, 1, 2)]
        self.backbone_strides_per_level = [32, 16, 8]


def lib_loss():
    from fastvision_amd.loss import Yolov3Loss
    return Yolov3Loss(_Shell(), 0.5, 0.05, 1.0, 0.5)


# ------------------------------------------------------------------------------------------------ G1 matcher, bit-exact
def _check_match(gold, prefix, grids_hw, batch):
    crit = lib_loss()
    tg = T(gold[f'{prefix}_targets']).to(DEV)
    shapes = [torch.empty((batch, 3, h, w, 0), device=DEV) for h, w in grids_hw]
    locs, cats, xywh, anc = crit.build_target(shapes, tg)
    for l in range(3):
        assert np.array_equal(locs[l][0].cpu().numpy(), gold[f'{prefix}_l{l}_b'])
        assert np.array_equal(locs[l][1].cpu().numpy(), gold[f'{prefix}_l{l}_gxy'])
        assert np.array_equal(locs[l][2].cpu().numpy(), gold[f'{prefix}_l{l}_a'])
        assert np.array_equal(cats[l].cpu().numpy(), gold[f'{prefix}_l{l}_cls'])
        assert np.array_equal(xywh[l].cpu().numpy(), gold[f'{prefix}_l{l}_xywh'])     # fp32 outputs are bit-exact too
        assert np.array_equal(anc[l].cpu().numpy(), gold[f'{prefix}_l{l}_anc'])


def test_matcher_bit_exact_vs_reference(gold_lib):
    for c in range(int(gold_lib['g1_cases'])):
        grids = gold_lib[f'g1_{c}_grids']
        _check_match(gold_lib, f'g1_{c}', [(int(s), int(s)) for s in grids], 8)
    _check_match(gold_lib, 'g1ns', [(20, 15), (40, 30), (80, 60)], 2)


def test_matcher_full_size_properties():
    """BASELINE config 3 sizes (B=32, 640 px): counts and index ranges agree with the oracle run on the same batch."""
    from oracle import losses as ol, model as om
    _, tg = synthetic_batch(32, 640)
    crit = lib_loss()
    shapes = [torch.empty((32, 3, g, g, 0), device=DEV) for g in (20, 40, 80)]
    locs, cats, xywh, anc = crit.build_target(shapes, tg.to(DEV))
    ref = ol.build_target([(32, 3, g, g, 85) for g in (20, 40, 80)], tg, [a for a in om.coco_anchors_px().view(3, 3, 1, 1, 2)],
                          om.LEVEL_STRIDES)
    for l, g in enumerate((20, 40, 80)):
        assert torch.equal(locs[l][0].cpu(), ref[0][l][0]) and torch.equal(locs[l][1].cpu(), ref[0][l][1])
        assert torch.equal(locs[l][2].cpu(), ref[0][l][2]) and torch.equal(cats[l].cpu(), ref[1][l])
        assert torch.equal(xywh[l].cpu(), ref[2][l]) and torch.equal(anc[l].cpu(), ref[3][l])
        assert int(locs[l][1].max()) < g and int(locs[l][1].min()) >= 0


# ------------------------------------------------------------------------------------------------ G2 IoU family
def test_iou_family_vs_reference(gold_lib, gold_demo):
    from fastvision_amd.demos.yolov3_u.utils import iou as DI
    from fastvision_amd.detection import tools as TT
    a, b = T(gold_lib['g2_a']).to(DEV), T(gold_lib['g2_b']).to(DEV)
    wa, wb = a[:, 2:] - a[:, :2], b[:, 2:] - b[:, :2]
    xa, xb = TT.xyxy2xywh(a), TT.xyxy2xywh(b)
    chk = lambda got, key: np.testing.assert_allclose(got.cpu().numpy(), gold_lib[key], rtol=2e-5, atol=2e-6)
    chk(TT.xyxy_iou(a, b), 'g2_xyxy_iou')
    chk(TT.xywh_iou(xa, xb), 'g2_xywh_iou')
    chk(TT.wh_iou(wa, wb), 'g2_wh_iou')
    chk(TT.xyxy_iou_batch(a[:40], b[:24]), 'g2_xyxy_iou_batch')
    chk(TT.xywh_iou_batch(xa[:40], xb[:24]), 'g2_xywh_iou_batch')
    chk(TT.wh_iou_batch(wa[:40], wb[:24]), 'g2_wh_iou_batch')
    chk(TT.GIOU(a, b), 'g2_giou')
    chk(TT.DIOU(a, b), 'g2_diou')
    chk(TT.CIOU(a, b), 'g2_ciou')
    chk(TT.CIOU(xa, xb, mode='xywh'), 'g2_ciou_xywh')
    a2, b2 = T(gold_demo['g2_a']).to(DEV), T(gold_demo['g2_b']).to(DEV)
    np.testing.assert_allclose(DI.DIOU(a2, b2).cpu().numpy(), gold_demo['g2_diou'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(DI.CIOU(a2, b2).cpu().numpy(), gold_demo['g2_ciou'], rtol=2e-5, atol=2e-6)
    # CIOULoss value and gradient w.r.t. the predicted boxes
    from fastvision_amd.loss import CIOULoss
    ar = a.clone().requires_grad_(True)
    l = CIOULoss('mean')(ar, b)
    np.testing.assert_allclose(l.detach().cpu().numpy().reshape(1), gold_lib['g2_cioul'], rtol=1e-5)
    l.backward()
    np.testing.assert_allclose(ar.grad.cpu().numpy(), gold_lib['g2_cioul_grad'], rtol=1e-3, atol=1e-6)
    with pytest.raises(Exception):
        TT.cal_iou(a, b, mode='nope')
    with pytest.raises(RuntimeError):
        TT.xyxy_iou(a.cpu(), b.cpu())
    # numpy callers (the reference's numpy branches, detection/tools/IOU.py:60-66,130-146): same kernels, numpy out in the input's dtype
    an, bn = gold_lib['g2_a'].astype(np.float64), gold_lib['g2_b'].astype(np.float64)
    got = TT.xyxy_iou(an, bn)
    assert isinstance(got, np.ndarray) and got.dtype == np.float64 and got.shape == gold_lib['g2_xyxy_iou'].shape
    np.testing.assert_allclose(got, gold_lib['g2_xyxy_iou'], rtol=2e-5, atol=2e-6)
    gb = TT.xyxy_iou_batch(an[:40].astype(np.float32), bn[:24].astype(np.float32))
    assert isinstance(gb, np.ndarray) and gb.dtype == np.float32
    np.testing.assert_allclose(gb, gold_lib['g2_xyxy_iou_batch'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(TT.wh_iou((an[:, 2:] - an[:, :2]), (bn[:, 2:] - bn[:, :2])), gold_lib['g2_wh_iou'], rtol=2e-5, atol=2e-6)
    assert TT.GIOU(an, bn).shape == gold_lib['g2_giou'].shape


# ------------------------------------------------------------------------------------------------ G3 losses + head grads
@pytest.mark.parametrize('tag', ['rand', 'empty', 'dup', 'syn'])
def test_library_loss_and_head_grads_vs_reference(gold_lib, tag):
    crit = lib_loss()
    tg = T(gold_lib[f'g3_{tag}_targets']).to(DEV)
    heads = [T(gold_lib[f'g3_{tag}_head{l}']).to(DEV).requires_grad_(True) for l in range(3)]
    loss = crit(heads, tg)
    assert tuple(loss.shape) == (1,)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), gold_lib[f'g3_{tag}_loss'], rtol=1e-4)
    loss.backward()
    for l in range(3):
        want = gold_lib[f'g3_{tag}_grad{l}']
        got = heads[l].grad.cpu().numpy()
        err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-12)
        assert err < 1e-3, f'{tag} level {l}: {err}'


@pytest.mark.parametrize('tag', ['syn', 'syn4', 'dup'])
def test_demo_loss_and_head_grads_vs_reference(gold_demo, tag):
    from fastvision_amd.demos.yolov3_u.utils import ComputeLoss

    class M:
        anchors = coco_anchors_feature()
    crit = ComputeLoss()
    tg = T(gold_demo[f'g3_{tag}_targets']).to(DEV)
    heads = [T(gold_demo[f'g3_{tag}_head{l}']).to(DEV).requires_grad_(True) for l in range(3)]
    loss = crit(heads, tg, M())
    np.testing.assert_allclose(loss.detach().cpu().numpy(), gold_demo[f'g3_{tag}_loss'], rtol=1e-4)
    np.testing.assert_allclose(crit.last_parts.detach().cpu().numpy(), gold_demo[f'g3_{tag}_parts'], rtol=1e-4)
    loss.backward()
    for l in range(3):
        want = gold_demo[f'g3_{tag}_grad{l}']
        err = np.abs(heads[l].grad.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-12)
        assert err < 1e-3, f'{tag} level {l}: {err}'


# ------------------------------------------------------------------------------------------------ G5 whole model, fp32
def test_library_model_fp32_vs_reference(gold_lib):
    import fastvision_amd
    with fastvision_amd.compute_dtype(torch.float32):
        net = lib_model()
        crit = lib_loss()
        images, tg = synthetic_batch(2, 64)
        pred = net(images.to(DEV))
        for l, h in enumerate(pred):
            want = gold_lib[f'g5_head{l}']
            assert tuple(h.shape) == want.shape
            err = np.abs(h.detach().cpu().numpy() - want).max() / np.abs(want).max()
            assert err < 1e-3, f'head {l}: {err}'
        loss = crit(pred, tg.to(DEV))
        np.testing.assert_allclose(loss.detach().cpu().numpy(), gold_lib['g5_loss'], rtol=1e-3)
        loss.backward()
    assert [k for k, _ in net.named_parameters()] == list(gold_lib['g5_gradkeys'])
    grads = np.stack([stats(p.grad) for _, p in net.named_parameters()])
    assert_stats_close(grads, gold_lib['g5_grads'], 2e-3, 'grads')
    after = np.stack([stats(v.float()) for v in net.state_dict().values()])
    assert_stats_close(after, gold_lib['g5_after'], 1e-3, 'state after step')


def test_library_eval_decode_vs_reference(gold_lib):
    import fastvision_amd
    with fastvision_amd.compute_dtype(torch.float32):
        net = lib_model()
        images, _ = synthetic_batch(2, 64)
        net(images.to(DEV))                          # one train-mode forward updates the running stats as in the fixture
        net.eval()
        with torch.no_grad():
            heads, dec = net(images.to(DEV), val=True)
    want = gold_lib['g5_decode']
    assert tuple(dec.shape) == want.shape
    err = np.abs(dec.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 2e-3, err


def test_demo_model_fp32_vs_reference(gold_demo):
    import fastvision_amd
    from fastvision_amd.demos.yolov3_u.models import YoloV3
    from fastvision_amd.demos.yolov3_u.utils import ComputeLoss
    with fastvision_amd.compute_dtype(torch.float32):
        torch.manual_seed(20220504)
        net = YoloV3(anchors=coco_anchors_feature()).to(DEV).train()
        crit = ComputeLoss()
        images, tg = synthetic_batch(2, 64)
        pred = net(images.to(DEV))
        for l, h in enumerate(pred):
            want = gold_demo[f'g5_head{l}']
            assert tuple(h.shape) == want.shape
            err = np.abs(h.detach().cpu().numpy() - want).max() / np.abs(want).max()
            assert err < 1e-3, f'head {l}: {err}'
        loss = crit(pred, tg.to(DEV), net)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), gold_demo['g5_loss'], rtol=1e-3)
        loss.backward()
    grads = np.stack([stats(p.grad) for _, p in net.named_parameters()])
    assert_stats_close(grads, gold_demo['g5_grads'], 2e-3, 'grads')


# ------------------------------------------------------------------------------------------------ G6 100-step loss curve
@pytest.mark.parametrize('surface', ['lib', 'demo'])
def test_loss_curve_100_steps_fp32_vs_reference(gold_lib, gold_demo, surface):
    import fastvision_amd
    from fastvision_amd import FusedAdam
    images, tg = synthetic_batch(2, 128)
    images, tg = images.to(DEV), tg.to(DEV)
    with fastvision_amd.compute_dtype(torch.float32):
        if surface == 'lib':
            net, crit, gold = lib_model(), lib_loss(), gold_lib['g6_curve']
            step_loss = lambda pred: crit(pred, tg)
        else:
            from fastvision_amd.demos.yolov3_u.models import YoloV3
            from fastvision_amd.demos.yolov3_u.utils import ComputeLoss
            torch.manual_seed(20220504)
            net = YoloV3(anchors=coco_anchors_feature()).to(DEV).train()
            cl, gold = ComputeLoss(), gold_demo['g6_curve']
            step_loss = lambda pred: cl(pred, tg, net)
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        curve = []
        for _ in range(100):
            pred = net(images)
            opt.zero_grad()
            loss = step_loss(pred)
            loss.backward()
            opt.step()
            curve.append(loss.detach())
    curve = torch.cat(curve).cpu().numpy()
    rel = np.abs(curve - gold) / np.abs(gold)
    print(f'{surface} curve: first {curve[:3]} last {curve[-3:]} max rel dev {rel.max():.2e} at step {rel.argmax()}'
          f' first-10 max {rel[:10].max():.2e}')
    if surface == 'demo':
        assert rel.max() < 1e-3, f'demo: max rel deviation {rel.max()} at step {rel.argmax()}'
    else:
        # The library loss trajectory is chaotic at the 1e-2 level IN THE REFERENCE ITSELF: its objectness target is the
        # non-detached IoU of the prediction (loss/yolov3_loss.py:60-61).  oracle/make_golden.py lib_curves re-ran the reference's
        # 100 steps with 1, 3 and 8 intra-op threads (only the order of the fp32 reductions changes): g6_curve_t{1,3,8}.  The GPU
        # curve must hold 1e-3 on the first 10 steps and afterwards stay inside 1.5 x the reference's own spread (its largest
        # deviation from itself over the 100 steps: where along the curve two runs part company is itself chaotic).
        spread = np.max([np.abs(gold_lib[f'g6_curve_t{t}'] - gold) / np.abs(gold) for t in (1, 3, 8)], axis=0)
        print(f'reference spread over thread counts: max {spread.max():.2e} at step {spread.argmax()} (first 10 steps {spread[:10].max():.2e}); '
              f'GPU max deviation / reference spread = {rel.max() / spread.max():.2f}')
        assert rel[:10].max() < 1e-3, f'lib: first-10 rel deviation {rel[:10].max()}'
        assert rel.max() <= 1.5 * spread.max(), f'lib: max rel deviation {rel.max()} at step {rel.argmax()}, reference spread {spread.max()}'


def test_loss_curve_100_steps_bf16_reported(gold_lib, gold_demo):
    """The bench dtype (bf16 storage, fp32 accumulate / statistics / loss / master weights) on the same 100 steps, both surfaces:
    observed deviation from the reference's fp32 curve is printed (first run: library 5.4e-3 over the first 10 steps, 1.8e-2 at
    most, 4.4e-3 on average; demo 3.6e-2 at step 4 -- its loss jumps 18 -> 12 -> 13.4 over the first steps -- and 2.1e-3 on
    average).  Bars: library 2e-2 over the first 10 steps and 2e-2 + 3 x the reference's own spread overall; demo 6e-2 overall
    and 1e-2 on average."""
    import fastvision_amd
    from fastvision_amd import FusedAdam
    images, tg = synthetic_batch(2, 128)
    images, tg = images.to(DEV), tg.to(DEV)
    for surface in ('lib', 'demo'):
        with fastvision_amd.compute_dtype(torch.bfloat16):
            if surface == 'lib':
                net, crit, gold = lib_model(), lib_loss(), gold_lib['g6_curve']
                step_loss = lambda pred: crit(pred, tg)
            else:
                from fastvision_amd.demos.yolov3_u.models import YoloV3
                from fastvision_amd.demos.yolov3_u.utils import ComputeLoss
                torch.manual_seed(20220504)
                net = YoloV3(anchors=coco_anchors_feature()).to(DEV).train()
                cl, gold = ComputeLoss(), gold_demo['g6_curve']
                step_loss = lambda pred: _silently(cl, pred, tg, net)
            opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
            curve = []
            for _ in range(100):
                pred = net(images)
                opt.zero_grad()
                loss = step_loss(pred)
                loss.backward()
                opt.step()
                curve.append(loss.detach().reshape(1))
        curve = torch.cat(curve).cpu().numpy()
        rel = np.abs(curve - gold) / np.abs(gold)
        print(f'bf16 {surface} curve: first {curve[:3]} last {curve[-3:]} (reference {gold[:3]} .. {gold[-3:]}); max rel dev {rel.max():.2e} at '
              f'step {rel.argmax()}, first-10 max {rel[:10].max():.2e}, mean {rel.mean():.2e}')
        assert np.all(np.isfinite(curve)) and curve[-1] < curve[0]
        if surface == 'lib':
            spread = np.max([np.abs(gold_lib[f'g6_curve_t{t}'] - gold) / np.abs(gold) for t in (1, 3, 8)], axis=0)
            # A bf16 trajectory is ONE realisation of the dtype's rounding noise: a last-bit change of the BatchNorm statistics' arithmetic
            # moves it as far as anything else does.  Observed over three builds of that arithmetic (rounds 3-4): first-10 max 5.4e-3 /
            # 6.1e-3 / 9.1e-3, overall max 1.8e-2 / 1.75e-2 / 2.65e-2 (the reference's own fp32 thread-count spread is 1.66e-2:
            # g6_curve_t{1,3,8}).  Bars = the docstring's: 2e-2 over the first ten steps, 2e-2 + 3 x that spread overall.
            assert spread.max() < 2e-2 and rel[:10].max() < 2e-2 and rel.max() < 2e-2 + 3 * spread.max()
        else:
            # observed over the same three builds: max 3.6e-2 / 3.2e-2 / 3.0e-2 (steps 2-4, where the loss jumps 18 -> 12 -> 13.4), mean 2.1e-3 / 3.9e-3 / 5.5e-3
            assert rel.max() < 6e-2 and rel.mean() < 1e-2


# ------------------------------------------------------------------------------------------------ bf16 path (the bench dtype)
def test_library_model_bf16_close_to_fp32_oracle():
    """bf16 storage / fp32 accumulate vs the fp32 CPU oracle on the same batch (B=2, 128 px; the 64-px fixture has
    only 8 pixels per channel at the deepest level, where BatchNorm amplifies any rounding).  Observed deviation is
    printed (first run: heads 9e-2 / 9e-2 / 6e-2 of the fp32 scale at random init, loss 9e-5, gradient norms median 4e-3,
    max 6e-2); bars: heads within 1.5e-1, loss within 2e-2, median per-tensor gradient norm within 5e-2."""
    import fastvision_amd
    from oracle import train as otrain
    images, tg = synthetic_batch(2, 128)
    ref, ref_crit = otrain.make_library(20220504)
    ref_pred = ref(images)
    ref_loss = ref_crit(ref_pred, tg)
    ref_loss.backward()
    with fastvision_amd.compute_dtype(torch.bfloat16):
        net = lib_model()
        crit = lib_loss()
        pred = net(images.to(DEV))
        errs = [((h.detach().cpu() - r.detach()).abs().max() / r.detach().abs().max()).item() for h, r in zip(pred, ref_pred)]
        loss = crit(pred, tg.to(DEV))
        loss.backward()
    lrel = abs(loss.item() - ref_loss.item()) / ref_loss.item()
    gn = np.array([[p.grad.float().norm().item(), r.grad.norm().item()] for (_, p), (_, r) in
                   zip(net.named_parameters(), ref.named_parameters())])
    r = np.abs(gn[:, 0] - gn[:, 1]) / np.maximum(gn[:, 1], 1e-12)
    print('bf16 head errs', errs, 'loss rel', lrel, 'grad-norm rel dev: median', np.median(r), 'max', r.max())
    assert max(errs) < 1.5e-1 and lrel < 2e-2
    assert np.median(r) < 5e-2


def test_full_size_step_runs_and_is_finite():
    """One bf16 train step at reduced batch of the bench shape (B=4, 640 px): finite loss, every parameter gets a
    finite gradient, BN buffers move -- size-independent sanity at the real spatial sizes (grids 20/40/80)."""
    import fastvision_amd
    from fastvision_amd import FusedAdam
    net, crit = lib_model(), lib_loss()
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
    images, tg = synthetic_batch(4, 640)
    pred = net(images.to(DEV))
    assert [tuple(p.shape) for p in pred] == [(4, 3, 20, 20, 85), (4, 3, 40, 40, 85), (4, 3, 80, 80, 85)]
    opt.zero_grad()
    loss = crit(pred, tg.to(DEV))
    loss.backward()
    opt.step()
    assert torch.isfinite(loss).all()
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert int(net.backbone.conv0.bn.num_batches_tracked) == 1


# ------------------------------------------------------------------------------------------------ data parallel on the GPU
def _dp_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      FVA_DIST_BACKEND='gloo')
    import torch.distributed as dist
    import fastvision_amd
    from fastvision_amd import FusedAdam, parallel
    parallel.init_from_env()
    with fastvision_amd.compute_dtype(torch.float32):
        net, crit = lib_model(seed=20220504 + rank), lib_loss()          # different init per rank: broadcast must fix it
        parallel.broadcast_parameters(net)
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        red = parallel.GradientReducer(net.parameters(), bucket_bytes=16 << 20)
        images, tg = synthetic_batch(4, 64)                              # global batch of 4, two images per rank
        mine = images[rank * 2:(rank + 1) * 2].to(DEV)
        mytg = parallel.shard_targets(tg, rank, 2).to(DEV)
        pred = net(mine)
        opt.zero_grad()
        loss = crit(pred, mytg)
        loss.backward()
        launched = red.next_launch
        red.finish()
        g = {k: p.grad.detach().cpu().clone() for k, p in list(net.named_parameters())[::29]}
        opt.step()
        w = {k: p.detach().cpu().clone() for k, p in list(net.named_parameters())[::29]}
    out[rank] = (g, w, launched, len(red.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_one_gpu():
    """Two ranks (sharing the one GPU of the test box, gloo transport) run the real DP step: parameters broadcast,
    bucketed all-reduce launched during backward, FusedAdam on the bucket views; both ranks must end bit-identical
    and the averaged gradient must equal the mean of the two single-rank gradients computed separately."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.start_processes(_dp_worker, args=(2, port, out), nprocs=2, join=True, start_method='spawn')
    (g0, w0, l0, nb), (g1, w1, l1, _) = out[0], out[1]
    assert l0 == l1 and nb >= 8 and l0 >= nb - 2                        # collectives went out during backward
    for k in g0:
        assert torch.equal(g0[k], g1[k]) and torch.equal(w0[k], w1[k]), k
    # reference: same two shards on one process, gradients averaged by hand
    import fastvision_amd
    with fastvision_amd.compute_dtype(torch.float32):
        images, tg = synthetic_batch(4, 64)
        acc = None
        for rank in range(2):
            net, crit = lib_model(seed=20220504), lib_loss()
            from fastvision_amd import parallel
            loss = crit(net(images[rank * 2:(rank + 1) * 2].to(DEV)), parallel.shard_targets(tg, rank, 2).to(DEV))
            loss.backward()
            gr = {k: p.grad.detach().cpu() / 2 for k, p in list(net.named_parameters())[::29]}
            acc = gr if acc is None else {k: acc[k] + gr[k] for k in gr}
    for k in g0:
        err = ((g0[k] - acc[k]).abs().max() / acc[k].abs().max().clamp_min(1e-12)).item()
        assert err < 1e-4, f'{k}: {err}'


def test_side_stream_gradient_consumers():
    """Weight gradients run on the library's side stream (ops.wgrad_stream).  Three ways of consuming them must see finished
    values: .grad after backward(), accumulation into an existing .grad (autograd adds on the main stream: those layers must
    stay there), and torch.autograd.grad()."""
    import fastvision_amd
    from fastvision_amd import ops
    assert ops._SIDE['on'], 'the side stream is the product default'
    images, tg = synthetic_batch(2, 128)
    images, tg = images.to(DEV), tg.to(DEV)
    with fastvision_amd.compute_dtype(torch.float32):
        net, crit = lib_model(), lib_loss()
        params = [p for p in net.parameters() if p.requires_grad]
        crit(net(images), tg).backward()
        once = [p.grad.clone() for p in params]
        bn0 = {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}
        net.load_state_dict(bn0, strict=False)
        crit(net(images), tg).backward()                      # second pass accumulates into the existing gradients
        for p, g in zip(params, once):
            assert torch.equal(p.grad, g + g)
        prev = ops.set_wgrad_side_stream(False)               # reference: everything on one stream
        try:
            for p in params:
                p.grad = None
            crit(net(images), tg).backward()
            serial = [p.grad.clone() for p in params]
        finally:
            ops.set_wgrad_side_stream(prev)
        got = torch.autograd.grad(crit(net(images), tg), params)
        for a, b, c in zip(once, serial, got):
            assert torch.equal(a, b) and torch.equal(a, c)
        # a fresh side stream (what ops.autotune_wgrad_side_stream asks for when the first one loses) serves the same gradients
        from fastvision_amd import _lib
        h0 = ops.fork_side_stream().cuda_stream
        ops.join_side_stream(force=True)
        torch.cuda.synchronize()
        _lib.call('fva_side_stream_renew')
        h1 = ops.fork_side_stream().cuda_stream
        ops.join_side_stream(force=True)
        assert h0 != h1, 'fva_side_stream_renew must hand out a new stream'
        for p in params:
            p.grad = None
        crit(net(images), tg).backward()
        for p, g in zip(params, once):
            assert torch.equal(p.grad, g)


def _rccl_worker(rank, world, port, out):
    import os
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    import fastvision_amd
    from fastvision_amd import FusedAdam, parallel
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    images, tg = synthetic_batch(4, 128)
    images, tg = images.to(DEV), tg.to(DEV)
    res = {}
    for mode in ('plain', 'reduced', 'wire16'):
        net, crit = lib_model(), lib_loss()                                  # bf16 compute: the bench's kernels
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        red = None
        if mode != 'plain':
            # world=2 on a one-rank group (which averages to the identity): forces the collectives to be ISSUED
            red = parallel.GradientReducer(net.parameters(), bucket_bytes=16 << 20, world=2,
                                           bucket_dtype=torch.bfloat16 if mode == 'wire16' else None)
        for it in range(1 if mode == 'wire16' else 2):
            opt.zero_grad()
            loss = crit(net(images), tg)
            loss.backward()
            if red is not None:
                launched = red.next_launch
                red.finish()
            if mode == 'plain' and it == 0:
                torch.cuda.synchronize()
                res['first_grad'] = {k: p.grad.detach().cpu().clone() for k, p in list(net.named_parameters())[::17]}
            opt.step()
        torch.cuda.synchronize()
        res[mode] = {k: p.detach().cpu().clone() for k, p in list(net.named_parameters())[::17]}
        res[mode + '_grad'] = {k: p.grad.detach().cpu().clone() for k, p in list(net.named_parameters())[::17]}
    out['res'] = (res, launched, len(red.buckets))
    dist.destroy_process_group()


def test_reducer_over_rccl_with_side_stream_single_rank():
    """The N > 1 mechanics on the real backend (RCCL), as far as one GPU allows: a one-rank nccl group with the reducer
    forced to issue its bucketed AVG all-reduces.  Weight gradients are computed on the library's low-priority side
    stream; the bucket copies and collectives are enqueued behind them.  Two optimizer steps must give bit-identical
    parameters and gradients to the same two steps without the reducer."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.start_processes(_rccl_worker, args=(1, port, out), nprocs=1, join=True, start_method='spawn')
    res, launched, nb = out['res']
    assert nb >= 8 and launched >= nb - 2                                   # collectives went out during backward
    for k in res['plain']:
        assert torch.equal(res['plain'][k], res['reduced'][k]), k
        assert torch.equal(res['plain_grad'][k], res['reduced_grad'][k]), k
    # bf16 on the wire: after ONE step the gradients are the plain run's first-step gradients rounded to bf16 (fva_gather_cast)
    for k in res['first_grad']:
        assert torch.equal(res['first_grad'][k].bfloat16().float(), res['wire16_grad'][k]), k


def test_config2_fp32_full_size_vs_oracle():
    """BASELINE config 2 at full size: YOLOv3 8x3x416x416 fp32 (grids 13/26/52), one train step vs the CPU oracle on
    the same seeded batch: heads and loss within 1e-3, per-tensor gradient norms within 2e-3, matcher indices exact."""
    import fastvision_amd
    from oracle import losses as ol, train as otrain
    images, tg = synthetic_batch(8, 416)
    torch.set_num_threads(max(1, min(16, len(__import__('os').sched_getaffinity(0)))))
    ref, ref_crit = otrain.make_library(20220504)
    ref_pred = ref(images)
    ref_loss = ref_crit(ref_pred, tg)
    ref_loss.backward()
    with fastvision_amd.compute_dtype(torch.float32):
        net, crit = lib_model(), lib_loss()
        pred = net(images.to(DEV))
        assert [tuple(p.shape) for p in pred] == [(8, 3, 13, 13, 85), (8, 3, 26, 26, 85), (8, 3, 52, 52, 85)]
        loss = crit(pred, tg.to(DEV))
        loss.backward()
        locs, cats, _, _ = crit.build_target(pred, tg.to(DEV))
    for h, r in zip(pred, ref_pred):
        err = ((h.detach().cpu() - r.detach()).abs().max() / r.detach().abs().max()).item()
        assert err < 1e-3, f'head {err}'
    assert abs(loss.item() - ref_loss.item()) / ref_loss.item() < 1e-3
    rlocs, rcats, _, _ = ol.build_target([p.shape for p in ref_pred], tg, ref.anchors_per_level, ref.backbone_strides_per_level)
    for l in range(3):
        assert torch.equal(locs[l][0].cpu(), rlocs[l][0]) and torch.equal(locs[l][1].cpu(), rlocs[l][1])
        assert torch.equal(locs[l][2].cpu(), rlocs[l][2]) and torch.equal(cats[l].cpu(), rcats[l])
    gn = np.array([[p.grad.norm().item(), r.grad.norm().item()] for (_, p), (_, r) in zip(net.named_parameters(), ref.named_parameters())])
    rel = np.abs(gn[:, 0] - gn[:, 1]) / np.maximum(gn[:, 1], 1e-12)
    print('cfg2 grad-norm rel dev: median', np.median(rel), 'max', rel.max())
    assert rel.max() < 2e-3


def test_demo_fit_loop_trains_validates_and_steps_lr(tmp_path, monkeypatch):
    """demos/yolov3_u/cfg/_fit.py mirror: _Train / _Validate return the mean batch loss and print the reference's log line;
    Fit saves the best checkpoint and divides the LR by ten after three epochs without improvement."""
    import types
    import fastvision_amd
    from fastvision_amd.demos.yolov3_u.cfg import _fit as F
    from fastvision_amd.demos.yolov3_u.models import YoloV3
    from fastvision_amd.demos.yolov3_u.utils import ComputeLoss
    from fastvision_amd.synthetic import coco_anchors_feature, synthetic_batch
    monkeypatch.chdir(tmp_path)
    with fastvision_amd.compute_dtype(torch.float32):
        torch.manual_seed(1)
        net = YoloV3(anchors=tuple(a.to('cuda:0') for a in coco_anchors_feature())).to('cuda:0')
        crit = ComputeLoss()
        quiet = lambda pred, tg, model: _silently(crit, pred, tg, model)
        loader = [synthetic_batch(2, 64, seed=s) for s in (1, 2)]
        opt = fastvision_amd.FusedAdam(net.parameters(), lr=1e-3)
        lines = []
        tr = F._Train(net, loader, opt, quiet, 0, log=lines.append)
        va = F._Validate(net, loader, quiet, 0, log=lines.append)
        assert np.isfinite(tr) and np.isfinite(va) and len(lines) == 4
        assert lines[0].startswith('epoch : 1 batch : 1 / 2 loss : ')
        assert not net.training                                   # _Validate leaves the model in eval mode, as the reference
        # LR rule: a criterion whose validation loss never improves after the first epoch
        calls = {'n': 0}
        def fake_validate(model, loader_, criterion, epoch, log=print):
            calls['n'] += 1
            return 1.0 + calls['n']
        monkeypatch.setattr(F, '_Validate', fake_validate)
        monkeypatch.setattr(F, '_Train', lambda *a, **k: 0.5)
        args = types.SimpleNamespace(start_epoch=0, max_epochs=6)
        F.Fit(net, args, opt, quiet, None, loader, loader)
        assert abs(opt.param_groups[0]['lr'] - 1e-4) < 1e-12     # one decay, at the epoch where patience reached 3
        assert os.path.exists(tmp_path / 'epoch_1_loss_2.0.pth')


def _silently(crit, pred, tg, model):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        return crit(pred, tg, model)


def test_bf16_step_same_with_and_without_8phase_kernels(tmp_path):
    """whole model, B = 32 at 416 px (ragged grids 13/26/52), one bf16 step in child processes:
    * 8-phase wgrad on / off: forward identical (same loss, bit for bit), weight gradients equal to the fp32 rounding of the
      different split-K order;
    * 8-phase igemm on / off: its conv outputs are bit-identical (kernel-level test), but BatchNorm partial sums are grouped
      by 256 instead of 128 rows, which moves the batch statistics in their last bits and flips bf16 roundings downstream --
      the step agrees to bf16 noise (loss 1e-4, gradient norms 5e-3 in the median)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'step_dump.py')

    def run(tag, **flags):
        out = str(tmp_path / f'{tag}.npz')
        r = subprocess.run([sys.executable, tool, out, '32', '416'], env=dict(os.environ, **flags), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        return np.load(out)

    base = run('none', FVA_IGEMM8='0', FVA_WGRAD8='0')
    wg = run('wgrad8', FVA_IGEMM8='0', FVA_WGRAD8='1')
    both = run('both')
    assert wg['loss'][0] == base['loss'][0]
    for k in base.files:
        if k.startswith('samp/'):
            assert np.abs(wg[k] - base[k]).max() <= 1e-5 * (np.abs(base[k]).max() + 1e-12), k
    assert abs(both['loss'][0] - base['loss'][0]) <= 1e-4 * abs(base['loss'][0])
    rel = np.array([abs(both[k][0] - base[k][0]) / max(base[k][0], 1e-12) for k in base.files if k.startswith('norm/')])
    assert np.median(rel) < 5e-3 and rel.max() < 5e-2, (np.median(rel), rel.max())


# ------------------------------------------------------------------------------------------------ a-9 stand-alone BCE class
def test_bicrossentropy_class_vs_reference(gold_lib):
    """loss/classification_loss.py:36-65 on the device kernel fva_bce_loss, against the reference's own outputs
    (g2_bce_*: probabilities with integer labels / mean, logits / sum) and against torch autograd for the gradient, the dense-
    target form (last dimension 1, as Yolov3Loss's objectness term calls it) and weights."""
    from fastvision_amd.loss import BiCrossEntropyLoss
    p, lab = T(gold_lib['g2_bce_p']).to(DEV), T(gold_lib['g2_bce_lab']).to(DEV)
    got = BiCrossEntropyLoss('mean')(p, lab, already_sigmoid=True)
    np.testing.assert_allclose(got.detach().cpu().numpy().reshape(1), gold_lib['g2_bce_mean'], rtol=1e-5)
    logits = (p * 4 - 2).clone().requires_grad_(True)
    got = BiCrossEntropyLoss('sum')(logits, lab)
    np.testing.assert_allclose(got.detach().cpu().numpy().reshape(1), gold_lib['g2_bce_logits_sum'], rtol=1e-5)
    got.backward()
    ref = (p * 4 - 2).detach().cpu().clone().requires_grad_(True)
    tgt = torch.zeros(12, 5).scatter_(1, lab.cpu().view(-1, 1), 1.0).view(-1, 1)
    s = ref.view(-1, 1).sigmoid()
    (-tgt * torch.log(s + 1e-8) - (1 - tgt) * torch.log(1 - s + 1e-8)).sum().backward()
    np.testing.assert_allclose(logits.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-4, atol=1e-6)
    # dense target + per-element weights, mean
    g = torch.Generator().manual_seed(2)
    y = torch.randn(50, 1, generator=g)
    t = torch.rand(50, 1, generator=g)
    w = torch.rand(50, generator=g)
    yd = y.to(DEV).requires_grad_(True)
    got = BiCrossEntropyLoss('mean')(yd, t.to(DEV), weights=w.to(DEV))
    got.backward()
    yr = y.clone().requires_grad_(True)
    sr = yr.sigmoid()
    want = ((-t * torch.log(sr + 1e-8) - (1 - t) * torch.log(1 - sr + 1e-8)).sum(1) * w).sum() / 50
    want.backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-5)
    np.testing.assert_allclose(yd.grad.cpu().numpy(), yr.grad.numpy(), rtol=1e-4, atol=1e-7)
    with pytest.raises(RuntimeError):
        BiCrossEntropyLoss()(p.cpu(), lab.cpu())


# ------------------------------------------------------------------------------------------------ DataParallel semantics, N ranks
def _dp_sum_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      FVA_DIST_BACKEND='gloo')
    import torch.distributed as dist
    import fastvision_amd
    from fastvision_amd import parallel
    parallel.init_from_env()
    with fastvision_amd.compute_dtype(torch.float32):
        net, crit = lib_model(seed=20220504), lib_loss()
        crit.data_parallel()                                             # job-wide match counts and batch in the loss kernel
        red = parallel.GradientReducer(net.parameters(), bucket_bytes=16 << 20, average=False)
        images, tg = synthetic_batch(4, 64, seed=77)                     # global batch of 4, two images per rank
        pred = net(images[rank * 2:(rank + 1) * 2].to(DEV))
        loss = crit(pred, parallel.shard_targets(tg, rank, 2).to(DEV))
        loss.backward()
        red.finish()
        share = loss.detach().clone()
        dist.all_reduce(share)                                           # the shares add up to the gathered-batch loss
        g = {k: p.grad.detach().cpu().clone() for k, p in list(net.named_parameters())[::23]}
    out[rank] = (g, float(share), float(loss))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_reproduces_the_reference_dataparallel_step():
    """Two ranks (one GPU, gloo), each with its own two images: the loss kernel normalises its per-match means by the match counts
    of the WHOLE job and multiplies by the job's batch (fva_yolov3_loss_dp), gradients are SUMMED.  Reference semantics
    (demos/yolov3_u/train.py:85 nn.DataParallel + demos/yolov3_u/cfg/_fit.py:48-51 + loss/yolov3_loss.py:69-71): every replica runs
    forward on its shard (per-GPU BatchNorm statistics), the outputs are gathered and ONE loss is evaluated on the 4-image batch.
    The checker is the CPU ORACLE (round 3; it was an in-process HIP emulation): two shard forwards of oracle.model.LibYolov3 with
    the same seeded initialisation, concatenated heads, oracle.losses.yolov3_loss once, autograd -- loss and parameter gradients of
    the two-rank HIP run must agree with it to 1e-3 (fp32)."""
    import socket
    import torch.multiprocessing as mp
    from oracle import train as otrain
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.start_processes(_dp_sum_worker, args=(2, port, out), nprocs=2, join=True, start_method='spawn')
    (g0, total0, share0), (g1, total1, share1) = out[0], out[1]
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    ref, ref_crit = otrain.make_library(20220504)
    images, tg = synthetic_batch(4, 64, seed=77)
    preds = [ref(images[r * 2:(r + 1) * 2]) for r in range(2)]          # what nn.DataParallel's replicas compute
    gathered = [torch.cat([preds[0][l], preds[1][l]], 0) for l in range(3)]
    loss = ref_crit(gathered, tg)                                       # ONE loss on the gathered batch
    loss.backward()
    want = {k: p.grad.detach() for k, p in ref.named_parameters()}
    assert abs(total0 - float(loss)) <= 1e-4 * abs(float(loss)) and total0 == total1 and share0 != share1
    worst = 0.0
    for k in g0:
        err = ((g0[k] - want[k]).abs().max() / want[k].abs().max().clamp_min(1e-12)).item()
        worst = max(worst, err)
        assert err < 1e-3, f'{k}: {err}'
    print(f'two ranks vs the oracle on the gathered batch: loss {total0:.6f} vs {float(loss):.6f}, worst gradient deviation {worst:.2e}')


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The REAL `python bench.py --gpus 2` path -- the parent starts torch.distributed.run with two ranks of itself before it touches
    the GPU (the reference wraps its model in nn.DataParallel, demos/yolov3_u/train.py:85) -- on the one GPU of this box, with the
    collectives over gloo (RCCL refuses two ranks on one device): rendezvous, broadcast, job-wide loss normalisation, gradient
    buckets launched from the autograd hooks, FusedAdam, the JSON contract incl. the `dp` block a driver checks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['FVA_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                        '--batch', '8', '--size', '320'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    dp = d['dp']
    print(f"bench --gpus 2 over gloo on one GPU: {d['value']} img/s, {d['ms_per_step']} ms/step, loss {d['loss']}, dp {dp}")
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 16
    assert np.isfinite(d['loss']) and d['value'] > 0
    assert dp['backend'] == 'gloo' and dp['world_size_seen_by_backend'] == 2 and dp['reduce_op'] == 'sum' and dp['wire_dtype'] == 'float32'
    assert dp['collectives_per_step'] == dp['buckets'] and dp['wire_bytes_per_step'] == 4 * 61949149
    assert dp['buckets_launched_before_backward_ended'] >= dp['buckets'] - 2       # all but the last buckets flew during backward
    assert dp['parameters_identical_across_ranks'] is True


def _nccl_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    import fastvision_amd
    from fastvision_amd import FusedAdam, parallel
    parallel.init_from_env('nccl')
    dev = f'cuda:{rank}'
    torch.manual_seed(20220504 + rank)
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                 in_channels=3, num_classes=80, training=True).to(dev).train()
    parallel.broadcast_parameters(net)
    crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5).data_parallel()
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
    red = parallel.GradientReducer(net.parameters(), average=False, bucket_dtype=torch.bfloat16)
    images, tg = synthetic_batch(4, 128, rank=rank)
    launched = 0
    for _ in range(2):
        opt.zero_grad()
        loss = crit(net(images.to(dev)), tg.to(dev))
        loss.backward()
        launched = red.next_launch
        red.finish()
        opt.step()
    torch.cuda.synchronize()
    out[rank] = ({k: p.detach().cpu().clone() for k, p in list(net.named_parameters())[::17]}, launched, len(red.buckets))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: RCCL refuses two ranks on one device')
def test_two_ranks_over_rccl():
    """The N > 1 path on its real backend: two ranks on two GPUs, RCCL all-reduce of bf16 gradient buckets launched from the
    autograd hooks on the side stream, job-wide loss normalisation, FusedAdam -- both ranks must hold bit-identical parameters
    after two steps, and all but the last two buckets must have been launched before backward ended."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.start_processes(_nccl_worker, args=(2, port, out), nprocs=2, join=True, start_method='spawn')
    (w0, l0, nb), (w1, l1, _) = out[0], out[1]
    assert l0 == l1 and l0 >= nb - 2
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k


def test_loss_gradients_follow_the_upstream_scalar_and_are_spent_by_backward(gold_lib, gold_demo):
    """Round 4: the fused losses scale their stored head gradients by the upstream scalar in place (an empty launch for the 1 that
    ``loss.backward()`` passes).  (loss * 0.5).backward() must give half the gradients of loss.backward() -- exactly: a power of two --
    on both surfaces (to rounding), and a second backward pass through the same loss node raises instead of scaling twice."""
    from fastvision_amd.demos.yolov3_u.utils import ComputeLoss

    class M:
        anchors = coco_anchors_feature()

    def grads(scale, lossf, gold):
        hs = [T(gold[f'g3_syn_head{l}']).to(DEV).requires_grad_(True) for l in range(3)]
        loss = lossf(hs, T(gold['g3_syn_targets']).to(DEV))
        (loss * scale).backward(retain_graph=True)
        with pytest.raises(RuntimeError, match='already been released'):
            loss.backward()
        return [h.grad.clone() for h in hs]

    crit, cl = lib_loss(), ComputeLoss()
    for lossf, gold in ((lambda hs, tg: crit(hs, tg), gold_lib), (lambda hs, tg: cl(hs, tg, M()), gold_demo)):
        full, half = grads(1.0, lossf, gold), grads(0.5, lossf, gold)
        for a, b in zip(full, half):
            # (two forward passes: the loss kernels' few fp32 atomics may land in another order, so equal to rounding, not bit for bit)
            assert a.abs().max() > 0 and torch.allclose(a * 0.5, b, rtol=1e-5, atol=1e-9)
            assert not torch.allclose(a, b, rtol=1e-2, atol=0)
