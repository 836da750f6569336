#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  The reference is
imported in place behind the four harness shims SURVEY.md section 8c documents (package-name symlink in
/tmp, stub modules for the absent torchvision/cv2, the never-defined ``offset`` helper, the torch>=2
integer ``clamp_`` bound).  No reference source is copied: the fixtures are inputs and outputs only.

    python oracle/make_golden.py lib      # library surface  (fastvision.*)
    python oracle/make_golden.py lib_curves   # the reference's own 100-step curve at 1 / 3 / 8 threads (appended to lib.npz)
    python oracle/make_golden.py demo     # demo surface     (demos/yolov3_u)   -- separate process: its
                                          # top-level ``utils``/``models`` names clash with nothing else then
    python oracle/make_golden.py eval_lib / eval_demo   # validation side (decode, NMS wrappers, mAP): scope row f-2
    python oracle/make_golden.py pipeline / pipeline_demo   # input side (resize, pad, flips, normalise, labels): row f-3
    python oracle/make_golden.py rpn / faster   # two-stage head (RPN ops and module; the whole Faster R-CNN step): row f-4
    python oracle/make_golden.py all      # everything, one child process per surface

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import subprocess
import sys
import types
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, 'tests', 'golden')
REF = '/root/reference'
BOOT = '/tmp/fvoracle'
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')


def _stub_missing():
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            return 0
    for n in ('torchvision', 'torchvision.ops', 'torchvision.transforms', 'cv2'):
        sys.modules[n] = _Any(n)


def _patch_clamp():
    import torch
    orig = torch.Tensor.clamp_

    def clamp_(self, min=None, max=None):
        if not self.is_floating_point():
            if isinstance(min, torch.Tensor) and min.dim() == 0:
                min = int(min.item())
            if isinstance(max, torch.Tensor) and max.dim() == 0:
                max = int(max.item())
        return orig(self, min, max)
    torch.Tensor.clamp_ = clamp_


def boot_lib():
    os.makedirs(BOOT, exist_ok=True)
    link = os.path.join(BOOT, 'fastvision')
    if not os.path.islink(link):
        os.symlink(REF, link)
    sys.path.insert(0, BOOT)
    _stub_missing()
    _patch_clamp()
    import fastvision.detection.tools as T
    T.offset = lambda h, w, mode='xy': T.grid(h, w, mode=mode, dtype='numpy')
    return T


def boot_demo():
    _stub_missing()
    _patch_clamp()
    sys.path.insert(0, os.path.join(REF, 'demos', 'yolov3_u'))


sys.path.insert(0, REPO)
from fastvision_amd.synthetic import synthetic_batch, coco_anchors_px, coco_anchors_feature  # noqa: E402


def rand_targets(gen, T, batch):
    """[T,6] targets incl. awkward ones: tiny/huge boxes, centres on cell borders and at 0 / just below 1."""
    import torch
    if T == 0:
        return torch.zeros(0, 6)
    b = torch.randint(0, batch, (T,), generator=gen).sort()[0].float()
    cls = torch.randint(0, 80, (T,), generator=gen).float()
    wh = torch.exp(np.log(0.005) + (np.log(0.99) - np.log(0.005)) * torch.rand(T, 2, generator=gen))
    xy = torch.rand(T, 2, generator=gen)
    if T >= 7:
        xy[0] = torch.tensor([0.0, 0.0])
        xy[1] = torch.tensor([0.999999, 0.999999])
        xy[2] = torch.tensor([0.5, 0.25])          # exactly on cell borders for every grid used
        xy[3] = torch.tensor([1.0, 1.0])           # needs the clamp (library) -- not used with the demo
        wh[4] = torch.tensor([1.0, 1.0])
        wh[5] = torch.tensor([0.001, 0.9])
    return torch.cat([b.view(-1, 1), cls.view(-1, 1), xy, wh], dim=1)


def stats(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item()] + t[:4].tolist() + [0.0] * max(0, 4 - t.numel()))


# ====================================================================================== library surface
def gen_lib():
    import torch
    T = boot_lib()
    from fastvision.classfication.models.darknet53 import darknet53, ConvBlock3x3, ConvBlock1x1, ResidualBlock
    from fastvision.detection.neck.yolov3neck import yolov3neck, UpSampling
    from fastvision.detection.head.yolov3head import yolov3head
    from fastvision.detection.models.yolov3 import yolov3
    from fastvision.loss.yolov3_loss import Yolov3Loss
    from fastvision.loss.iou_loss import CIOULoss
    from fastvision.loss.classification_loss import BiCrossEntropyLoss

    def build(seed, training=True):
        torch.manual_seed(seed)
        m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                   num_anchors_per_level=[3, 3, 3], in_channels=3, num_classes=80, training=training)
        m.train(training)
        return m

    out = {}
    # ---- G1: matcher (build_target) -------------------------------------------------------------------
    net = build(0)
    crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
    case = 0
    for grids in ((13, 26, 52), (20, 40, 80), (2, 4, 8)):
        for T_ in (0, 1, 7, 64, 234):
            for seed in range(3):
                g = torch.Generator().manual_seed(1000 * seed + T_)
                tg = rand_targets(g, T_, 8)
                shapes = [torch.zeros(8, 3, s, s, 1) for s in grids]
                locs, cats, xywh, anc = crit.build_target(shapes, tg)
                out[f'g1_{case}_grids'] = np.array(grids)
                out[f'g1_{case}_targets'] = tg.numpy()
                for l in range(3):
                    out[f'g1_{case}_l{l}_b'] = locs[l][0].numpy()
                    out[f'g1_{case}_l{l}_gxy'] = locs[l][1].numpy()
                    out[f'g1_{case}_l{l}_a'] = locs[l][2].numpy()
                    out[f'g1_{case}_l{l}_cls'] = cats[l].numpy()
                    out[f'g1_{case}_l{l}_xywh'] = xywh[l].numpy()
                    out[f'g1_{case}_l{l}_anc'] = anc[l].numpy()
                case += 1
    out['g1_cases'] = np.array(case)
    # non-square grid (H != W): the [W,H,W,H] scale and the x/y clamp bounds differ
    g = torch.Generator().manual_seed(77)
    tg = rand_targets(g, 64, 2)
    shapes = [torch.zeros(2, 3, 20, 15, 1), torch.zeros(2, 3, 40, 30, 1), torch.zeros(2, 3, 80, 60, 1)]
    locs, cats, xywh, anc = crit.build_target(shapes, tg)
    out['g1ns_targets'] = tg.numpy()
    for l in range(3):
        out[f'g1ns_l{l}_b'], out[f'g1ns_l{l}_gxy'], out[f'g1ns_l{l}_a'] = [x.numpy() for x in locs[l]]
        out[f'g1ns_l{l}_cls'], out[f'g1ns_l{l}_xywh'], out[f'g1ns_l{l}_anc'] = cats[l].numpy(), xywh[l].numpy(), anc[l].numpy()

    # ---- G2: IoU family known answers ----------------------------------------------------------------
    g = torch.Generator().manual_seed(2)
    n = 256
    a = torch.rand(n, 4, generator=g) * 10
    b = torch.rand(n, 4, generator=g) * 10
    a[:, 2:] = a[:, :2] + torch.rand(n, 2, generator=g) * 8      # valid xyxy
    b[:, 2:] = b[:, :2] + torch.rand(n, 2, generator=g) * 8
    edge_a = torch.tensor([[0, 0, 2, 2], [0, 0, 1, 1], [0, 0, 1, 1], [1, 1, 1, 1], [0, 0, 4, 4], [0, 0, 2, 2.]])
    edge_b = torch.tensor([[0, 0, 2, 2], [2, 2, 3, 3], [1, 0, 2, 1], [1, 1, 1, 1], [1, 1, 2, 2], [0, 0, 2, 4.]])
    a, b = torch.cat([a, edge_a]), torch.cat([b, edge_b])
    wa = torch.cat([a[:, 2:] - a[:, :2]])
    wb = torch.cat([b[:, 2:] - b[:, :2]])
    xa, xb = T.xyxy2xywh(a), T.xyxy2xywh(b)
    out['g2_a'], out['g2_b'] = a.numpy(), b.numpy()
    out['g2_xyxy_iou'] = T.xyxy_iou(a, b).numpy()
    out['g2_xywh_iou'] = T.xywh_iou(xa, xb).numpy()
    out['g2_wh_iou'] = T.wh_iou(wa, wb).numpy()
    out['g2_xyxy_iou_batch'] = T.xyxy_iou_batch(a[:40], b[:24]).numpy()
    out['g2_xywh_iou_batch'] = T.xywh_iou_batch(xa[:40], xb[:24]).numpy()
    out['g2_wh_iou_batch'] = T.wh_iou_batch(wa[:40], wb[:24]).numpy()
    out['g2_giou'] = T.GIOU(a, b).numpy()
    out['g2_diou'] = T.DIOU(a, b).numpy()
    out['g2_ciou'] = T.CIOU(a, b).numpy()
    out['g2_ciou_xywh'] = T.CIOU(xa, xb, mode='xywh').numpy()
    out['g2_xywh2xyxy'] = T.xywh2xyxy(xa).numpy()
    out['g2_xyxy2xywhn'] = T.xyxy2xywhn(a, 480, 640).numpy()
    out['g2_grid_xy'] = T.grid(3, 5, mode='xy').numpy()
    out['g2_grid_yx'] = T.grid(3, 5, mode='yx').numpy()
    ar = a.clone().requires_grad_(True)
    out['g2_cioul'] = CIOULoss('mean')(ar, b).detach().numpy().reshape(1)
    CIOULoss('mean')(ar, b).backward()
    out['g2_cioul_grad'] = ar.grad.numpy()
    p = torch.rand(12, 5, generator=g)
    lab = torch.randint(0, 5, (12,), generator=g)
    out['g2_bce_p'], out['g2_bce_lab'] = p.numpy(), lab.numpy()
    out['g2_bce_mean'] = BiCrossEntropyLoss('mean')(p, lab, already_sigmoid=True).numpy().reshape(1)
    out['g2_bce_logits_sum'] = BiCrossEntropyLoss('sum')(p * 4 - 2, lab).numpy().reshape(1)

    # ---- G3: Yolov3Loss value + d loss / d head_out --------------------------------------------------
    def run_loss(tag, tg, grids, batch, seed, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        heads = [(torch.randn(batch, 3, s, s, 85, generator=g) * scale).requires_grad_(True) for s in grids]
        loss = crit(heads, tg)
        loss.backward()
        out[f'g3_{tag}_targets'] = tg.numpy()
        out[f'g3_{tag}_grids'] = np.array(grids)
        out[f'g3_{tag}_loss'] = loss.detach().numpy()
        for l, h in enumerate(heads):
            out[f'g3_{tag}_head{l}'] = h.detach().numpy()
            out[f'g3_{tag}_grad{l}'] = h.grad.numpy()

    g = torch.Generator().manual_seed(5)
    run_loss('rand', rand_targets(g, 9, 2), (2, 4, 8), 2, 11)
    run_loss('empty', torch.zeros(0, 6), (2, 4, 8), 2, 12)
    dup = torch.tensor([[0, 3, 0.30, 0.30, 0.20, 0.25], [0, 5, 0.31, 0.32, 0.22, 0.24],
                        [1, 7, 0.70, 0.60, 0.50, 0.45], [1, 7, 0.72, 0.61, 0.45, 0.50],
                        [1, 9, 0.10, 0.90, 0.05, 0.06]])
    run_loss('dup', dup, (2, 4, 8), 2, 13, scale=2.0)
    _, tg = synthetic_batch(4, 128)
    run_loss('syn', tg, (4, 8, 16), 4, 14)

    # ---- G4: block-level conv/BN/SiLU fixtures --------------------------------------------------------
    def run_block(tag, mod, x, extra=None):
        mod.train()
        x = x.clone().requires_grad_(True)
        y = mod(x) if extra is None else extra(mod, x)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
        (y * gy).sum().backward()
        out[f'g4_{tag}_x'], out[f'g4_{tag}_y'], out[f'g4_{tag}_gy'] = x.detach().numpy(), y.detach().numpy(), gy.numpy()
        out[f'g4_{tag}_dx'] = x.grad.numpy()
        for k, v in mod.state_dict().items():
            out[f'g4_{tag}_sd_{k}'] = v.numpy()
        for k, v in mod.named_parameters():
            out[f'g4_{tag}_gr_{k}'] = v.grad.numpy()

    g = torch.Generator().manual_seed(3)
    torch.manual_seed(41); m = ConvBlock3x3(32, 64)
    out['g4_cb3_w0'] = m.conv.weight.detach().numpy().copy()
    run_block('cb3', m, torch.randn(2, 32, 10, 12, generator=g))
    torch.manual_seed(42); m = ConvBlock3x3(32, 64, stride=(2, 2))
    out['g4_cb3s2_w0'] = m.conv.weight.detach().numpy().copy()
    run_block('cb3s2', m, torch.randn(2, 32, 10, 12, generator=g))
    torch.manual_seed(43); m = ConvBlock1x1(64, 32)
    out['g4_cb1_w0'] = m.conv.weight.detach().numpy().copy()
    run_block('cb1', m, torch.randn(2, 64, 6, 6, generator=g))
    torch.manual_seed(44); m = ResidualBlock(64, 32)
    out['g4_res_w1'] = m.conv1.conv.weight.detach().numpy().copy()
    out['g4_res_w2'] = m.conv2.conv.weight.detach().numpy().copy()
    run_block('res', m, torch.randn(2, 64, 8, 8, generator=g))
    torch.manual_seed(45); m = UpSampling(64, 32)
    out['g4_up_w0'] = m.squeeze.conv.weight.detach().numpy().copy()
    skip = torch.randn(2, 32, 8, 8, generator=g)
    out['g4_up_skip'] = skip.numpy()
    run_block('up', m, torch.randn(2, 64, 4, 4, generator=g), extra=lambda mod, x: torch.cat([mod(x), skip], dim=1))

    # ---- G5: whole model: init checksums, head outputs, grads and BN buffers after one step ----------
    seed = 20220504
    net = build(seed)
    sd = net.state_dict()
    out['g5_keys'] = np.array(list(sd.keys()))
    out['g5_init'] = np.stack([stats(v.float()) for v in sd.values()])
    images, tg = synthetic_batch(2, 64)
    out['g5_targets'] = tg.numpy()
    crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
    pred = net(images)
    loss = crit(pred, tg)
    loss.backward()
    out['g5_loss'] = loss.detach().numpy()
    for l, h in enumerate(pred):
        out[f'g5_head{l}'] = h.detach().numpy()
    out['g5_gradkeys'] = np.array([k for k, _ in net.named_parameters()])
    out['g5_grads'] = np.stack([stats(p.grad) for _, p in net.named_parameters()])
    out['g5_after'] = np.stack([stats(v.float()) for v in net.state_dict().values()])
    # eval-branch decode (inferred ``offset``, App. B-14)
    net.eval()
    with torch.no_grad():
        _, dec = net(images, val=True)
    out['g5_decode'] = dec.numpy()

    # ---- G6: 100-step loss curve, fixed synthetic batch, B=2 S=128, Adam(1e-4) ------------------------
    if os.environ.get('GOLDEN_SKIP_CURVE') != '1':
        net = build(seed)
        crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        images, tg = synthetic_batch(2, 128)
        curve = []
        for _ in range(100):
            pred = net(images)
            opt.zero_grad()
            loss = crit(pred, tg)
            loss.backward()
            opt.step()
            curve.append(loss.item())
        out['g6_curve'] = np.array(curve)
    np.savez_compressed(os.path.join(GOLD, 'lib.npz'), **out)
    print('lib fixtures:', len(out), 'arrays')


def gen_lib_curves():
    """The reference's OWN spread on the library 100-step curve: the same 100 steps (G6) with 1, 3 and 8 intra-op threads.  Only
    the order of the floating-point reductions changes with the thread count, yet the trajectory of this loss -- whose objectness
    target is the non-detached IoU of the prediction (loss/yolov3_loss.py:60-61) -- drifts by ~1e-2 after a few dozen steps.  The
    GPU test bounds its own deviation by this envelope instead of a hand-picked number.  Appends g6_curve_t{1,3,8} to lib.npz."""
    import torch
    boot_lib()
    from fastvision.classfication.models.darknet53 import darknet53
    from fastvision.detection.neck.yolov3neck import yolov3neck
    from fastvision.detection.head.yolov3head import yolov3head
    from fastvision.detection.models.yolov3 import yolov3
    from fastvision.loss.yolov3_loss import Yolov3Loss
    path = os.path.join(GOLD, 'lib.npz')
    out = dict(np.load(path, allow_pickle=False))
    for threads in (1, 3, 8):
        torch.set_num_threads(threads)
        torch.manual_seed(20220504)
        net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                     num_anchors_per_level=[3, 3, 3], in_channels=3, num_classes=80, training=True)
        net.train()
        crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        images, tg = synthetic_batch(2, 128)
        curve = []
        for _ in range(100):
            pred = net(images)
            opt.zero_grad()
            loss = crit(pred, tg)
            loss.backward()
            opt.step()
            curve.append(loss.item())
        out[f'g6_curve_t{threads}'] = np.array(curve)
        dev = np.abs(out[f'g6_curve_t{threads}'] - out['g6_curve']) / np.abs(out['g6_curve'])
        print(f'threads {threads}: max rel deviation from g6_curve {dev.max():.3e} (first 10 steps {dev[:10].max():.3e})', flush=True)
    np.savez_compressed(path, **out)
    print('lib.npz now holds', len(out), 'arrays')


# ====================================================================================== demo surface
def gen_demo():
    import torch
    boot_demo()
    from models.yolov3 import YoloV3
    from utils.lossv3 import ComputeLoss
    import utils.iou as DI
    import builtins
    out = {}
    anchors = coco_anchors_feature()

    def build(seed):
        torch.manual_seed(seed)
        m = YoloV3(in_channels=3, num_classes=80, anchors=anchors)
        m.train()
        return m

    class _Shell:                      # ComputeLoss only reads model.anchors
        pass
    shell = _Shell()
    shell.anchors = anchors
    crit = ComputeLoss()
    real_print = builtins.print
    captured = []

    def quiet(*a, **k):                # the loss prints its four partial losses (lossv3.py:108): capture them
        captured.append(a)

    def run_loss(tag, tg, grids, batch, seed, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        heads = [(torch.randn(batch, 255, s, s, generator=g) * scale).requires_grad_(True) for s in grids]
        builtins.print = quiet
        try:
            loss = crit(heads, tg, shell)
        finally:
            builtins.print = real_print
        loss.backward()
        out[f'g3_{tag}_targets'], out[f'g3_{tag}_grids'] = tg.numpy(), np.array(grids)
        out[f'g3_{tag}_loss'] = loss.detach().numpy()
        out[f'g3_{tag}_parts'] = np.array(captured[-1], dtype=np.float64)
        for l, h in enumerate(heads):
            out[f'g3_{tag}_head{l}'] = h.detach().numpy()
            out[f'g3_{tag}_grad{l}'] = h.grad.numpy()

    _, tg = synthetic_batch(2, 64)
    run_loss('syn', tg, (2, 4, 8), 2, 21)
    _, tg = synthetic_batch(4, 128, seed=99)
    run_loss('syn4', tg, (4, 8, 16), 4, 22, scale=2.0)
    dup = torch.tensor([[0, 3, 0.30, 0.30, 0.20, 0.25], [0, 5, 0.31, 0.32, 0.22, 0.24],
                        [1, 7, 0.70, 0.60, 0.50, 0.45], [1, 7, 0.72, 0.61, 0.45, 0.50]])
    run_loss('dup', dup, (2, 4, 8), 2, 23)

    # demo IoU variants (centre sums, minus sign) -- demos/yolov3_u/utils/iou.py
    g = torch.Generator().manual_seed(2)
    a = torch.rand(64, 4, generator=g) * 10
    b = torch.rand(64, 4, generator=g) * 10
    a[:, 2:] = a[:, :2] + torch.rand(64, 2, generator=g) * 8
    b[:, 2:] = b[:, :2] + torch.rand(64, 2, generator=g) * 8
    out['g2_a'], out['g2_b'] = a.numpy(), b.numpy()
    out['g2_diou'] = DI.DIOU(a, b).numpy()
    out['g2_ciou'] = DI.CIOU(a, b).numpy()

    # whole model
    seed = 20220504
    net = build(seed)
    sd = net.state_dict()
    out['g5_keys'] = np.array(list(sd.keys()))
    out['g5_init'] = np.stack([stats(v.float()) for v in sd.values()])
    images, tg = synthetic_batch(2, 64)
    pred = net(images)
    builtins.print = quiet
    try:
        loss = crit(pred, tg, net)
    finally:
        builtins.print = real_print
    loss.backward()
    out['g5_loss'] = loss.detach().numpy()
    for l, h in enumerate(pred):
        out[f'g5_head{l}'] = h.detach().numpy()
    out['g5_gradkeys'] = np.array([k for k, _ in net.named_parameters()])
    out['g5_grads'] = np.stack([stats(p.grad) for _, p in net.named_parameters()])
    out['g5_after'] = np.stack([stats(v.float()) for v in net.state_dict().values()])

    if os.environ.get('GOLDEN_SKIP_CURVE') != '1':
        net = build(seed)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        images, tg = synthetic_batch(2, 128)
        curve = []
        builtins.print = quiet
        try:
            for _ in range(100):
                pred = net(images)
                opt.zero_grad()
                loss = crit(pred, tg, net)
                loss.backward()
                opt.step()
                curve.append(loss.item())
        finally:
            builtins.print = real_print
        out['g6_curve'] = np.array(curve)
    np.savez_compressed(os.path.join(GOLD, 'demo.npz'), **out)
    print('demo fixtures:', len(out), 'arrays')


# ====================================================================================== validation side (scope row f-2)
def _install_nms():
    """torchvision is absent, so the reference's NMS wrappers run with the oracle's restatement of torchvision.ops.nms
    standing in for it (oracle/detect.py: the suppression core itself stays 'parity unpinned')."""
    from oracle.detect import nms
    tv, ops = sys.modules['torchvision'], sys.modules['torchvision.ops']
    tv.ops = ops
    ops.nms = nms


def _numpy_aliases():
    """fifth harness shim: metrics/map.py spells float64 / int64 as np.float / np.long (numpy < 1.24)."""
    np.float = float
    np.long = np.int64
    if not hasattr(np, 'trapz'):
        np.trapz = np.trapezoid


def synth_predictions(gen, R, n_obj, size, conf_low=0.02):
    """[R,85] decoded-looking rows: n_obj clusters of overlapping boxes with high objectness over a low-conf background,
    a few exact duplicates (score ties) included."""
    import torch
    p = torch.rand(R, 85, generator=gen)
    p[:, 0:2] = torch.rand(R, 2, generator=gen) * size
    p[:, 2:4] = 8 + torch.rand(R, 2, generator=gen) * size / 3
    p[:, 4] = torch.rand(R, generator=gen) * conf_low
    per = max(1, R // (4 * max(n_obj, 1)))
    for o in range(n_obj):
        c = torch.rand(4, generator=gen)
        base = torch.tensor([c[0] * size, c[1] * size, 20 + c[2] * size / 2, 20 + c[3] * size / 2])
        idx = torch.randperm(R, generator=gen)[:per]
        p[idx, 0:4] = base + torch.randn(per, 4, generator=gen) * 3
        p[idx, 4] = 0.3 + 0.7 * torch.rand(per, generator=gen)
        cls = int(torch.randint(0, 80, (1,), generator=gen))
        p[idx, 5 + cls] = 0.9 + 0.1 * torch.rand(per, generator=gen)
        if per >= 3:
            p[idx[1]] = p[idx[0]]              # exact duplicate: tie in score, IoU 1
    return p


def gen_eval_lib():
    import torch
    T = boot_lib()
    _install_nms()
    _numpy_aliases()
    from fastvision.classfication.models.darknet53 import darknet53
    from fastvision.detection.neck.yolov3neck import yolov3neck
    from fastvision.detection.head.yolov3head import yolov3head
    from fastvision.detection.models.yolov3 import yolov3
    from fastvision.detection.tools.NMS import non_max_suppression
    from fastvision.metrics.map import CalculateMAP
    out = {}
    # ---- E1: eval branch of Yolov3.forward (decode) on the whole model, S = 64 and 96 ---------------------
    for case, (S, B, seed) in enumerate(((64, 2, 11), (96, 1, 12))):
        torch.manual_seed(seed)
        m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                   num_anchors_per_level=[3, 3, 3], in_channels=3, num_classes=80, training=False)
        m.eval()
        g = torch.Generator().manual_seed(seed)
        # give BatchNorm non-trivial running statistics, as a trained checkpoint has
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
        images = torch.rand(B, 3, S, S, generator=g)
        with torch.no_grad():
            head_out, results = m(images)
        out[f'e1_{case}_cfg'] = np.array([S, B, seed])
        out[f'e1_{case}_images'] = images.numpy()
        for l in range(3):
            out[f'e1_{case}_head{l}'] = head_out[l].numpy()
        out[f'e1_{case}_results'] = results.numpy()
    # ---- E2: non_max_suppression (NMS.py) around the restated nms --------------------------------------------
    cases = [(2000, 6, 416, 0.25, 0.45, 300, 0.02), (500, 3, 416, 0.25, 0.45, 5, 0.02), (300, 0, 416, 0.25, 0.45, 300, 0.02),
             (700, 4, 640, 0.001, 0.6, 300, 1.0), (64, 2, 64, 0.25, 0.45, 300, 0.02), (1, 1, 64, 0.0, 0.45, 300, 1.0)]
    for case, (R, n_obj, size, ct, it, md, low) in enumerate(cases):
        g = torch.Generator().manual_seed(500 + case)
        pred = synth_predictions(g, R, n_obj, size, low)
        sc, cat, box = non_max_suppression(pred.clone(), conf_thres=ct, iou_thres=it, max_det=md)
        out[f'e2_{case}_cfg'] = np.array([ct, it, md], dtype=np.float64)
        out[f'e2_{case}_pred'] = pred.numpy()
        out[f'e2_{case}_scores'] = sc.numpy().reshape(-1)
        out[f'e2_{case}_cats'] = cat.numpy().reshape(-1).astype(np.int64)
        out[f'e2_{case}_boxes'] = box.numpy().reshape(-1, 4)
    # ---- E3: CalculateMAP over a few synthetic images --------------------------------------------------------
    est = CalculateMAP(map_iou_values=np.linspace(0.5, 0.95, 10))
    g = torch.Generator().manual_seed(77)
    n_img = 6
    for i in range(n_img):
        nt = [5, 0, 9, 3, 12, 1][i]
        tcls = torch.randint(0, 4, (nt,), generator=g).float()
        txy = torch.rand(nt, 2, generator=g) * 300
        twh = 20 + torch.rand(nt, 2, generator=g) * 100
        target = torch.cat([tcls.view(-1, 1), txy, txy + twh], dim=1)
        keep = torch.rand(nt, generator=g) > 0.25                     # detected targets, jittered
        pbox = target[keep, 1:] + torch.randn(int(keep.sum()), 4, generator=g) * 6
        pcls = target[keep, 0].clone()
        nf = [3, 2, 4, 0, 5, 0][i]                                    # false positives
        fxy = torch.rand(nf, 2, generator=g) * 300
        fbox = torch.cat([fxy, fxy + 30 + torch.rand(nf, 2, generator=g) * 60], dim=1)
        fcls = torch.randint(0, 5, (nf,), generator=g).float()
        box = torch.cat([pbox, fbox])
        cls = torch.cat([pcls, fcls])
        conf = torch.rand(box.size(0), generator=g)
        if i == 2 and box.size(0) >= 2:
            box[1] = box[0]
            cls[1] = cls[0]                                           # duplicate detection of one target
        pred = torch.cat([cls.view(-1, 1), conf.view(-1, 1), box], dim=1)
        est.process_one(pred, target)
        out[f'e3_{i}_pred'] = pred.numpy()
        out[f'e3_{i}_target'] = target.numpy()
    for i, c in enumerate(est.correct_all_images):
        out[f'e3_correct{i}'] = c
    out['e3_n'] = np.array([n_img, len(est.correct_all_images)])
    map_iou, map_cls, cls_idx = est.fetch()
    out['e3_map_each_iou'] = map_iou
    out['e3_map_each_cls'] = map_cls
    out['e3_cls_idx'] = np.array(cls_idx)
    np.savez_compressed(os.path.join(GOLD, 'eval_lib.npz'), **out)
    print('eval_lib fixtures:', len(out), 'arrays')


def gen_eval_demo():
    import torch
    boot_demo()
    for n in ('albumentations', 'albumentations.pytorch'):
        sys.modules[n] = type(sys.modules['cv2'])(n)
    _install_nms()
    _numpy_aliases()
    # inference.py is a script: it calls Inference() (GPU + COCO files) at import time.  Execute it as a module and keep
    # what it had defined by then (postProcess is all that is used).
    import importlib.util
    spec = importlib.util.spec_from_file_location('inference', os.path.join(REF, 'demos', 'yolov3_u', 'inference.py'))
    INF = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(INF)
    except (RuntimeError, AssertionError, FileNotFoundError):
        pass
    from utils.nms import non_max_suppression, non_max_suppression_batch
    out = {}
    strides = [32, 16, 8]
    anchors = [a.view(-1, 2) for a in coco_anchors_feature()]
    # ---- D1: postProcess (decode, un-letterbox, clamps, size filter, NMS) on raw head tensors -----------------
    for case, (S, rr, pl, pt, ow, oh, ct) in enumerate(((64, 0.1, 0, 8, 640, 480, 0.3), (96, 0.2, 12, 0, 360, 480, 0.5))):
        g = torch.Generator().manual_seed(900 + case)
        layers = [torch.randn(1, 255, S // s, S // s, generator=g) * 1.5 for s in strides]
        sc, cat, box = INF.postProcess([l.clone() for l in layers], strides, anchors, ct, 0.45, rr, pl, pt, ow, oh)
        out[f'd1_{case}_cfg'] = np.array([S, rr, pl, pt, ow, oh, ct, 0.45], dtype=np.float64)
        for l in range(3):
            out[f'd1_{case}_layer{l}'] = layers[l].numpy()
        out[f'd1_{case}_scores'] = sc.numpy().reshape(-1)
        out[f'd1_{case}_cats'] = cat.numpy().reshape(-1)
        out[f'd1_{case}_boxes'] = box.numpy().reshape(-1, 4)
    # ---- D2: non_max_suppression (xyxy rows in) and non_max_suppression_batch (xywh rows in) -------------------
    for case, (R, n_obj, ct, it, md) in enumerate(((1500, 5, 0.25, 0.45, 300), (400, 3, 0.25, 0.45, 4), (200, 0, 0.25, 0.45, 300))):
        g = torch.Generator().manual_seed(950 + case)
        pred = synth_predictions(g, R, n_obj, 416)
        xyxy = pred.clone()
        xyxy[:, 2:4] = xyxy[:, 0:2] + pred[:, 2:4]
        res = non_max_suppression(xyxy.clone(), conf_thres=ct, iou_thres=it, max_det=md)
        resb = non_max_suppression_batch([pred.clone(), pred.flip(0).clone()], conf_thres=ct, iou_thres=it, max_det=md)
        out[f'd2_{case}_cfg'] = np.array([ct, it, md], dtype=np.float64)
        out[f'd2_{case}_pred'] = pred.numpy()
        out[f'd2_{case}_single'] = res.numpy().reshape(-1, 6)
        out[f'd2_{case}_batch0'] = resb[0].numpy().reshape(-1, 6)
        out[f'd2_{case}_batch1'] = resb[1].numpy().reshape(-1, 6)
    np.savez_compressed(os.path.join(GOLD, 'eval_demo.npz'), **out)
    print('eval_demo fixtures:', len(out), 'arrays')


# ====================================================================================== input side (scope row f-3)
def _install_cv2():
    """cv2 is absent: the reference's dataset classes run with the oracle's restatements of the four OpenCV calls they make
    (oracle/pipeline.py: the resampler itself stays 'parity unpinned'); imread serves synthetic images from memory."""
    from oracle import pipeline as P
    cv2 = sys.modules['cv2']
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT, cv2.COLOR_BGR2RGB = 1, 0, 4
    cv2.resize = lambda img, dsize, interpolation=1: P.resize_linear_u8(img, dsize)
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=0: P.copy_make_border_constant(img, t, b, l, r, value)
    cv2.flip = P.flip
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    cv2._images = {}
    cv2.imread = lambda path: cv2._images[path].copy()
    return cv2


def synth_image(gen, h, w):
    """uint8 HWC image with structure at several scales (so that resampling errors show)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), dtype=np.float32)
    for c in range(3):
        f = gen.uniform(0.01, 0.3, size=4)
        img[..., c] = 127 + 60 * np.sin(xx * f[0] + yy * f[1] + c) + 50 * np.cos(xx * f[2] - yy * f[3])
    img += gen.normal(0, 12, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_boxes(gen, n, h, w):
    x0, y0 = gen.uniform(0, w * 0.7, n), gen.uniform(0, h * 0.7, n)
    bw, bh = gen.uniform(4, w * 0.3, n), gen.uniform(4, h * 0.3, n)
    cls = gen.integers(0, 80, n)
    return [(float(c), float(a), float(b), float(a + d), float(b + e)) for c, a, b, d, e in zip(cls, x0, y0, bw, bh)]


# (source h, w, input size, boxes): small on purpose (fixtures stay small); 256x256 -> 128 is the exact-2x (area) case,
# 128x128 -> 128 the identity, the last one a full-size 640 sample kept as a strided subsample + checksums
PIPE_CASES = [(120, 160, 128, 5), (111, 167, 128, 3), (160, 107, 128, 7), (256, 256, 128, 2), (100, 37, 64, 1), (128, 128, 128, 4),
              (75, 248, 96, 6), (97, 211, 64, 2), (480, 640, 640, 5)]


def gen_pipeline():
    boot_lib()
    cv2 = _install_cv2()
    sys.modules['fastvision.detection.plot'] = types.ModuleType('fastvision.detection.plot')      # plotting helper (cv2 drawing)
    sys.modules['fastvision.detection.plot'].draw_box_label = lambda *a, **k: None
    from fastvision.datasets.detection_dataloader import BaseDataset
    from fastvision.datasets.common.padding import Padding
    gen = np.random.default_rng(2024)
    out = {}
    samples = []
    for i, (h, w, S, n) in enumerate(PIPE_CASES):
        rgb = synth_image(gen, h, w)
        cv2._images[f'img{i}'] = np.ascontiguousarray(rgb[:, :, ::-1])          # imread hands out BGR
        ann = synth_boxes(gen, n, h, w)
        out[f'p_{i}_rgb'] = rgb
        out[f'p_{i}_ann'] = np.array(ann, dtype=np.float64)
        samples.append((f'img{i}', ann, S))
    out['p_cases'] = np.array(PIPE_CASES)
    # ---- L1: BaseDataset.__getitem__ under every flip combination ------------------------------------------
    for i, (path, ann, S) in enumerate(samples):
        ds = BaseDataset([(path, ann)], S, max_det=200)
        for hf in (0, 1):
            for vf in (0, 1):
                ds.augmentation.augmentations[0].p = 2.0 if hf else -1.0      # random.random() <= p
                ds.augmentation.augmentations[1].p = 2.0 if vf else -1.0
                img, lab = ds[0]
                if S <= 128:
                    out[f'l1_{i}_{hf}{vf}_img'] = img.numpy()
                else:
                    out[f'l1_{i}_{hf}{vf}_img_sub'] = img.numpy()[:, ::7, ::5]
                    out[f'l1_{i}_{hf}{vf}_img_sums'] = np.array([img.double().sum().item(), (img.double() ** 2).sum().item()])
                out[f'l1_{i}_{hf}{vf}_lab'] = lab.numpy()
    # ---- L2: collate of a mixed batch (same input size) ------------------------------------------------------
    ds = BaseDataset([(p, a) for p, a, S in samples if S == 128], 128, max_det=200)
    for a in ds.augmentation.augmentations[:2]:
        a.p = -1.0
    imgs, labs = BaseDataset.collate_fn([ds[j] for j in range(len(ds))])
    out['l2_index'] = np.array([i for i, (_, _, S) in enumerate(samples) if S == 128])
    out['l2_labels'] = labs.numpy()
    out['l2_image_sums'] = imgs.double().sum(dim=(1, 2, 3)).numpy()
    # ---- L3: Padding offsets over a range of sizes (its round(x -/+ 0.1) rule) --------------------------------
    rows = []
    for rh in range(1, 41):
        for rw in (1, 2, 3, 17, 40):
            _, pos = Padding(np.zeros((rh, rw, 3), np.uint8), input_size=(40, 40), color=(114, 114, 114), align='center')
            rows.append((rh, rw) + tuple(pos))
    out['l3_padding'] = np.array(rows)
    np.savez_compressed(os.path.join(GOLD, 'pipeline.npz'), **out)
    print('pipeline fixtures:', len(out), 'arrays')


def gen_pipeline_demo():
    boot_demo()
    for n in ('albumentations', 'albumentations.pytorch'):
        sys.modules[n] = type(sys.modules['cv2'])(n)
    _install_cv2()
    import data_gen as DG
    gen = np.random.default_rng(777)
    out = {}
    imgs = []
    for i, (h, w) in enumerate(((120, 160), (125, 83), (53, 80), (160, 160))):
        rgb = synth_image(gen, h, w)
        ann = np.array(synth_boxes(gen, 3 + i, h, w), dtype=np.float32)
        out[f'q_{i}_rgb'] = rgb
        out[f'q_{i}_ann'] = ann
        imgs.append((rgb, ann[:, 1:].copy(), ann[:, 0].copy()))
    # ---- D1: ResizeByMax, flips, Padding (validation path) -----------------------------------------------------
    for i, (rgb, xyxy, cat) in enumerate(imgs):
        im, lb = DG.ResizeByMax(rgb.copy(), xyxy.copy(), 96)
        out[f'd1_{i}_resized'], out[f'd1_{i}_resized_lab'] = im, lb
        imh, lbh = DG.HorizontalFlip(im.copy(), lb.copy())
        imv, lbv = DG.VerticalFlip(imh.copy(), lbh.copy())
        out[f'd1_{i}_hv'], out[f'd1_{i}_hv_lab'] = imv, lbv
        pim, plb = DG.Padding(imv.copy(), lbv.copy(), 96, fill_value=128)
        out[f'd1_{i}_padded'], out[f'd1_{i}_padded_lab'] = pim, plb
    # ---- D2: Mosaic01 -------------------------------------------------------------------------------------------------
    for case, S in enumerate((96, 160)):
        pre = [DG.ResizeByMax(a.copy(), b.copy(), S) + (c.copy(),) for a, b, c in imgs]      # preprocess_image_label does this first
        m, xyxy, cat = DG.Mosaic01(pre, S, fill_value=128)
        out[f'd2_{case}_image'], out[f'd2_{case}_xyxy'], out[f'd2_{case}_cat'] = m, xyxy, cat
    np.savez_compressed(os.path.join(GOLD, 'pipeline_demo.npz'), **out)
    print('pipeline_demo fixtures:', len(out), 'arrays')


# ====================================================================================== two-stage head (scope row f-4)
def gen_rpn():
    """RPN proposal layer: the reference's own RPN class (demos/faster_rcnn/models/rpn.py) on random head outputs.  Captured:
    the inputs, the rows right after the clamp (by re-running its helper methods exactly as filter_proposals does) and the
    per-image proposals of filter_proposals itself (torchvision.ops.nms = the oracle's stand-in)."""
    import torch
    _stub_missing()
    _install_nms()
    sys.path.insert(0, os.path.join(REF, 'demos', 'faster_rcnn'))
    from models.rpn import RPN
    from utils.anchor_generator import get_base_anchor
    out = {}
    for case, (B, H, W, pre, post, scale) in enumerate([(2, 14, 14, 2000, 2000, 1.0), (1, 19, 25, 300, 50, 2.0), (3, 7, 9, 2000, 2000, 0.3)]):
        base = torch.from_numpy(get_base_anchor(scales=[128, 256, 512], ratios=[0.5, 1, 2]))
        rpn = RPN(training=False, base_anchors=base, backbone_stride=16, in_channels=8, rpn_pre_nms_top_n=pre, rpn_post_nms_top_n=post,
                  rpn_nms_thresh=0.7)
        g = torch.Generator().manual_seed(100 + case)
        A = base.size(0)
        cls = torch.randn(B, H, W, A, 2, generator=g) * 2
        d = torch.randn(B, H, W, A, 4, generator=g) * scale
        anchors = rpn.make_anchors_xywh(H, W, 'cpu')
        props = rpn.filter_proposals(cls, d, anchors, H, W)
        out[f'c{case}_shape'] = np.array([B, H, W, A, pre, post])
        out[f'c{case}_cls'], out[f'c{case}_d'] = cls.numpy(), d.numpy()
        out[f'c{case}_base_wh'] = rpn.base_anchors.numpy()
        out[f'c{case}_anchors'] = anchors.numpy()
        xyxy = rpn.xywh2xyxy(rpn.dxdydwdh2xywh(d.clone(), anchors))
        out[f'c{case}_xyxy_unclamped'] = xyxy.numpy()
        out[f'c{case}_score'] = torch.softmax(cls.clone(), dim=4)[..., 1].numpy()
        for b, p in enumerate(props):
            out[f'c{case}_prop{b}'] = p.numpy()
    np.savez_compressed(os.path.join(GOLD, 'rpn_proposals.npz'), **out)
    gen_rpn_match(RPN, get_base_anchor)
    print('rpn fixtures:', len(out), 'arrays', os.path.getsize(os.path.join(GOLD, 'rpn_proposals.npz')), 'bytes')


def gen_rpn_step(RPN, get_base_anchor):
    """One training forward + backward of the reference's RPN module on a random feature map: weights, inputs, the permutations
    torch.randperm produced (recorded, so that the path under test can use the same draws), the two losses and gradient
    statistics of every parameter and of the feature map."""
    import torch
    out = {}
    B, C, H, W, T = 2, 64, 12, 16, 7
    torch.manual_seed(77)
    base = torch.from_numpy(get_base_anchor(scales=[64, 128, 256], ratios=[0.5, 1, 2]))
    rpn = RPN(training=True, base_anchors=base, backbone_stride=16, in_channels=C, rpn_positives_per_image=16, rpn_negatives_per_image=48)
    g = torch.Generator().manual_seed(78)
    for p in rpn.parameters():                                 # livelier than the std = 0.01 init: losses that depend on every path
        p.data = torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.1)
    feature = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    tb = torch.sort(torch.cat([torch.arange(B), torch.randint(0, B, (T - B,), generator=g)]))[0].float()
    wh = torch.exp(np.log(0.15) + (np.log(0.7) - np.log(0.15)) * torch.rand(T, 2, generator=g))
    xy = wh / 2 + (1 - wh) * torch.rand(T, 2, generator=g)
    targets = torch.cat([tb[:, None], torch.zeros(T, 1), xy, wh], 1)
    perms, real_perm = [], torch.randperm
    pg = torch.Generator().manual_seed(79)

    def recorded(n, device=None):
        p = real_perm(n, generator=pg)
        perms.append(p.clone())
        return p
    torch.randperm = recorded
    try:
        proposals, loss_cls, loss_box = rpn(feature, targets)
    finally:
        torch.randperm = real_perm
    (loss_cls + loss_box).backward()
    out['shape'] = np.array([B, C, H, W, T])
    out['base_anchors_px'] = base.numpy()
    for k, v in rpn.state_dict().items():
        out['w_' + k] = v.numpy()
    out['feature'], out['targets'] = feature.detach().numpy(), targets.numpy()
    for i, p in enumerate(perms):
        out[f'perm{i}'] = p.numpy()
    out['loss_cls'], out['loss_box'] = loss_cls.detach().numpy(), loss_box.detach().numpy()
    out['grad_feature'] = feature.grad.numpy()
    for k, p in rpn.named_parameters():
        out['g_' + k] = p.grad.numpy()
    for b, p in enumerate(proposals):
        out[f'prop{b}'] = p.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, 'rpn_step.npz'), **out)
    print('rpn step fixtures:', len(out), 'arrays', os.path.getsize(os.path.join(GOLD, 'rpn_step.npz')), 'bytes; losses', float(loss_cls), float(loss_box))


def gen_rpn_match(RPN, get_base_anchor):
    """The labelling inside RPN.computet_loss is not returned by anything, so the reference is made to reveal it: the class
    logits fed to it carry (anchor index, image index) instead of scores, the regression carries the anchor index, torch.randperm
    is the identity and the per-image sample sizes are unbounded.  The tensors the reference then hands to its two loss
    functions list, per image, every negative anchor followed by every positive anchor (classification) and, for the positive
    ones, the regression targets computed from the box each was matched with."""
    import torch
    import torch.nn.functional as F
    import models.rpn as M
    out = {}
    for case, (B, H, W, T) in enumerate([(2, 14, 14, 9), (3, 19, 25, 31), (1, 7, 9, 2)]):
        base = torch.from_numpy(get_base_anchor(scales=[128, 256, 512], ratios=[0.5, 1, 2]))
        rpn = RPN(training=True, base_anchors=base, backbone_stride=16, in_channels=8, rpn_positives_per_image=10 ** 7,
                  rpn_negatives_per_image=10 ** 7)
        A = base.size(0)
        Na = H * W * A
        g = torch.Generator().manual_seed(500 + case)
        tb = torch.sort(torch.randint(0, B, (T,), generator=g))[0].float()
        tb[:B] = torch.arange(B).float()                      # every image has at least one box (the reference needs it)
        tb = torch.sort(tb)[0]
        wh = torch.exp(np.log(0.05) + (np.log(0.9) - np.log(0.05)) * torch.rand(T, 2, generator=g))
        xy = wh / 2 + (1 - wh) * torch.rand(T, 2, generator=g)
        targets = torch.cat([tb[:, None], torch.randint(0, 20, (T, 1), generator=g).float(), xy, wh], 1)
        if T > 4:                                             # a box identical to an anchor (IoU 1) and a tiny one (claims by rule 3 only)
            targets[2, 2:] = torch.tensor([5.0 / W, 4.0 / H, float(rpn.base_anchors[4, 0]) / W, float(rpn.base_anchors[4, 1]) / H])
            targets[3, 4:] = torch.tensor([0.004, 0.003])
        idx = torch.arange(Na, dtype=torch.float32).view(1, H, W, A, 1).expand(B, H, W, A, 1)
        img = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1, 1).expand(B, H, W, A, 1)
        cls = torch.cat([idx, img], 4).contiguous()
        d = idx.expand(B, H, W, A, 4).contiguous()
        anchors = rpn.make_anchors_xywh(H, W, 'cpu')
        seen = {}
        class Capture(torch.nn.Module):
            def forward(self, p, t):
                seen['cls'] = (p.clone(), t.clone())
                return p.sum() * 0
        rpn.focal_loss = Capture()
        real_sl1, real_perm = F.smooth_l1_loss, torch.randperm
        F.smooth_l1_loss = lambda p, t, reduction='mean': (seen.__setitem__('box', (p.clone(), t.clone())), p.sum() * 0)[1]
        torch.randperm = lambda n, device=None: torch.arange(n)
        try:
            rpn.computet_loss(cls, d, anchors, targets)
        finally:
            F.smooth_l1_loss, torch.randperm = real_sl1, real_perm
        pc, tc = seen['cls']
        pb, tbx = seen['box']
        out[f'm{case}_shape'] = np.array([B, H, W, A, T])
        out[f'm{case}_base_wh'] = rpn.base_anchors.numpy()
        out[f'm{case}_targets'] = targets.numpy()
        out[f'm{case}_anchor'] = pc[:, 0].long().numpy()           # sampled anchors: per image negatives then positives
        out[f'm{case}_image'] = pc[:, 1].long().numpy()
        out[f'm{case}_is_pos'] = tc.long().numpy()
        out[f'm{case}_pos_anchor'] = pb[:, 0].long().numpy()       # positives again, image after image
        out[f'm{case}_pos_dxdydwdh'] = tbx.numpy()                 # xywh2dxdydwdh(matched box, anchor)
    # Fast head: select_positive_negative_samples RETURNS its samples; with randperm = identity and unbounded sample sizes they
    # are every positive / negative proposal of every image (fast.py:100-166)
    from models.fast import Fast
    for case, (B, n, T, H, W) in enumerate([(2, 300, 9, 38, 50), (3, 80, 14, 14, 14)]):
        fast = Fast(training=True, module_after_roi=torch.nn.Identity(), in_channels=8, num_classes=20, fast_positives_per_image=10 ** 7,
                    fast_negatives_per_image=10 ** 7)
        g = torch.Generator().manual_seed(900 + case)
        tb = torch.sort(torch.cat([torch.arange(B), torch.randint(0, B, (T - B,), generator=g)]))[0].float()
        twh = torch.exp(np.log(0.08) + (np.log(0.8) - np.log(0.08)) * torch.rand(T, 2, generator=g))
        txy = twh / 2 + (1 - twh) * torch.rand(T, 2, generator=g)
        targets = torch.cat([tb[:, None], torch.randint(0, 20, (T, 1), generator=g).float(), txy, twh], 1) * torch.tensor([1, 1, W, H, W, H])
        proposals = []
        for b in range(B):
            mine = targets[targets[:, 0] == b][:, 2:]
            near = mine[torch.randint(0, mine.size(0), (n // 2,), generator=g)] * (1 + 0.35 * (torch.rand(n // 2, 4, generator=g) - 0.5))
            wh = torch.exp(np.log(1.0) + (np.log(30.0) - np.log(1.0)) * torch.rand(n - n // 2, 2, generator=g))
            far = torch.cat([torch.rand(n - n // 2, 2, generator=g) * torch.tensor([W, H]), wh], 1)
            p = torch.cat([near, far], 0)[torch.randperm(n, generator=g)]
            if b == 0:
                p[0] = mine[0]                                   # IoU exactly 1
            proposals.append(p)
        real_perm = torch.randperm
        torch.randperm = lambda k, device=None: torch.arange(k)
        try:
            pos, neg = fast.select_positive_negative_samples([p.clone() for p in proposals], targets.clone(), 'cpu')
        finally:
            torch.randperm = real_perm
        out[f'f{case}_shape'] = np.array([B, n, T, H, W])
        out[f'f{case}_targets'] = targets.numpy()
        for b in range(B):
            out[f'f{case}_prop{b}'] = proposals[b].numpy()
        out[f'f{case}_pos'], out[f'f{case}_neg'] = pos.numpy(), neg.numpy()
    np.savez_compressed(os.path.join(GOLD, 'rpn_match.npz'), **out)
    gen_rpn_step(RPN, get_base_anchor)
    print('rpn match fixtures:', len(out), 'arrays', os.path.getsize(os.path.join(GOLD, 'rpn_match.npz')), 'bytes')


def gen_faster():
    """One training forward + backward of the reference's whole Faster_Rcnn model (VGG16 + RPN + Fast head) on the CPU.  Weights are
    NOT stored (the VGG classifier alone is 103 M values): the model is built right after torch.manual_seed(SEED), so the mirror,
    which creates the same modules in the same order, starts from the same values -- per-parameter checksums are stored to prove
    it.  torchvision is absent: nms = the oracle's stand-in, roi_align = oracle.roi_align wrapped as an autograd function;
    Dropout is set to p = 0 (its masks are not part of the path); torch.randperm draws are recorded."""
    import torch
    _stub_missing()
    _install_nms()
    from oracle import roi_align as RA

    class RoiAlignStandIn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, feat, boxes, ph, pw):
            ctx.boxes, ctx.shape = boxes.detach().numpy().copy(), tuple(feat.shape)
            return torch.from_numpy(RA.roi_align(feat.detach().numpy(), ctx.boxes, (ph, pw)))

        @staticmethod
        def backward(ctx, g):
            return torch.from_numpy(RA.roi_align_backward(g.numpy(), ctx.boxes, ctx.shape)), None, None, None
    sys.modules['torchvision.ops'].roi_align = lambda feat, boxes, output_size: RoiAlignStandIn.apply(feat, boxes, output_size[0], output_size[1])
    sys.path.insert(0, os.path.join(REF, 'demos', 'faster_rcnn'))
    from models.faster import Faster_Rcnn
    from utils.anchor_generator import get_base_anchor
    SEED, B, H, W, T, NC = 4321, 2, 96, 128, 6, 20
    base = torch.from_numpy(get_base_anchor(scales=[32, 64, 128], ratios=[0.5, 1, 2]))
    torch.manual_seed(SEED)
    model = Faster_Rcnn(training=True, num_classes=NC, base_anchors=base, rpn_positives_per_image=16, rpn_negatives_per_image=48,
                        fast_positives_per_image=8, fast_negatives_per_image=24)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    # the default initialisation shrinks activations by ~0.6x per layer (1e-3 after 13 layers: the heads would see zeros and the
    # losses would not depend on the backbone): scale the backbone's filters so that the feature map stays O(1)
    CONV_SCALE = 1.7
    for m in model.backbone.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data *= CONV_SCALE
    g = torch.Generator().manual_seed(SEED + 1)
    images = torch.rand(B, 3, H, W, generator=g)
    tb = torch.sort(torch.cat([torch.arange(B), torch.randint(0, B, (T - B,), generator=g)]))[0].float()
    wh = torch.exp(np.log(0.2) + (np.log(0.7) - np.log(0.2)) * torch.rand(T, 2, generator=g))
    xy = wh / 2 + (1 - wh) * torch.rand(T, 2, generator=g)
    targets = torch.cat([tb[:, None], torch.randint(0, NC, (T, 1), generator=g).float(), xy, wh], 1)
    perms, real_perm = [], torch.randperm
    pg = torch.Generator().manual_seed(SEED + 2)

    def recorded(n, device=None):
        p = real_perm(n, generator=pg)
        perms.append(p.clone())
        return p
    torch.randperm = recorded
    try:
        proposals, l_rc, l_rb, l_fc, l_fb = model(images, targets.clone())
    finally:
        torch.randperm = real_perm
    (l_rc + l_rb + l_fc + l_fb).backward()
    out = {'meta': np.array([SEED, B, H, W, T, NC]), 'conv_scale': np.array([CONV_SCALE]), 'base_anchors_px': base.numpy(), 'images': images.numpy(), 'targets': targets.numpy(),
           'losses': np.array([float(l_rc), float(l_rb), float(l_fc), float(l_fb)])}
    for i, p in enumerate(perms):
        out[f'perm{i}'] = p.numpy()
    names = []
    for k, p in model.named_parameters():
        names.append(k)
        out['wsum_' + k] = np.array([p.detach().double().sum().item(), p.detach().double().abs().sum().item()])
        gr = p.grad.double()
        out['gstat_' + k] = np.array([gr.sum().item(), gr.abs().sum().item(), gr.norm().item()] + gr.flatten()[:3].tolist())
    out['param_names'] = np.array(names)
    for b, p in enumerate(proposals):
        out[f'nprop{b}'] = np.array([p.size(0)])
    np.savez_compressed(os.path.join(GOLD, 'faster_step.npz'), **out)
    print('faster fixtures:', len(out), 'arrays', os.path.getsize(os.path.join(GOLD, 'faster_step.npz')), 'bytes; losses', out['losses'], 'perms', [len(p) for p in perms])


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    os.makedirs(GOLD, exist_ok=True)
    if which == 'lib':
        gen_lib()
    elif which == 'lib_curves':
        gen_lib_curves()
    elif which == 'demo':
        gen_demo()
    elif which == 'eval_lib':
        gen_eval_lib()
    elif which == 'eval_demo':
        gen_eval_demo()
    elif which == 'pipeline':
        gen_pipeline()
    elif which == 'pipeline_demo':
        gen_pipeline_demo()
    elif which == 'rpn':
        gen_rpn()
    elif which == 'faster':
        gen_faster()
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        for s in ('lib', 'demo', 'eval_lib', 'eval_demo', 'pipeline', 'pipeline_demo', 'rpn', 'faster'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), s], env=env)
