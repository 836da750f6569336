"""CPU-side tests: the C-ABI library loads and exports every symbol include/fastvision_amd.h declares, the host
modules mirror the reference's state_dict keys / seeded init (checked against the pinned oracle), the product
path refuses CPU tensors loudly, and the synthetic batch generator is deterministic.  No compute calls here."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as ge
    ge.build()
    from fastvision_amd import _lib
    return _lib


def test_library_exports_every_header_symbol(built):
    hdr = open(os.path.join(ROOT, 'include', 'fastvision_amd.h')).read()
    declared = set(re.findall(r'\b(fva_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'fva_status', 'fva_dtype'}
    lib = built.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in the header but not exported'
    assert declared == set(built.PROTOTYPES), declared ^ set(built.PROTOTYPES)
    assert lib.fva_version() >= 1


def test_argument_errors_are_reported_not_crashed(built):
    import ctypes as C
    lib = built.load()
    d = built.ConvDesc(1, 2, 8, 8, 48, 64, 3, 1, 1, 1)        # Cin=48 is not a multiple of 64 (bf16 k-tile)
    rc = lib.fva_conv_fwd(C.byref(d), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None, None)
    assert rc == -1 and b'reduction channels' in lib.fva_last_error()
    d = built.ConvDesc(0, 2, 8, 8, 64, 64, 5, 1, 2, 1)
    assert lib.fva_conv_fwd(C.byref(d), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None, None) == -1
    with pytest.raises(RuntimeError):
        built.call('fva_adam_step', None, None, 0, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None)
    # the accumulator forms: a replica count that is no power of two (or beyond FVA_BN_ACC_MAX_REPLICAS), a consumer asked to zero the accumulator it reads
    d = built.ConvDesc(1, 2, 8, 8, 64, 64, 3, 1, 1, 1)
    for bad in (0, 3, 64):
        assert lib.fva_conv_fwd_acc(C.byref(d), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), bad, None) == -1
        assert b'replicas' in lib.fva_last_error()
    fin = built.BnFwdAcc(64, 64, 1, 16, 16, None, None, None, 0.1, 1e-5, 16, 16, 16, 16)
    assert lib.fva_bn_silu_apply_acc(1, C.c_void_p(16), C.byref(fin), None, 0, C.c_void_p(16), 1, 2, 8, 8, 64, None) == -1
    assert b'another accumulator' in lib.fva_last_error()


def test_modules_mirror_reference_keys_and_seeded_init():
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.demos.yolov3_u.models import YoloV3
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.synthetic import coco_anchors_feature, coco_anchors_px
    from oracle import train as otrain
    torch.manual_seed(20220504)
    net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                 num_anchors_per_level=[3, 3, 3], training=True)
    ref, _ = otrain.make_library(20220504)
    sd, rsd = net.state_dict(), ref.state_dict()
    assert list(sd) == list(rsd) and len(sd) == 438
    assert all(torch.equal(sd[k], rsd[k]) for k in sd)
    assert net.backbone_strides_per_level == [32, 16, 8] and net.backbone_channels_per_level == [1024, 512, 256]
    assert [tuple(a.shape) for a in net.anchors_per_level] == [(3, 1, 1, 2)] * 3
    torch.manual_seed(20220504)
    demo = YoloV3(anchors=coco_anchors_feature())
    dref, _ = otrain.make_demo(20220504)
    sd, rsd = demo.state_dict(), dref.state_dict()
    assert list(sd) == list(rsd) and all(torch.equal(sd[k], rsd[k]) for k in sd)
    assert list(sd)[:312] == [k for k in net.state_dict() if k.startswith('backbone.')]      # shared backbone keys


def test_product_path_refuses_cpu_tensors():
    from fastvision_amd.classfication.models.darknet53 import ConvBlock3x3
    from fastvision_amd.detection.tools import xyxy_iou
    from fastvision_amd.loss import Yolov3Loss
    with pytest.raises(RuntimeError, match='no CPU path'):
        ConvBlock3x3(32, 64)(torch.zeros(1, 32, 8, 8))
    with pytest.raises(RuntimeError, match='no CPU path'):
        xyxy_iou(torch.zeros(2, 4), torch.zeros(2, 4))

    class M:
        anchors_per_level = [torch.ones(3, 1, 1, 2)] * 3
        backbone_strides_per_level = [32, 16, 8]
    with pytest.raises(RuntimeError, match='no CPU path'):
        Yolov3Loss(M(), 0.5, 0.05, 1.0, 0.5)([torch.zeros(1, 3, 2, 2, 85)] * 3, torch.zeros(1, 6))


def test_missing_library_fails_loudly(monkeypatch):
    from fastvision_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libfastvision_amd.so')
    with pytest.raises(RuntimeError, match='REQUIRED'):
        _lib.load()


def test_synthetic_batch_is_deterministic_and_well_formed():
    from fastvision_amd.synthetic import synthetic_batch
    a_img, a_t = synthetic_batch(4, 64)
    b_img, b_t = synthetic_batch(4, 64)
    assert torch.equal(a_img, b_img) and torch.equal(a_t, b_t)
    assert a_t.shape[1] == 6 and (a_t[:, 0].diff() >= 0).all()
    assert set(a_t[:, 0].long().tolist()) == {0, 1, 2, 3}             # every image has >= 1 box
    assert (a_t[:, 2:4] < 1).all() and (a_t[:, 2:] > 0).all()
    c_img, _ = synthetic_batch(4, 64, rank=1)
    assert not torch.equal(a_img, c_img)


def test_halo_view_detection():
    from fastvision_amd import ops
    buf, view = ops.halo_alloc(2, 8, 5, 6, torch.float32, 'cpu', pad=1)
    info = ops.halo_info(view, torch.float32)
    assert info is not None and info[1] == 1 and info[0] == buf.data_ptr()
    dense = torch.zeros(2, 5, 6, 8).permute(0, 3, 1, 2)
    assert ops.halo_info(dense, torch.float32)[1] == 0
    assert ops.halo_info(torch.zeros(2, 8, 5, 6), torch.float32) is None      # plain NCHW is foreign
    assert ops.halo_info(view, torch.bfloat16) is None


def test_lr_schedules_match_reference_values():
    """utils/sheduler.py mirror vs values captured from the reference's own functions (tests/golden/sched.npz)."""
    import numpy as np
    from fastvision_amd.utils import sheduler as S
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'sched.npz'))

    def curve(make, steps=12):
        p = [torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD(p, lr=1.0)
        sch = make(opt)
        lrs = []
        for _ in range(steps):
            lrs.append(opt.param_groups[0]['lr'])
            opt.step()
            sch.step()
        return np.array(lrs)

    def wc(o):
        for g in o.param_groups:
            g['lr'] = 0.01
        return S.WarmupCosineLR(o, milestones=[8, 14], min_ratio=0.1, cycle_decay=0.5, warmup_iters=4, warmup_factor=0.1)
    np.testing.assert_allclose(curve(lambda o: S.CosineLR(o, 10, 0.01, 0.0001)), gold['cosine'], rtol=1e-12)
    np.testing.assert_allclose(curve(lambda o: S.LinearLR(o, 10, 0.01, 0.0001)), gold['linear'], rtol=1e-12)
    np.testing.assert_allclose(curve(lambda o: S.ExponentialLR(o, 10, 0.01, 0.0001)), gold['exp'], rtol=1e-12)
    np.testing.assert_allclose(curve(wc, 20), gold['warmcos'], rtol=1e-12)
    with pytest.raises(ValueError):
        S.WarmupCosineLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0), milestones=[5, 3])


def test_checkpoint_helpers_roundtrip(tmp_path):
    from fastvision_amd.classfication.models.darknet53 import ConvBlock3x3
    from fastvision_amd.utils import LoadFromParrel, LoadStatedict, SaveModel, SqueezeModel
    torch.manual_seed(1)
    a, b, c = ConvBlock3x3(32, 64), ConvBlock3x3(32, 64), ConvBlock3x3(32, 64)
    path = str(tmp_path / 'last.pth')
    SaveModel({'model': a, 'optimizer': {}}, path, weights_only=True)
    blob = torch.load(path)
    assert set(blob) == {'model', 'optimizer', 'date'} and list(blob['model']) == list(a.state_dict())
    LoadStatedict(b, path, 'cpu')
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    torch.save({'module.' + k: v for k, v in a.state_dict().items()}, path)        # a DataParallel checkpoint
    LoadFromParrel(c, path, 'cpu')
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())
    SqueezeModel(c, ['bn'], False)
    assert not c.bn.weight.requires_grad and c.conv.weight.requires_grad


def test_fit_run_epoches_contract(tmp_path, monkeypatch):
    """utils/fit.py: per batch model -> zero_grad -> loss -> backward -> step; scheduler.step() once per epoch; a
    'last.pth' checkpoint {'model': state_dict, 'optimizer': ..., 'date': ...} after every epoch (fit.py:29-71)."""
    import torch
    from fastvision_amd.utils import Fit
    monkeypatch.chdir(tmp_path)
    events = []

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))

        def forward(self, x, val=False):
            events.append('forward')
            return (x * self.w).sum()

    class Opt(torch.optim.SGD):
        def zero_grad(self, *a, **k):
            events.append('zero_grad')
            return super().zero_grad(*a, **k)

        def step(self, *a, **k):
            events.append('step')
            return super().step(*a, **k)

    class Sched:
        def step(self):
            events.append('sched')

    net = Net()
    opt = Opt(net.parameters(), lr=0.1)

    def loss(pred, labels):
        events.append('loss')
        return pred + labels.sum()
    loader = [(torch.ones(3), torch.zeros(1)), (torch.full((3,), 2.0), torch.zeros(1))]
    fit = Fit(net, torch.device('cpu'), opt, Sched(), loss, end_epoch=2, train_loader=loader)
    fit.run_epoches()
    per_batch = ['forward', 'zero_grad', 'loss', 'step']
    assert events == (per_batch * 2 + ['sched']) * 2
    assert len(fit.history) == 2 and len(fit.history[0]) == 2
    ckpt = torch.load(tmp_path / 'last.pth', weights_only=False)
    assert set(ckpt) == {'model', 'optimizer', 'date'} and 'w' in ckpt['model']
    assert torch.allclose(ckpt['model']['w'], net.w.detach())


def test_config1_resnet18_cpu_fit_plumbing(tmp_path, monkeypatch):
    """BASELINE config 1: ResNet-18 classification, 32x3x224x224 synthetic, through utils/fit.py on the CPU -- plumbing only
    (stock torch ops: the classification families are out of scope, the Fit loop is the caller of the accelerated path).
    The reference drives `Fit._train` with a list-of-tuples loader, CrossEntropyLoss and SGD the same way."""
    import torch
    import torch.nn as nn
    from fastvision_amd.utils import Fit
    from fastvision_amd.utils.sheduler import LinearLR
    monkeypatch.chdir(tmp_path)

    class Block(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.c1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.b1 = nn.BatchNorm2d(cout)
            self.c2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.b2 = nn.BatchNorm2d(cout)
            self.down = None
            if stride != 1 or cin != cout:
                self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

        def forward(self, x):
            y = self.b2(self.c2(torch.relu(self.b1(self.c1(x)))))
            return torch.relu(y + (x if self.down is None else self.down(x)))

    def resnet18(classes):
        layers, cin = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)], 64
        for cout, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            layers += [Block(cin, cout, stride), Block(cout, cout, 1)]
            cin = cout
        return nn.Sequential(*layers, nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(512, classes))

    torch.manual_seed(0)
    net = resnet18(10)
    assert sum(p.numel() for p in net.parameters()) == 11_181_642            # ResNet-18 with a 10-way head
    g = torch.Generator().manual_seed(1)
    batch = (torch.rand(32, 3, 224, 224, generator=g), torch.randint(0, 10, (32,), generator=g))
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    sched = LinearLR(opt, steps=2, initial_lr=0.05, last_lr=0.005)
    fit = Fit(net, torch.device('cpu'), opt, sched, nn.CrossEntropyLoss(), end_epoch=2, train_loader=[batch, batch])
    fit.run_epoches()
    losses = [l for epoch in fit.history for l in epoch]
    assert len(losses) == 4 and all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]                                             # the same batch four times: it must fit it
    assert (tmp_path / 'last.pth').exists()


def test_faster_rcnn_fit_loop_contract(tmp_path, monkeypatch):
    """demos/faster_rcnn/cfg/_fit.py mirror: four losses summed, backward, global-norm clipping at 10, step; lr / 10 at every
    ninth epoch; a state_dict checkpoint per epoch (reference _fit.py:6-50)."""
    import types
    import torch
    from fastvision_amd.demos.faster_rcnn.cfg import _fit
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([3.0, 4.0]))

        def forward(self, images, targets):
            l = (self.w * images).sum()
            return None, l * 100, l * 0, l * 0, targets.sum() * self.w.sum() * 0
    net = Net()
    net.w.grad = torch.tensor([30.0, 40.0])
    assert abs(_fit.clip_gradient(net, 10.) - 50.0) < 1e-4 and torch.allclose(net.w.grad, torch.tensor([6.0, 8.0]))
    net.w.grad = torch.tensor([0.3, 0.4])
    _fit.clip_gradient(net, 10.)
    assert torch.allclose(net.w.grad, torch.tensor([0.3, 0.4]))                      # below the bound: untouched
    opt = torch.optim.SGD(net.parameters(), lr=1.0)
    logged = []
    _fit._Train(net, [(torch.ones(2), torch.zeros(1))], opt, log=lambda *v: logged.append(v))
    assert torch.allclose(net.w.detach(), torch.tensor([3.0, 4.0]) - 10 * torch.tensor([1.0, 1.0]) / 2 ** 0.5, atol=1e-5)   # gradient (100, 100) clipped to norm 10
    assert len(logged) == 1 and len(logged[0]) == 5 and abs(logged[0][0] - 700.0) < 1e-3
    args = types.SimpleNamespace(start_epoch=7, total_epoch=9)
    _fit.Fit(net, args, opt, [(torch.ones(2), torch.zeros(1))])
    assert abs(opt.param_groups[0]['lr'] - 0.1) < 1e-12                               # epoch 9: divided by ten
    assert (tmp_path / '8.pth').exists() and (tmp_path / '9.pth').exists()
    assert set(torch.load(tmp_path / '9.pth')) == {'w'}


def test_rpn_sample_refuses_a_permutation_that_does_not_fit():
    """Round 2 hit a GPU exception (HSA 0x1016 inside ATen's index kernel, gpurun_out/r2f/cfg5.log) when a sampling permutation
    drawn for the oracle's candidate list indexed past the HIP path's shorter list (demos/faster_rcnn/models/rpn.py:279-290 draws
    its own torch.randperm): rpn_sample range-checks a given permutation on the host, before any indexing kernel runs."""
    import pytest
    import torch
    from fastvision_amd import rpn_ops
    lab = torch.tensor([0, -1, -1, 1, -2, -1, 0])                   # 3 positives (>= 0), 3 negatives (-1), 1 ignored
    pos, neg = rpn_ops.rpn_sample(lab, 2, 2, torch.tensor([1, 0, 2]), torch.tensor([2, 1, 0]))
    assert pos.tolist() == [3, 0] and neg.tolist() == [5, 2]
    with pytest.raises(ValueError, match='positive permutation'):
        rpn_ops.rpn_sample(lab, 2, 2, torch.tensor([1]), torch.tensor([2, 1, 0]))             # too short for the 2 to draw
    with pytest.raises(ValueError, match='negative permutation'):
        rpn_ops.rpn_sample(lab, 2, 2, torch.tensor([1, 0, 2]), torch.tensor([3, 1, 0]))       # 3 indexes past the 3 candidates
    with pytest.raises(ValueError, match='positive permutation'):
        rpn_ops.rpn_sample(lab, 2, 2, torch.tensor([-1, 0, 2]), torch.tensor([2, 1, 0]))      # negative index


def test_accumulator_bookkeeping_of_the_batchnorm_statistics():
    """Host side of the accumulator form of the BatchNorm statistics (ops._AccState): one replica per 65536 output pixels (a power of two,
    at most 32 = FVA_BN_ACC_MAX_REPLICAS), five words per channel and replica; a producer that finds its accumulator not zero clears it
    first, each direction's consumer leaves the OTHER direction marked zero (it zeroed it on the device) and its own marked consumed."""
    from fastvision_amd import ops
    assert [ops._replicas(m) for m in (1, 65536, 65537, 204800, 819200, 3276800, 10 ** 9)] == [1, 1, 2, 4, 16, 32, 32]
    st = ops._AccState(64, 819200, torch.device('cpu'))
    assert st.replicas == 16 and tuple(st.buf.shape) == (2, 16 * 5 * 64) and st.buf.dtype == torch.int64 and not st.buf.any()
    st.produce(0)                                   # forward producer: clean -> produced, nothing to clear
    assert st.state == [1, 0]
    st.buf[0].fill_(7)                              # (what the tiles add)
    st.buf[1].fill_(9)                              # (backward sums of the previous step)
    st.consumed(0)                                  # the forward consumer zeroes the backward accumulator ON THE DEVICE; the host only notes it
    assert st.state == [2, 0]
    st.produce(1)                                   # backward producer: marked zero -> no clearing launch
    assert st.state == [2, 1] and st.buf[1].eq(9).all()
    st.consumed(1)
    assert st.state == [0, 2]
    st.produce(1)                                   # a second backward without a forward in between: the stale sums are cleared first
    assert st.state == [0, 1] and not st.buf[1].any()
    # the header's constants
    import re
    hdr = open(os.path.join(ROOT, 'include', 'fastvision_amd.h')).read()
    assert int(re.search(r'#define FVA_BN_ACC_WORDS (\d+)', hdr).group(1)) == 5
    assert int(re.search(r'#define FVA_BN_ACC_MAX_REPLICAS (\d+)', hdr).group(1)) == 32
