"""The training step captured in a HIP graph (graphs.GraphedTrainStep; the loop it replaces: utils/fit.py:52-66) against the same
step issued eagerly: identical kernels in identical order on identical data, so losses, parameters, BatchNorm buffers and Adam
moments must agree BIT FOR BIT after several steps -- including batches with fewer targets than the captured capacity (zero rows
match no anchor) and a learning rate changed between replays (device-resident LR scalar)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def lib_model(seed=20220504):
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.synthetic import coco_anchors_px
    torch.manual_seed(seed)
    m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
               in_channels=3, num_classes=80, training=True)
    return m.to(DEV).train()


def make(capturable=True):
    from fastvision_amd import FusedAdam
    from fastvision_amd.loss import Yolov3Loss
    net = lib_model()
    crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4, capturable=capturable)
    return net, crit, opt


def eager_step(net, crit, opt, images, tg):
    pred = net(images)
    opt.zero_grad()
    loss = crit(pred, tg)
    loss.backward()
    opt.step()
    return loss.detach().clone()


def state_of(net, opt):
    out = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for i, p in enumerate(net.parameters()):
        out[f'm{i}'] = opt.state[p]['exp_avg'].clone()
        out[f'v{i}'] = opt.state[p]['exp_avg_sq'].clone()
    return out


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_graphed_step_is_bit_identical_with_eager(dtype):
    import fastvision_amd
    from fastvision_amd.graphs import GraphedTrainStep
    from fastvision_amd.synthetic import synthetic_batch
    batches = [synthetic_batch(2, 128, seed=s) for s in (1234, 7, 99)]
    cap = max(t.shape[0] for _, t in batches) + 5
    lrs = [1e-4, 1e-4, 3e-5, 3e-5]
    order = [0, 1, 2, 0]
    with fastvision_amd.compute_dtype(dtype):
        net, crit, opt = make()
        want = []
        for i, lr in zip(order, lrs):
            opt.param_groups[0]['lr'] = lr
            im, tg = batches[i]
            want.append(eager_step(net, crit, opt, im.to(DEV), tg.to(DEV)))
        want_state = state_of(net, opt)
        torch.cuda.synchronize()

        net2, crit2, opt2 = make()
        im0, tg0 = batches[0]
        step = GraphedTrainStep(net2, lambda p, t: crit2(p, t), opt2, im0.to(DEV), tg0.to(DEV), max_targets=cap)
        got = []
        for i, lr in zip(order, lrs):
            opt2.param_groups[0]['lr'] = lr
            im, tg = batches[i]
            got.append(step(im.to(DEV), tg.to(DEV)).clone())
        got_state = state_of(net2, opt2)
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b), (a, b)
    for k in want_state:
        assert torch.equal(got_state[k], want_state[k]), k
    assert opt2._step_of(next(iter(net2.parameters()))) == 4
    assert float(opt2._dev[0]['state'][0]) == 4.0
    # the eval path after replays must see the updated parameters (packed-weight cache invalidated)
    net.eval(); net2.eval()
    with torch.no_grad(), fastvision_amd.compute_dtype(dtype):
        a = net(batches[1][0].to(DEV), val=True)[1]
        b = net2(batches[1][0].to(DEV), val=True)[1]
    assert torch.equal(a, b)


def test_two_graphs_on_one_optimizer_alternate_like_eager():
    """Two captures on ONE FusedAdam (a second batch size -- the tail batch of an epoch, utils/fit.py:52-66 iterates a loader whose
    last batch is smaller): each graph owns the pinned words its pointer-table upload re-reads, so alternating replays must equal
    the same steps issued eagerly, bit for bit.  (Round 2 kept one staging buffer per optimizer: the second capture rewrote the
    gradient pointers the first graph's Adam launch reads -- ADVICE.)"""
    import fastvision_amd
    from fastvision_amd.graphs import GraphedTrainStep
    from fastvision_amd.synthetic import synthetic_batch
    big, small = synthetic_batch(3, 128, seed=11), synthetic_batch(2, 128, seed=12)
    order = [big, small, big, small, small, big]
    with fastvision_amd.compute_dtype(torch.float32):
        net, crit, opt = make()
        want = [eager_step(net, crit, opt, im.to(DEV), tg.to(DEV)) for im, tg in order]
        want_state = state_of(net, opt)
        torch.cuda.synchronize()
        net2, crit2, opt2 = make()
        g_big = GraphedTrainStep(net2, lambda p, t: crit2(p, t), opt2, big[0].to(DEV), big[1].to(DEV))
        g_small = GraphedTrainStep(net2, lambda p, t: crit2(p, t), opt2, small[0].to(DEV), small[1].to(DEV))
        assert g_big._adam_staging[0][0].data_ptr() != g_small._adam_staging[0][0].data_ptr()
        got = [(g_big if b is big else g_small)(b[0].to(DEV), b[1].to(DEV)).clone() for b in order]
        got_state = state_of(net2, opt2)
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b), (a, b)
    for k in want_state:
        assert torch.equal(got_state[k], want_state[k]), k


def test_device_resident_adam_matches_host_scalar_adam():
    """fva_adam_step_dev (step count / LR in device memory) against fva_adam_step (host scalars): the bias corrections are computed
    by pow() on the device instead of the host's libm, so allow one ulp of the step size (1e-6 relative on the update)."""
    import fastvision_amd
    from fastvision_amd.synthetic import synthetic_batch
    im, tg = synthetic_batch(2, 64)
    im, tg = im.to(DEV), tg.to(DEV)
    res = []
    with fastvision_amd.compute_dtype(torch.float32):
        for cap in (False, True):
            net, crit, opt = make(capturable=cap)
            p0 = [p.detach().clone() for p in net.parameters()]
            for _ in range(3):
                eager_step(net, crit, opt, im, tg)
            res.append([(p.detach() - q) for p, q in zip(net.parameters(), p0)])
    worst = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item() for a, b in zip(*res))
    print('largest relative difference of the 3-step parameter update', worst)
    assert worst < 1e-5


def test_graphed_step_refuses_what_it_cannot_capture():
    from fastvision_amd import FusedAdam
    from fastvision_amd.graphs import GraphedTrainStep
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.synthetic import synthetic_batch
    net = lib_model()
    crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
    im, tg = synthetic_batch(2, 64)
    with pytest.raises(RuntimeError):
        GraphedTrainStep(net, crit, FusedAdam(net.parameters(), lr=1e-4), im.to(DEV), tg.to(DEV))     # host-scalar optimizer
    opt = FusedAdam(net.parameters(), lr=1e-4, capturable=True)
    step = GraphedTrainStep(net, crit, opt, im.to(DEV), tg.to(DEV), max_targets=tg.shape[0])
    with pytest.raises(ValueError):
        step(im.to(DEV), torch.cat([tg, tg]).to(DEV))                                                 # more targets than captured
    # a two-branch graph with fewer than four hardware queues segfaults in hipGraphLaunch (diagnosed: tools/hwq_capture_check.py):
    # refused before anything is captured
    import os
    old = os.environ.get('GPU_MAX_HW_QUEUES')
    os.environ['GPU_MAX_HW_QUEUES'] = '2'
    try:
        with pytest.raises(RuntimeError, match='GPU_MAX_HW_QUEUES'):
            GraphedTrainStep(net, crit, opt, im.to(DEV), tg.to(DEV), side_stream=True)
    finally:
        if old is None:
            del os.environ['GPU_MAX_HW_QUEUES']
        else:
            os.environ['GPU_MAX_HW_QUEUES'] = old


def test_graphed_demo_step_is_bit_identical_with_eager():
    """The demo surface (YoloV3 + ComputeLoss, demos/yolov3_u/utils/fit.py:52-66): ComputeLoss assigns every target row, so the step
    is captured at its exact target count; the anchors it reads from the model are cached on the host before capture."""
    import fastvision_amd
    from fastvision_amd import FusedAdam
    from fastvision_amd.demos.yolov3_u.models import YoloV3
    from fastvision_amd.demos.yolov3_u.utils import ComputeLoss
    from fastvision_amd.graphs import GraphedTrainStep
    from fastvision_amd.synthetic import coco_anchors_feature, synthetic_batch
    im, tg = synthetic_batch(2, 128, seed=5)
    im, tg = im.to(DEV), tg.to(DEV)
    im2 = torch.flip(im, dims=[3]).contiguous()

    def make_demo():
        torch.manual_seed(3)
        net = YoloV3(anchors=tuple(a.to(DEV) for a in coco_anchors_feature())).to(DEV).train()
        cl = ComputeLoss()
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4, capturable=True)
        return net, (lambda p, t: cl(p, t, net)), opt
    with fastvision_amd.compute_dtype(torch.bfloat16):
        net, loss_fn, opt = make_demo()
        want = []
        for x in (im, im2, im):
            pred = net(x)
            opt.zero_grad()
            loss = loss_fn(pred, tg)
            loss.backward()
            opt.step()
            want.append(loss.detach().clone())
        net2, loss_fn2, opt2 = make_demo()
        step = GraphedTrainStep(net2, loss_fn2, opt2, im, tg)
        got = [step(x, tg).clone() for x in (im, im2, im)]
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b), (a, b)
    for (k, a), (_, b) in zip(net2.state_dict().items(), net.state_dict().items()):
        assert torch.equal(a, b), k


def test_replays_eager_steps_and_training_mode_forwards_interleave_like_eager():
    """Round 4: the BatchNorm statistics live in per-layer accumulators whose zeroing is a protocol between launches (each direction's
    consumer zeroes the other direction's accumulator; the host clears one only in front of a producer that finds it dirty).  A captured
    graph replays the launches of a NORMAL step, so the protocol must also hold across the seams: replay -> eager step -> a training-mode
    forward without backward (leaves forward sums behind) -> replay -> eager step.  The mixed sequence must equal the same steps issued
    eagerly throughout, bit for bit."""
    import fastvision_amd
    from fastvision_amd.graphs import GraphedTrainStep
    from fastvision_amd.synthetic import synthetic_batch
    batches = [synthetic_batch(2, 128, seed=s) for s in (5, 6, 7)]
    cap = max(t.shape[0] for _, t in batches) + 5
    seq = ['g', 'e', 'f', 'g', 'e', 'g']             # g: graph replay, e: eager step, f: training-mode forward only (under no_grad)
    pick = [0, 1, 2, 1, 0, 2]
    with fastvision_amd.compute_dtype(torch.bfloat16):
        net, crit, opt = make()
        want = []
        for kind, i in zip(seq, pick):
            im, tg = batches[i]
            if kind == 'f':
                with torch.no_grad():
                    net(im.to(DEV))                  # updates the running statistics, as nn.BatchNorm2d in training mode does
            else:
                want.append(eager_step(net, crit, opt, im.to(DEV), tg.to(DEV)))
        want_state = state_of(net, opt)
        torch.cuda.synchronize()

        net2, crit2, opt2 = make()
        step = GraphedTrainStep(net2, lambda p, t: crit2(p, t), opt2, batches[0][0].to(DEV), batches[0][1].to(DEV), max_targets=cap)
        got = []
        for kind, i in zip(seq, pick):
            im, tg = batches[i]
            if kind == 'g':
                got.append(step(im.to(DEV), tg.to(DEV)).clone())
            elif kind == 'e':
                got.append(eager_step(net2, crit2, opt2, im.to(DEV), tg.to(DEV)))
            else:
                with torch.no_grad():
                    net2(im.to(DEV))
        got_state = state_of(net2, opt2)
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b), (a, b)
    for k in want_state:
        assert torch.equal(got_state[k], want_state[k]), k
