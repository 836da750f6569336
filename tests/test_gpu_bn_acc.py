"""BatchNorm statistics in fixed-point accumulators (round 4): every tile of the convolution ADDS its partial sums to a per-layer
int64 accumulator, and the launch that consumes the statistics finalises them in its prologue -- forward: fva_conv_fwd_acc ->
fva_bn_silu_apply_acc / fva_conv1x1_fwd_apply_acc / fva_bn_acc_finalize; backward: fva_conv_dgrad_bnstats (acc) / fva_bn_silu_bwd_reduce_acc
-> fva_bn_silu_bwd_apply_acc -- with no finalize launch in between.  Checked against the table form (fva_conv_fwd + fva_bn_finalize +
fva_bn_silu_apply, fva_bn_bwd_finalize + fva_bn_silu_bwd_apply), which the oracle-parity tests pin:

  * y is the same launch with another epilogue store: bit-identical;
  * the sums are the same fp32 partials added exactly (integers) instead of in double: mean / rstd / scale / shift / running statistics
    within 1e-6 relative (bit-equal in practice), z within one bf16 step of that; dgamma / dbeta within 1e-5;
  * integer addition is associative: two runs give the same bits whatever order the atomics land in;
  * each direction's consumer returns the OTHER direction's accumulator to zero, so a step (and a graph replay) starts clean.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


#        B   Cin  Cout  HW  k  s  dtype            kernel that produces the statistics
CASES = [
    (4, 64, 128, 40, 3, 1, torch.bfloat16),       # 128 x 128 tile
    (4, 128, 64, 40, 1, 1, torch.bfloat16),       # 256 x 64 tile
    (2, 32, 64, 64, 3, 1, torch.bfloat16),        # patch kernel (pconv)
    (2, 32, 64, 64, 3, 2, torch.bfloat16),        # patch kernel, stride 2
    (32, 128, 256, 40, 3, 1, torch.bfloat16),     # 8-phase 256 x 256 tile
    (2, 64, 128, 24, 3, 2, torch.float32),        # fp32 sibling
    (3, 64, 32, 20, 1, 1, torch.float32),
    (8, 32, 64, 128, 3, 1, torch.bfloat16),       # 131072 output pixels: two replicas of the accumulator (patch kernel)
    (16, 64, 32, 96, 1, 1, torch.bfloat16),       # 147456 pixels: four replicas (256 x 64 tile)
]


def block(case, seed=0):
    import torch.nn as nn
    B, Cin, Cout, HW, k, s, dt = case
    torch.manual_seed(seed)
    conv = nn.Conv2d(Cin, Cout, k, s, k // 2, bias=False).to(dev())
    bn = nn.BatchNorm2d(Cout).to(dev())
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(B, Cin, HW, HW, device=dev())
    return conv, bn, x


def run(case, acc_on, backward=False):
    import fastvision_amd
    from fastvision_amd import ops
    conv, bn, x = block(case)
    prev = ops.set_bn_accumulators(acc_on)
    try:
        with fastvision_amd.compute_dtype(case[6]):
            x.requires_grad_(True)
            z = ops.conv_bn_silu(x, conv, bn)
            s = z._fva_prod
            kern = ops._lib.load().fva_conv_last_kernel().decode()
            torch.cuda.synchronize()
            out = dict(z=z.detach().float().clone(), y=s.y.clone(), mean=s.mean.clone(), rstd=s.rstd.clone(), scale=s.scale.clone(), shift=s.shift.clone(),
                       rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nbt=int(bn.num_batches_tracked), kern=kern)
            st = getattr(bn.weight, '_fva_acc', None)
            out['acc'] = None if st is None else (st.buf.clone(), list(st.state))
            if backward:
                g = torch.Generator().manual_seed(11)
                gz = torch.randn(z.shape, generator=g).to(dev()).to(z.dtype)
                z.backward(gz)
                torch.cuda.synchronize()
                out.update(dx=x.grad.float().clone(), dw=conv.weight.grad.clone(), dgamma=bn.weight.grad.clone(), dbeta=bn.bias.grad.clone())
                out['acc_after'] = None if st is None else (st.buf.clone(), list(st.state))
    finally:
        ops.set_bn_accumulators(prev)
    return out


@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join(str(v).replace('torch.', '') for v in c))
def test_accumulator_form_matches_the_table_form(case):
    a, t = run(case, True, backward=True), run(case, False, backward=True)
    assert t['acc'] is None and a['acc'] is not None
    assert a['kern'] == t['kern']
    assert torch.equal(a['y'], t['y'])
    for k in ('mean', 'rstd', 'scale', 'shift', 'rm', 'rv'):
        torch.testing.assert_close(a[k], t[k], rtol=1e-6, atol=1e-7, msg=lambda m, k=k: f'{k}: {m}')
    assert a['nbt'] == t['nbt'] == 1
    bf = case[6] == torch.bfloat16
    step = 2 ** -7 if bf else 1e-5
    assert ((a['z'] - t['z']).abs() <= step * t['z'].abs().clamp_min(1.0)).all()
    # after the forward pass: the forward sums are still there (its consumer cannot zero what its other blocks read), the backward accumulator is zero
    buf, state = a['acc']
    assert state == [2, 0] and buf[0].any() and not buf[1].any()
    # backward: the same sums added exactly instead of in double -> dgamma / dbeta to 1e-5 of their scale, dx / dw to a rounding step of theirs
    for k, tol in (('dgamma', 1e-5), ('dbeta', 1e-5), ('dx', 2 ** -6 if bf else 1e-5), ('dw', 2e-3 if bf else 1e-5)):
        err = ((a[k] - t[k]).abs().max() / t[k].abs().max().clamp_min(1e-20)).item()
        assert err <= tol, (k, err)
    buf, state = a['acc_after']
    assert state == [0, 2] and not buf[0].any() and buf[1].any(), 'the backward consumer returns the forward accumulator to zero'


def test_second_step_starts_from_zero_accumulators_and_repeats_the_first():
    """Two identical steps through one layer: the second forward's consumer zeroes the backward accumulator of the first step, the second
    backward finds it clean -- and everything repeats bit for bit (integer sums do not depend on the order of the atomics)."""
    import fastvision_amd
    from fastvision_amd import ops
    case = CASES[0]
    conv, bn, x = block(case)
    outs = []
    with fastvision_amd.compute_dtype(case[6]):
        for _ in range(2):
            bn.running_mean.zero_(); bn.running_var.fill_(1)
            xx = x.clone().requires_grad_(True)
            z = ops.conv_bn_silu(xx, conv, bn)
            st = bn.weight._fva_acc
            torch.cuda.synchronize()
            assert not st.buf[1].any() and st.state == [2, 0]
            conv.weight.grad = bn.weight.grad = bn.bias.grad = None
            z.backward(torch.ones_like(z))
            torch.cuda.synchronize()
            outs.append((z.detach().clone(), xx.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_var.clone()))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


def test_accumulator_form_is_run_to_run_bit_identical():
    case = CASES[0]
    a, b = run(case, True, backward=True), run(case, True, backward=True)
    for k in ('z', 'y', 'mean', 'rstd', 'scale', 'shift', 'rm', 'rv', 'dx', 'dw', 'dgamma', 'dbeta'):
        assert torch.equal(a[k], b[k]), k


def test_fixed_point_sum_is_exact_and_poisoned_by_nan():
    """fva_conv_fwd_acc + fva_bn_acc_finalize on a 1x1 layer whose output is known: the mean and variance of the accumulator form equal a
    float64 reduction of the stored outputs' fp32 partial sums to 1e-6, and a NaN input reads NaN statistics (as a floating-point sum would)."""
    from fastvision_amd import _lib, ops
    B, Cc, N, H = 2, 64, 64, 16
    dt = torch.float32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H + 2, H + 2, Cc, generator=g).to(dev())
    x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0
    w = (torch.randn(N, Cc, 1, 1, generator=g) / 8).to(dev())
    d = _lib.ConvDesc(ops._code(dt), B, H, H, Cc, N, 1, 1, 1, 1)
    wf, _ = ops.packed_weights(w, d, dt, cache=False)
    M = B * H * H
    for poison in (False, True):
        if poison:
            x[1, 5, 5, 3] = float('nan')
        y = torch.empty(M, N, device=dev())
        R = 4 if poison else 1               # (also: four replicas give the same sums as one)
        acc = torch.zeros(R * 5 * N, dtype=torch.int64, device=dev())
        other = torch.ones(R * 5 * N, dtype=torch.int64, device=dev())
        gamma, beta = torch.ones(N, device=dev()), torch.zeros(N, device=dev())
        mean, rstd, scale, shift = (torch.empty(N, device=dev()) for _ in range(4))
        _lib.call('fva_conv_fwd_acc', C.byref(d), ops._p(x), ops._p(wf), ops._p(y), ops._p(acc), R, ops._stream())
        fin = _lib.BnFwdAcc(acc.data_ptr(), other.data_ptr(), R, gamma.data_ptr(), beta.data_ptr(), None, None, None, 0.1, 1e-5,
                            mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
        _lib.call('fva_bn_acc_finalize', C.byref(fin), M, N, ops._stream())
        torch.cuda.synchronize()
        assert not acc.any() and not other.any()
        if poison:
            assert torch.isnan(mean).all() and torch.isnan(scale).all()
        else:
            yd = y.double()
            torch.testing.assert_close(mean.double(), yd.mean(0), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(rstd.double(), 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5), rtol=1e-5, atol=0)


def test_abandoned_forward_does_not_leak_into_the_next_one():
    """A producer whose consumer never ran (an exception between the two launches) leaves sums behind; the next forward of that layer
    clears them first."""
    import fastvision_amd
    from fastvision_amd import ops
    case = CASES[0]
    conv, bn, x = block(case)
    with fastvision_amd.compute_dtype(case[6]):
        z0 = ops.conv_bn_silu(x, conv, bn).detach().clone()
        st = bn.weight._fva_acc
        st.buf[0].fill_(12345)       # what an abandoned producer leaves
        st.state[0] = 1
        bn.running_mean.zero_(); bn.running_var.fill_(1)
        z1 = ops.conv_bn_silu(x, conv, bn).detach().clone()
    torch.cuda.synchronize()
    assert torch.equal(z0, z1) and st.state == [2, 0]


def test_whole_step_with_accumulators_matches_the_table_form_and_repeats_bit_for_bit():
    """YOLOv3 train step (B = 2, 128 px, bf16): accumulators on twice (bit-identical: the atomics' order does not matter) and off (table
    form): loss within 1e-5 relative, gradients within 2e-2 of each tensor's scale (bf16 re-rounding of activations whose statistics
    moved in the last bit), every accumulator back at zero."""
    import fastvision_amd
    from fastvision_amd import ops
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    images, tg = synthetic_batch(2, 128)
    out = []
    with fastvision_amd.compute_dtype(torch.bfloat16):
        for on in (True, True, False):
            prev = ops.set_bn_accumulators(on)
            try:
                torch.manual_seed(5)
                net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                             training=True).to(dev()).train()
                crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
                pred = net(images.to(dev()))
                loss = crit(pred, tg.to(dev()))
                loss.backward()
                torch.cuda.synchronize()
                out.append((loss.detach().clone(), [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]))
                if on:
                    accs = [p._fva_acc for p in net.parameters() if hasattr(p, '_fva_acc')]
                    assert len(accs) >= 70
                    assert all(a.state == [0, 2] and not a.buf[0].any() for a in accs), 'after a whole step every forward accumulator is zero again'
            finally:
                ops.set_bn_accumulators(prev)
    a, b, t = out
    assert torch.equal(a[0], b[0])
    for u, v in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.equal(u, v)
    assert abs(float(a[0]) - float(t[0])) <= 1e-5 * abs(float(t[0]))
    worst = max(((u - v).abs().max() / v.abs().max().clamp_min(1e-20)).item() for u, v in zip(a[1], t[1]))
    print('accumulator vs table form, worst gradient element / tensor scale: %.2e' % worst)
    assert worst < 2e-2


@pytest.mark.parametrize('chans', [(64, 32), (256, 128)], ids=['thin-fused-1x1', 'wide-fused-1x1'])
def test_residual_chain_accumulators_vs_table(chans):
    """conv3x3 -> [ResidualBlock: 1x1 -> 3x3 + identity] under defer_apply_scope (B = 4, 32 x 32, bf16): the chain where one launch consumes the
    statistics of the block before it and produces its own (fva_conv1x1_fwd_apply_acc; 256 -> 128 finalises in the prologue, 64 -> 32 through
    fva_bn_acc_finalize), and where the dgrad epilogues ADD the backward sums of the producer layer (fva_bn_bwd_fuse::acc)."""
    import fastvision_amd
    from fastvision_amd import ops
    from fastvision_amd.classfication.models.darknet53 import ConvBlock3x3, ResidualBlock
    cin, mid = chans
    outs = {}
    with fastvision_amd.compute_dtype(torch.bfloat16):
        for on in (True, True, False):
            prev = ops.set_bn_accumulators(on)
            try:
                torch.manual_seed(2)
                head = ConvBlock3x3(cin // 2, cin).to(dev()).train()
                blk = ResidualBlock(cin, mid).to(dev()).train()
                x = torch.randn(4, cin // 2, 32, 32, device=dev(), requires_grad=True)
                n0 = ops._DEFER['fused']
                with ops.defer_apply_scope():
                    z = blk(head(x))
                assert ops._DEFER['fused'] - n0 == 1, 'the 1x1 layer must take the apply pass of the block before it'
                gz = torch.randn(z.shape, generator=torch.Generator().manual_seed(4)).to(dev()).to(z.dtype)
                z.backward(gz)
                torch.cuda.synchronize()
                params = list(head.parameters()) + list(blk.parameters())
                rec = [z.detach().float().clone(), x.grad.clone()] + [p.grad.clone() for p in params] + [b.clone().float() for m in (head, blk) for b in m.buffers()]
                outs.setdefault(on, []).append(rec)
                if on:
                    accs = [p._fva_acc for p in params if hasattr(p, '_fva_acc')]
                    assert len(accs) == 3 and all(a.state == [0, 2] and not a.buf[0].any() for a in accs)
            finally:
                ops.set_bn_accumulators(prev)
    a, b = outs[True]
    for u, v in zip(a, b):
        assert torch.equal(u, v), 'two runs of the accumulator form differ'
    for i, (u, v) in enumerate(zip(a, outs[False][0])):
        err = ((u.float() - v.float()).abs().max() / v.float().abs().max().clamp_min(1e-20)).item()
        assert err < 2e-2, (i, err)


def test_accumulators_of_another_batch_size_do_not_replace_the_first_ones():
    """A layer used at two sizes whose replica counts differ (here 65536-pixel threshold: 2 x 128 x 128 -> 1 copy, 8 x 128 x 128 -> 2) keeps
    BOTH accumulator sets alive: a captured graph holds the addresses of the set it was recorded with."""
    import fastvision_amd
    from fastvision_amd import ops
    conv, bn, _ = block((2, 32, 64, 128, 3, 1, torch.bfloat16))
    with fastvision_amd.compute_dtype(torch.bfloat16):
        ops.conv_bn_silu(torch.randn(2, 32, 128, 128, device=dev(), requires_grad=True), conv, bn)
        small = bn.weight._fva_acc
        ops.conv_bn_silu(torch.randn(8, 32, 128, 128, device=dev(), requires_grad=True), conv, bn)
        big = bn.weight._fva_acc
        ops.conv_bn_silu(torch.randn(2, 32, 128, 128, device=dev(), requires_grad=True), conv, bn)
    torch.cuda.synchronize()
    assert small.replicas == 1 and big.replicas == 2 and small is not big
    assert bn.weight._fva_acc is small and small.buf.data_ptr() != big.buf.data_ptr()
    assert set(bn.weight._fva_accs.values()) == {small, big}


def test_frozen_backbone_step_with_and_without_accumulators():
    """Only the neck and the heads train (backbone parameters frozen: its blocks build no autograd state and keep the table form, the first
    trainable block's backward has no producer to hand statistics to): two such steps with the accumulators on equal the table form to the
    dtype's noise, repeat bit for bit, and leave every accumulator in its end-of-step state."""
    import fastvision_amd
    from fastvision_amd import ops
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    images, tg = synthetic_batch(2, 128)
    out = []
    with fastvision_amd.compute_dtype(torch.bfloat16):
        for on in (True, True, False):
            prev = ops.set_bn_accumulators(on)
            try:
                torch.manual_seed(5)
                net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                             training=True).to(dev()).train()
                for p in net.backbone.parameters():
                    p.requires_grad_(False)
                crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
                losses = []
                for _ in range(2):
                    for p in net.parameters():
                        p.grad = None
                    loss = crit(net(images.to(dev())), tg.to(dev()))
                    loss.backward()
                    losses.append(loss.detach().clone())
                torch.cuda.synchronize()
                assert all(p.grad is None for p in net.backbone.parameters())
                grads = [p.grad.clone() for p in net.parameters() if p.grad is not None]
                assert len(grads) > 40
                out.append((losses, grads, [b.clone() for b in net.buffers()]))
                if on:
                    accs = [p._fva_acc for p in net.parameters() if hasattr(p, '_fva_acc')]
                    assert 0 < len(accs) < 40, 'only the trainable blocks keep accumulators'
                    assert all(a.state == [0, 2] and not a.buf[0].any() for a in accs)
            finally:
                ops.set_bn_accumulators(prev)
    a, b, t = out
    for u, v in zip(a[0] + a[1] + a[2], b[0] + b[1] + b[2]):
        assert torch.equal(u, v)
    assert abs(float(a[0][1]) - float(t[0][1])) <= 1e-4 * abs(float(t[0][1]))
    worst = max(((u - v).abs().max() / v.abs().max().clamp_min(1e-20)).item() for u, v in zip(a[1], t[1]))
    assert worst < 2e-2, worst
