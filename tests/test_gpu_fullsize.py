"""BASELINE config 3 at its full size: ONE YOLOv3 train step on 32x3x640x640 -- the workload bench.py times, with the kernels it
dispatches there (8-phase 256x256 igemm / wgrad tiles, weight gradients on the low-priority side stream) -- against the CPU oracle
(oracle.train, pinned by the reference's vectors) on the same synthetic_batch(32, 640).

Bars (north_star; the tolerance of each is in the assert that enforces it):
  * anchor/target indexing: bit-exact;
  * fp32 (exact f32-input MFMA): heads and loss within 1e-3 of the oracle, per-tensor gradient norms within 2e-3;
  * bf16 (the bench dtype: bf16 storage, fp32 accumulate / statistics / loss): loss within 2e-2, per-tensor gradient norms within
    5e-2 in the median and 2.5e-1 at worst; the observed values are printed.
The oracle step costs ~15-40 s of CPU and ~60 GB of host memory at this size; it runs once per module.
"""
import os

import numpy as np
import pytest
import torch

from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
B, S = 32, 640
BF16_ELEMENTWISE_GATE = 4e-1        # observed in round 4: 6e-4 (head bias) ... 2.5e-1 (neck.up1.squeeze, BatchNorm scales); a mis-packed tap or channel reads ~1

# Parameter tensors compared ELEMENT by element with the oracle's gradient (a per-tensor norm survives a permutation of taps or
# channels inside a kernel's tile packing; max |g - g_ref| / max |g_ref| does not: a permuted tensor reads ~1): the stem, every
# residual stage with the kernel that serves it at this size, the down-sampling convolutions, the neck (incl. the 768- and
# 384-channel concat consumers) and the heads, plus BatchNorm scales / shifts.
GRAD_SAMPLE = [
    'backbone.conv0.conv.weight',            # stem: stem_wgrad via wgrad_kernel<thin>
    'backbone.conv1.conv.weight',            # 32 -> 64 stride 2 @640: pwgrad<32,64,2>
    'backbone.res1.0.conv2.conv.weight',     # 32 -> 64 @320: pwgrad<32,64,1>
    'backbone.res1.0.conv1.conv.weight',     # 64 -> 32 1x1: wgrad128thin
    'backbone.res2.1.conv1.conv.weight',     # 128 -> 64 1x1
    'backbone.conv3.conv.weight',            # 128 -> 256 stride 2: wgrad8, two taps per column tile
    'backbone.res3.3.conv2.conv.weight',     # 128 -> 256 @80: wgrad8, two taps per column tile
    'backbone.res3.3.conv1.conv.weight',     # 256 -> 128 1x1: wgrad128
    'backbone.res4.2.conv2.conv.weight',     # 256 -> 512 @40: wgrad8
    'backbone.conv5.conv.weight',            # 512 -> 1024 stride 2
    'backbone.res5.1.conv2.conv.weight',     # 512 -> 1024 @20: wgrad8
    'neck.neck1.conv2.conv.weight', 'neck.up1.squeeze.conv.weight', 'neck.neck2.conv1.conv.weight', 'neck.neck3.conv1.conv.weight',
    'neck.neck3.conv4.conv.weight', 'neck.conv3.conv.weight',
    'head.heads.0.weight', 'head.heads.2.weight', 'head.heads.1.bias',
    'backbone.conv0.bn.weight', 'backbone.res3.3.conv2.bn.weight', 'backbone.res4.2.conv2.bn.bias', 'neck.neck2.conv5.bn.weight', 'neck.conv3.bn.bias',
]


def lib_model(seed=20220504):
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    torch.manual_seed(seed)
    m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
               in_channels=3, num_classes=80, training=True)
    return m.to(DEV).train()


@pytest.fixture(scope='module')
def oracle_step():
    """The oracle's step on the full batch: head maxima, loss, matcher output, per-parameter gradient norms (kept small: the 60 GB of
    autograd state are released before the GPU runs)."""
    from oracle import losses as ol, train as otrain
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    images, tg = synthetic_batch(B, S)
    ref, ref_crit = otrain.make_library(20220504)
    pred = ref(images)
    loss = ref_crit(pred, tg)
    loss.backward()
    out = {
        'heads': [p.detach().clone() for p in pred],
        'loss': float(loss),
        'match': ol.build_target([p.shape for p in pred], tg, ref.anchors_per_level, ref.backbone_strides_per_level),
        'gnorm': np.array([p.grad.double().norm().item() for _, p in ref.named_parameters()]),
        'names': [k for k, _ in ref.named_parameters()],
        'gsel': {k: p.grad.detach().clone() for k, p in ref.named_parameters() if k in GRAD_SAMPLE},
    }
    assert len(out['gsel']) == len(GRAD_SAMPLE), 'a sampled parameter name does not exist'

    del ref, pred, loss
    import gc
    gc.collect()
    return out


def gpu_step(dtype):
    import fastvision_amd
    from fastvision_amd import ops
    from fastvision_amd.loss import Yolov3Loss
    assert ops._SIDE['on'], 'the weight-gradient side stream is the product default and must be on in this test'
    images, tg = synthetic_batch(B, S)
    with fastvision_amd.compute_dtype(dtype):
        net = lib_model()
        crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
        pred = net(images.to(DEV))
        loss = crit(pred, tg.to(DEV))
        loss.backward()
        match = crit.build_target(pred, tg.to(DEV))
    torch.cuda.synchronize()
    heads = [p.detach().float().cpu() for p in pred]
    gnorm = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
    names = [k for k, _ in net.named_parameters()]
    finite = all(torch.isfinite(p.grad).all().item() for p in net.parameters())
    gpu_step.gsel = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters() if k in GRAD_SAMPLE}
    return heads, float(loss), match, gnorm, names, finite


def elementwise_grad_errors(want):
    """max |g - g_ref| / max |g_ref| per sampled tensor (the gradients of the last gpu_step)."""
    return {k: ((gpu_step.gsel[k].double() - want[k].double()).abs().max() / want[k].double().abs().max().clamp_min(1e-30)).item() for k in GRAD_SAMPLE}


def grad_cosines(want):
    """cosine between the GPU gradient and the oracle's, per sampled tensor: 1 - O(noise^2) for a correct tensor, ~0 for a permuted one."""
    out = {}
    for k in GRAD_SAMPLE:
        a, b = gpu_step.gsel[k].double().flatten(), want[k].double().flatten()
        out[k] = (a @ b / (a.norm() * b.norm()).clamp_min(1e-300)).item()
    return out


def check_match(got, want):
    locs, cats, xywh, anc = got
    rlocs, rcats, rxywh, ranc = want
    for l in range(3):
        assert torch.equal(locs[l][0].cpu(), rlocs[l][0]) and torch.equal(locs[l][1].cpu(), rlocs[l][1])
        assert torch.equal(locs[l][2].cpu(), rlocs[l][2]) and torch.equal(cats[l].cpu(), rcats[l])
        assert torch.equal(xywh[l].cpu(), rxywh[l]) and torch.equal(anc[l].cpu(), ranc[l])


def test_config3_bf16_full_size_step_vs_oracle(oracle_step):
    heads, loss, match, gnorm, names, finite = gpu_step(torch.bfloat16)
    assert names == oracle_step['names'] and finite
    assert [tuple(h.shape) for h in heads] == [(B, 3, 20, 20, 85), (B, 3, 40, 40, 85), (B, 3, 80, 80, 85)]
    check_match(match, oracle_step['match'])                                   # integer indices AND fp32 box targets: bit-exact
    herr = [((h - r).abs().max() / r.abs().max()).item() for h, r in zip(heads, oracle_step['heads'])]
    lrel = abs(loss - oracle_step['loss']) / abs(oracle_step['loss'])
    rel = np.abs(gnorm - oracle_step['gnorm']) / np.maximum(oracle_step['gnorm'], 1e-12)
    print(f'config3 bf16 B={B} {S}px: head max-err/scale {herr}, loss {loss:.6f} vs oracle {oracle_step["loss"]:.6f} (rel {lrel:.2e}), '
          f'gradient-norm rel dev median {np.median(rel):.2e} p90 {np.quantile(rel, 0.9):.2e} max {rel.max():.2e} ({names[int(rel.argmax())]})')
    # gates = about twice what was observed in rounds 2 and 3 (heads 9.6e-2 of the scale, loss 2.9e-5, gradient norms median 3.8e-3 /
    # max 5.7e-2): a regression of the benchmarked dtype by more than that fails
    assert lrel < 1e-3
    assert max(herr) < 1.3e-1                                                   # bf16 activations through 75 layers, random init
    assert np.median(rel) < 1e-2 and rel.max() < 1.2e-1
    # element by element (round 4): bf16 storage of every activation and activation gradient leaves each dW element a few percent of
    # the tensor's scale from the fp32 oracle (observed: up to 0.25 at the worst ELEMENT of a tensor whose norm is off by < 4e-2); a
    # mis-packed tap or channel would read ~1 -- and the fp32 run below, same kernels' fp32 siblings and the same host code, is at 6e-5.
    ew = elementwise_grad_errors(oracle_step['gsel'])
    print('config3 bf16 element-wise gradient error / tensor scale: ' + ', '.join(f'{k} {v:.2e}' for k, v in ew.items()))
    assert max(ew.values()) < BF16_ELEMENTWISE_GATE, max(ew, key=ew.get)
    cs = grad_cosines(oracle_step['gsel'])
    print('config3 bf16 gradient cosine with the oracle: min %.5f (%s)' % (min(cs.values()), min(cs, key=cs.get)))
    assert min(cs.values()) > 0.96, min(cs, key=cs.get)      # observed 0.9796 (neck.up1.squeeze) ... 0.99999 (head bias); a permuted tensor reads ~0


def test_config3_fp32_full_size_step_vs_oracle(oracle_step):
    heads, loss, match, gnorm, names, finite = gpu_step(torch.float32)
    assert names == oracle_step['names'] and finite
    check_match(match, oracle_step['match'])
    herr = [((h - r).abs().max() / r.abs().max()).item() for h, r in zip(heads, oracle_step['heads'])]
    lrel = abs(loss - oracle_step['loss']) / abs(oracle_step['loss'])
    rel = np.abs(gnorm - oracle_step['gnorm']) / np.maximum(oracle_step['gnorm'], 1e-12)
    print(f'config3 fp32 B={B} {S}px: head max-err/scale {herr}, loss rel {lrel:.2e}, gradient-norm rel dev median {np.median(rel):.2e} '
          f'max {rel.max():.2e} ({names[int(rel.argmax())]})')
    assert max(herr) < 1e-3 and lrel < 1e-3
    assert rel.max() < 2e-3
    ew = elementwise_grad_errors(oracle_step['gsel'])
    print('config3 fp32 element-wise gradient error / tensor scale: ' + ', '.join(f'{k} {v:.2e}' for k, v in ew.items()))
    assert max(ew.values()) < 5e-3, max(ew, key=ew.get)                         # exact-fp32 MFMA path: north_star's 1e-3 on loss / grads, 5e-3 at the worst ELEMENT
    cs = grad_cosines(oracle_step['gsel'])
    assert min(cs.values()) > 1 - 1e-6, min(cs, key=cs.get)
