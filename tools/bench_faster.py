#!/usr/bin/env python3
"""Faster R-CNN (the reference's demo model: VGG16 stride-16 backbone + RPN + Fast head) training step on the HIP ops, at
BASELINE config 5's input: 4 x 3 x 800 x 1333 synthetic images, 20 classes, the demo's defaults (128 + 128 RPN samples,
16 + 48 Fast samples per image), bf16 compute, the demo's step: SGD (momentum 0.937, Nesterov) after gradient-norm clipping at 10.  Prints one JSON object.
usage: python tools/bench_faster.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fastvision_amd
from fastvision_amd.demos.faster_rcnn.cfg._fit import clip_gradient
from fastvision_amd.demos.faster_rcnn.models import Faster_Rcnn

DEV = 'cuda:0'


def main(steps=None, warmup=3, cpu_baseline=True, emit=True):
    if steps is None:
        steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B, H, W, NC = 4, 800, 1333, 20
    torch.manual_seed(0)
    scales, ratios = [128, 256, 512], [0.5, 1, 2]
    base = torch.tensor([[(s * s / r) ** 0.5, s * s / (s * s / r) ** 0.5] for r in ratios for s in scales], dtype=torch.float32)
    model = Faster_Rcnn(training=True, num_classes=NC, base_anchors=base).to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.937, nesterov=True)      # demos/faster_rcnn/train.py:102
    g = torch.Generator().manual_seed(1)
    images = torch.rand(B, 3, H, W, generator=g).to(DEV)
    T = 28
    tb = torch.sort(torch.cat([torch.arange(B), torch.randint(0, B, (T - B,), generator=g)]))[0].float()
    wh = torch.exp(np.log(0.08) + (np.log(0.6) - np.log(0.08)) * torch.rand(T, 2, generator=g))
    xy = wh / 2 + (1 - wh) * torch.rand(T, 2, generator=g)
    targets = torch.cat([tb[:, None], torch.randint(0, NC, (T, 1), generator=g).float(), xy, wh], 1).to(DEV)

    def step():
        _, a, b, c, d = model(images, targets.clone())
        opt.zero_grad()
        loss = a + b + c + d
        loss.backward()
        clip_gradient(model, 10.)                  # part of the reference's step (cfg/_fit.py:46); one read-back
        opt.step()
        return loss
    with fastvision_amd.compute_dtype(torch.bfloat16):
        for _ in range(max(1, warmup)):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    # exclusive per-class convolution figures (HIP events inside the library, weight gradients on the launch stream for this probe)
    from fastvision_amd import ops as fva_ops
    from fastvision_amd.profiler import KernelTimer
    was = fva_ops.set_wgrad_side_stream(False)
    with fastvision_amd.compute_dtype(torch.bfloat16):
        step()
        torch.cuda.synchronize()
        with KernelTimer(pool=4096) as kt:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
    fva_ops.set_wgrad_side_stream(was)
    summ = kt.summary()
    dom = max(summ, key=lambda k: summ[k]['ms_total'])
    peak = 2500.0
    roofline = {'bound': 'mfma', 'kernel': {'conv_fwd': 'implicit-GEMM convolution forward (igemm8_kernel / igemm_kernel, bias + ReLU epilogue)',
                                              'conv_dgrad': 'implicit-GEMM convolution dgrad', 'conv_wgrad': 'weight gradient (wgrad8_kernel / wgrad_kernel + reduce)'}[dom],
                'achieved': round(summ[dom]['tflops'], 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(summ[dom]['tflops'] / peak, 4), 'traffic': None,
                'avg_launch_ms': round(summ[dom]['ms_avg'], 4), 'launches_per_step': summ[dom]['launches'] // 3,
                'note': 'dominant convolution class by summed launch time; exclusive timings of a 3-step probe with every kernel on one stream'}
    kernels = {k: {'tflops': round(v['tflops'], 2), 'ms_per_step': round(v['ms_total'] / 3, 3), 'launches_per_step': v['launches'] // 3} for k, v in summ.items()}
    cpu = None
    if cpu_baseline:
        # CPU baseline: the oracle (oracle/faster.py: plain torch fp32 ops over the same parameters) on a bounded sample -- one image of
        # the same size per step, on this host's cores
        import copy
        from oracle import faster as OF
        cores = len(os.sched_getaffinity(0))                      # scheduler affinity capped by the cgroup CPU quota (containers)
        try:
            quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
            if quota != 'max':
                cores = min(cores, max(1, int(float(quota) / float(period) + 0.5)))
        except (OSError, ValueError):
            pass
        cores = max(1, min(cores, 64))
        torch.set_num_threads(cores)
        ref = copy.deepcopy(model).cpu().float()
        img1, tg1 = images[:1].cpu(), targets[targets[:, 0] == 0].cpu()
        cpu_times = []
        for i in range(2):
            t0 = time.perf_counter()
            for p in ref.parameters():
                p.grad = None
            out = OF.training_losses(ref, img1, tg1, [(None, None)] * 2)
            torch.stack([l.reshape(()) for l in out[1:]]).sum().backward()
            cpu_times.append(time.perf_counter() - t0)
        cpu = {'value': round(1.0 / cpu_times[-1], 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
               'sample': f'CPU oracle (oracle/faster.py, fp32), forward + backward of 1x3x{H}x{W}: 1 warm-up + 1 timed step, {cpu_times[-1]:.2f} s'}
    out = {'metric': 'images/sec (4x3x800x1333) Faster R-CNN (VGG16 + RPN + Fast head) train step, 1 MI355X', 'value': round(B / (ms * 1e-3), 2),
           'unit': 'images/sec', 'n_gpus': 1, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(ms, 2), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': f'Faster R-CNN (the reference demo: VGG16 stride-16 backbone + RPN + Fast head) train step {B}x3x{H}x{W} bf16, '
                                  f'{NC} classes, 128+128 RPN / 16+48 Fast samples per image, grad-norm clip + Nesterov SGD (BASELINE config 5)',
                      'global_batch': B, 'parallelism': 'dp1'},
           'roofline': roofline, 'kernels': kernels, 'loss': round(float(loss.detach()), 4),
           'note': 'conv / pool / RoIAlign / matchers / proposal layer / fully connected layers / losses on the HIP kernels; host-inclusive '
                   '(each step reads sample counts back like the reference)'}
    if cpu is not None:
        out['cpu_baseline'] = cpu
    if emit:
        print(json.dumps(out))
    return out


if __name__ == '__main__':
    main()
