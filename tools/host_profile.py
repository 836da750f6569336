#!/usr/bin/env python3
"""Where does the HOST spend a training step?  cProfile over a few steps of the bench workload (library surface).
usage: python tools/host_profile.py [steps]   -> top functions by own time and by cumulative time on stdout"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    import fastvision_amd
    from fastvision_amd import FusedAdam
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    dev = torch.device('cuda', 0)
    fastvision_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(1)
    net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                 in_channels=3, num_classes=80, training=True).to(dev).train()
    crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
    images, targets = synthetic_batch(32, 640)
    images, targets = images.to(dev), targets.to(dev)

    def step():
        pred = net(images)
        opt.zero_grad()
        loss = crit(pred, targets)
        loss.backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # host-only pace: issue steps without waiting for the GPU
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    print(f'host issue time {host * 1e3:.2f} ms/step (wall incl. GPU {(time.perf_counter() - t0) / steps * 1e3:.2f})')
    # the backward Functions run on autograd's device thread: a second profiler is switched on from inside that thread
    from fastvision_amd import ops
    pr_bwd, armed = cProfile.Profile(), []
    orig = ops.HeadFn.backward

    def first_backward(ctx, dout):
        if not armed:
            armed.append(1)
            pr_bwd.enable()
        return orig(ctx, dout)
    ops.HeadFn.backward = staticmethod(first_backward)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    print(f'---- backward thread, by tottime (per {steps} steps)')
    pstats.Stats(pr_bwd).sort_stats('tottime').print_stats(40)
    print(f'---- backward thread, by cumulative (per {steps} steps)')
    pstats.Stats(pr_bwd).sort_stats('cumulative').print_stats(30)
    for key in ('tottime', 'cumulative'):
        print(f'---- by {key} (per {steps} steps)')
        pstats.Stats(pr).sort_stats(key).print_stats(35)


if __name__ == '__main__':
    main()
