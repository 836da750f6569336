#!/usr/bin/env python3
"""Host time of the data-parallel step's extra work on one GPU: a one-rank RCCL group with the reducer told that the world has two
ranks (buckets are filled, narrowed to bf16 and all-reduced for real; one rank averages to the identity).  Prints ms per step
(wall, host issue) with and without the reducer.  usage: python tools/dp_host_time.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'))
import torch
import torch.distributed as dist


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    import fastvision_amd
    from fastvision_amd import FusedAdam, parallel
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    dev = torch.device('cuda', 0)
    fastvision_amd.set_compute_dtype(torch.bfloat16)
    images, targets = synthetic_batch(32, 640)
    images, targets = images.to(dev), targets.to(dev)
    for mode in ('plain', 'reducer'):
        torch.manual_seed(1)
        net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                     in_channels=3, num_classes=80, training=True).to(dev).train()
        crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        red = parallel.GradientReducer(net.parameters(), bucket_bytes=16 << 20, average=False, bucket_dtype=torch.bfloat16, world=2) if mode == 'reducer' else None

        def step():
            pred = net(images)
            opt.zero_grad()
            loss = crit(pred, targets)
            loss.backward()
            if red is not None:
                red.finish()
            opt.step()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        import gc
        gc.collect(); gc.freeze()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        hs = []
        for i in range(steps):
            h0 = time.perf_counter()
            step()
            hs.append(round((time.perf_counter() - h0) * 1e3, 1))
            marks[i + 1].record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(mode, 'GPU ms per step:', [round(marks[i].elapsed_time(marks[i + 1]), 1) for i in range(steps)], flush=True)
        print(mode, 'host ms per step:', hs, flush=True)
        print(f'{mode}: {wall / steps * 1e3:.2f} ms/step wall, {host / steps * 1e3:.2f} ms/step host issue', flush=True)
        if os.environ.get('DP_PROFILE') and red is not None:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(5):
                step()
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
        if red is not None:
            red.remove()
        del net, opt, crit, red
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
