#!/usr/bin/env python3
"""Does the eager train step make torch's caching allocator go to the driver in steady state?  Prints, per step, the allocator's
counters (live bytes, reserved bytes, segments per pool, driver calls).   usage: python tools/alloc_probe.py [steps] [detach]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastvision_amd
from fastvision_amd import FusedAdam
from fastvision_amd.classfication.models import darknet53
from fastvision_amd.detection.head import yolov3head
from fastvision_amd.detection.models import yolov3
from fastvision_amd.detection.neck import yolov3neck
from fastvision_amd.loss import Yolov3Loss
from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
detach = len(sys.argv) > 2 and sys.argv[2] == 'detach'
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3], training=True).to(dev).train()
crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
images, targets = [t.to(dev) for t in synthetic_batch(32, 640)]


def step():
    pred = net(images)
    opt.zero_grad()
    loss = crit(pred, targets)
    loss.backward()
    opt.step()
    return loss.detach() if detach else loss


keys = ['allocated_bytes.all.current', 'allocated_bytes.all.peak', 'reserved_bytes.all.current', 'segment.small_pool.current', 'segment.large_pool.current',
        'num_device_alloc', 'num_device_free', 'allocation.all.current']
prev = None
for i in range(steps):
    loss = step()
    if i % 1 == 0:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        row = [st.get(k, 0) for k in keys]
        print(f'step {i:2d}: live {row[0] / 2**30:6.2f} GiB (peak {row[1] / 2**30:6.2f}) reserved {row[2] / 2**30:6.2f} GiB, segments small {row[3]} large {row[4]}, '
              f'device mallocs {row[5]} frees {row[6]}, live allocations {row[7]}' + ('' if prev is None else f'  [+{row[5] - prev[5]} mallocs, {row[7] - prev[7]:+d} allocations]'), flush=True)
        prev = row
# without a sync between the steps (the benchmark's timed loop)
torch.cuda.synchronize()
a0 = torch.cuda.memory_stats(dev)['num_device_alloc']
for i in range(10):
    loss = step()
torch.cuda.synchronize()
st = torch.cuda.memory_stats(dev)
print(f'10 steps without a sync in between: +{st["num_device_alloc"] - a0} device mallocs, reserved {st["reserved_bytes.all.current"] / 2**30:.2f} GiB, live allocations {st["allocation.all.current"]}')
