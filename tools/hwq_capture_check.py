#!/usr/bin/env python3
"""One capture + three replays of the YOLOv3 train step with the weight gradients as a second graph branch (side_stream=True),
for the GPU_MAX_HW_QUEUES question of DESIGN.md section 3.3c (round 2 saw a segfault during capture with GPU_MAX_HW_QUEUES=2 and kept no
log).  Run once per setting under `python -X faulthandler`:   GPU_MAX_HW_QUEUES=2 python -X faulthandler tools/hwq_capture_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastvision_amd
from fastvision_amd import FusedAdam
from fastvision_amd.graphs import GraphedTrainStep
from fastvision_amd.classfication.models import darknet53
from fastvision_amd.detection.head import yolov3head
from fastvision_amd.detection.models import yolov3
from fastvision_amd.detection.neck import yolov3neck
from fastvision_amd.loss import Yolov3Loss
from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch

dev = 'cuda:0'
side = len(sys.argv) < 2 or sys.argv[1] != 'single'
torch.manual_seed(1)
net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
             in_channels=3, num_classes=80, training=True).to(dev).train()
crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
opt = FusedAdam(net.parameters(), lr=1e-4, capturable=True)
im, tg = synthetic_batch(4, 256)
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'), '| two-branch capture' if side else '| single-stream capture', flush=True)
step = GraphedTrainStep(net, lambda p, t: crit(p, t), opt, im.to(dev), tg.to(dev), side_stream=side)
print('captured', flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    loss = step()
torch.cuda.synchronize()
print(f'3 replays ok, {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms each, loss {float(loss):.5f}', flush=True)
