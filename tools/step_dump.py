#!/usr/bin/env python3
"""One bf16 training step of the library model on a seeded synthetic batch; dumps the loss, every parameter's gradient
norm and a strided sample of each gradient.  Used to A/B kernel switches (FVA_IGEMM8, FVA_WGRAD8, ...) in child processes.
tools/step_dump.py <out.npz> [batch] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastvision_amd
from fastvision_amd.classfication.models import darknet53
from fastvision_amd.detection.head import yolov3head
from fastvision_amd.detection.models import yolov3
from fastvision_amd.detection.neck import yolov3neck
from fastvision_amd.loss import Yolov3Loss
from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch

out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 416
dev = 'cuda:0'
fastvision_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(20220504)
net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
             in_channels=3, num_classes=80, training=True).to(dev).train()
crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
images, targets = synthetic_batch(B, S)
pred = net(images.to(dev))
loss = crit(pred, targets.to(dev))
loss.backward()
res = {'loss': np.array([float(loss)])}
for k, p in net.named_parameters():
    g = p.grad.detach().float().flatten()
    res['norm/' + k] = np.array([float(g.norm())])
    res['samp/' + k] = g[::max(1, g.numel() // 64)].cpu().numpy()
np.savez(out, **res)
print('loss', float(loss))
