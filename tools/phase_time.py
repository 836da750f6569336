"""Forward and backward of the benchmarked step timed separately (events on the current stream), with switches flipped in-process:
python tools/phase_time.py [steps]   ->  one line per setting."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastvision_amd
from fastvision_amd import ops
from fastvision_amd.classfication.models import darknet53
from fastvision_amd.detection.head import yolov3head
from fastvision_amd.detection.models import yolov3
from fastvision_amd.detection.neck import yolov3neck
from fastvision_amd.loss import Yolov3Loss
from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
fastvision_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(20220504)
net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3], in_channels=3,
             num_classes=80, training=True).to(dev).train()
crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
images, targets = synthetic_batch(32, 640)
images, targets = images.to(dev), targets.to(dev)


def measure(label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    f = b = 0.0
    for i in range(steps + 3):
        ev[0].record()
        pred = net(images)
        loss = crit(pred, targets)
        ev[1].record()
        for p in net.parameters():
            p.grad = None
        ev[2].record()
        loss.backward()
        ops.join_side_stream(force=True)
        ev[3].record()
        torch.cuda.synchronize()
        if i >= 3:
            f += ev[0].elapsed_time(ev[1])
            b += ev[2].elapsed_time(ev[3])
        del pred, loss
    print('%-28s forward+loss %.3f ms   backward %.3f ms' % (label, f / steps, b / steps), flush=True)


for rep in range(2):
    for acc in ((True,) if os.environ.get('FVA_PHASE_ON_ONLY') else (False, True)):
        ops.set_bn_accumulators(acc)
        measure('accumulators %s' % ('on' if acc else 'off'))
