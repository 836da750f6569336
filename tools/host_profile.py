"""cProfile of the host side of the train step (how long Python needs to ISSUE a step)."""
import cProfile, pstats, sys, os, time
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
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3], training=True).to(dev).train()
crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
images, tg = synthetic_batch(B, 320)
images, tg = images.to(dev), tg.to(dev)
def step():
    pred = net(images); opt.zero_grad(); loss = crit(pred, tg); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
print(f'host issue time per step: {host*1e3:.2f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
