import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench_conv import run
# config 2: B=8, 416x416 -> stages at 208, 104, 52, 26, 13
L = [(32, 64, 416, 3, 2), (64, 32, 208, 1, 1), (32, 64, 208, 3, 1), (64, 128, 208, 3, 2), (128, 64, 104, 1, 1), (64, 128, 104, 3, 1),
     (128, 256, 104, 3, 2), (256, 128, 52, 1, 1), (128, 256, 52, 3, 1), (256, 512, 52, 3, 2), (512, 256, 26, 1, 1), (256, 512, 26, 3, 1),
     (512, 1024, 26, 3, 2), (1024, 512, 13, 1, 1), (512, 1024, 13, 3, 1)]
for ci, co, h, k, s in L:
    r = run(8, ci, co, h, h, k, s, iters=10, dtype=torch.float32)
    print(f'{ci:>5} {co:>5} {h:>4} {k} {s} | ' + ' | '.join(f'{p} {r[p][0]:8.1f} us {r[p][1]:6.1f} TF' for p in r), flush=True)
