#!/usr/bin/env python3
"""Per-layer-class timing of the conv entry points at the bench workload (B=32, 640x640): tools/sweep_conv.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv import run

# (count, Cin, Cout, H(in), k, stride) for the library YOLOv3 at 640x640 input
LAYERS = [
    (1, 32, 64, 640, 3, 2), (1, 64, 32, 320, 1, 1), (1, 32, 64, 320, 3, 1),
    (1, 64, 128, 320, 3, 2), (2, 128, 64, 160, 1, 1), (2, 64, 128, 160, 3, 1),
    (1, 128, 256, 160, 3, 2), (8, 256, 128, 80, 1, 1), (8, 128, 256, 80, 3, 1),
    (1, 256, 512, 80, 3, 2), (8, 512, 256, 40, 1, 1), (8, 256, 512, 40, 3, 1),
    (1, 512, 1024, 40, 3, 2), (4, 1024, 512, 20, 1, 1), (4, 512, 1024, 20, 3, 1),
    # neck
    (3, 1024, 512, 20, 1, 1), (3, 512, 1024, 20, 3, 1), (1, 512, 256, 20, 1, 1),
    (1, 768, 256, 40, 1, 1), (2, 512, 256, 40, 1, 1), (3, 256, 512, 40, 3, 1), (1, 256, 128, 40, 1, 1),
    (1, 384, 128, 80, 1, 1), (2, 256, 128, 80, 1, 1), (3, 128, 256, 80, 3, 1),
]

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    flops = 0.0
    print(f'{"n":>2} {"Cin":>5} {"Cout":>5} {"H":>4} k s | ' + ' | '.join(f'{p:>7} us   TF' for p in tot) + ' | GFLOP')
    for n, ci, co, h, k, s in LAYERS:
        r = run(B, ci, co, h, h, k, s, iters=10)
        oh = (h - 1) // s + 1
        fl = 2.0 * B * oh * oh * co * ci * k * k
        flops += n * fl
        for p in tot: tot[p] += n * r[p][0]
        print(f'{n:>2} {ci:>5} {co:>5} {h:>4} {k} {s} | ' + ' | '.join(f'{r[p][0]:>8.1f} {r[p][1]:>5.0f}' for p in tot) + f' | {fl / 1e9:.0f}', flush=True)
    print('total ms:', {p: round(v / 1e3, 2) for p, v in tot.items()}, 'TFLOP per pass', round(flops / 1e12, 2),
          'avg TF', {p: round(flops / v / 1e6) for p, v in tot.items()})
