#!/usr/bin/env python3
"""Timings of the two-stage-head operators that exist so far at the reference's training shape: a VGG16
stride-16 map of an 800x608 image (38 x 50 cells, 512 channels, 9 anchors), 4 images, rpn.py / fast.py defaults.
Prints one JSON object.  usage: python tools/bench_twostage.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fastvision_amd import ops
from fastvision_amd.roi_ops import roi_align
from fastvision_amd.rpn_ops import fast_select_samples, filter_proposals, rpn_match

DEV = 'cuda:0'


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    g = torch.Generator().manual_seed(0)
    B, H, W, A, C, T = 4, 38, 50, 9, 512, 40
    base = torch.tensor([[11.3, 5.7], [22.6, 11.3], [45.3, 22.6], [8, 8], [16, 16], [32, 32], [5.7, 11.3], [11.3, 22.6], [22.6, 45.3]])
    cls = (torch.randn(B, H, W, A, 2, generator=g) * 2).to(DEV)
    d = (torch.randn(B, H, W, A, 4, generator=g) * 0.3).to(DEV)
    tb = torch.sort(torch.randint(0, B, (T,), generator=g))[0].float()
    wh = torch.exp(np.log(0.05) + (np.log(0.6) - np.log(0.05)) * torch.rand(T, 2, generator=g))
    xy = wh / 2 + (1 - wh) * torch.rand(T, 2, generator=g)
    targets = torch.cat([tb[:, None], torch.randint(0, 20, (T, 1), generator=g).float(), xy, wh], 1).to(DEV)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    anchors = torch.cat([torch.stack([xs, ys], -1).float().view(H, W, 1, 2).expand(H, W, A, 2), base.view(1, 1, A, 2).expand(H, W, A, 2)], -1).to(DEV)
    out = {'shape': {'B': B, 'H': H, 'W': W, 'A': A, 'C': C, 'boxes': T}}
    props = filter_proposals(cls, d, base, 2000, 2000, 0.7)
    out['rpn_filter_proposals_us'] = round(timed(lambda: filter_proposals(cls, d, base, 2000, 2000, 0.7)), 1)
    out['proposals_per_image'] = [int(p.size(0)) for p in props]
    out['rpn_match_us'] = round(timed(lambda: rpn_match(anchors, targets, B, H, W)), 1)
    tg_cells = targets * torch.tensor([1, 1, W, H, W, H], device=DEV)
    out['fast_select_samples_us'] = round(timed(lambda: fast_select_samples(props, tg_cells)), 1)
    pos, neg = fast_select_samples(props, tg_cells)
    rois = torch.cat([pos[:, :5], neg], 0)
    rois_xyxy = torch.cat([rois[:, :1], rois[:, 1:3] - rois[:, 3:5] / 2, rois[:, 1:3] + rois[:, 3:5] / 2], 1)
    buf, feat = ops.halo_alloc(B, C, H, W, torch.float32, torch.device(DEV), 1)
    buf.normal_()
    out['roi_align_fwd_us'] = round(timed(lambda: roi_align(feat, rois_xyxy, (7, 7))), 1)
    out['roi_count'] = int(rois.size(0))
    f = feat.detach().clone().requires_grad_(True)

    def fb():
        f.grad = None
        roi_align(f, rois_xyxy, (7, 7)).sum().backward()
    out['roi_align_fwd_bwd_us'] = round(timed(fb), 1)
    out['note'] = 'host-inclusive wall time per call (each call reads counts back where the reference does); torch indexing glue included'
    print(json.dumps(out))


if __name__ == '__main__':
    main()
