#!/usr/bin/env python3
"""Input-side throughput on one MI355X: 32 decoded images of mixed sizes -> [32,3,640,640] fp32 (library pipeline: resize,
letterbox 114, flips, ImageNet normalisation, CHW), with the CPU oracle beside it.  tools/bench_pipeline.py [--steps 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastvision_amd import _lib
from fastvision_amd.datasets import BaseDataset
from fastvision_amd.datasets.detection_dataloader import IMAGENET_MEAN, IMAGENET_STD, PAD_VALUE
from fastvision_amd.pipeline_ops import _job_tables, value_table
from fastvision_amd.ops import _p, _stream
from oracle import pipeline as P
from oracle.make_golden import synth_image, synth_boxes


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=20); ap.add_argument('--cpu-images', type=int, default=8)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    g = np.random.default_rng(1)
    sizes = [(480, 640), (427, 640), (640, 480), (375, 500), (1280, 1280), (640, 640), (333, 500), (720, 1280)] * 4
    ds = BaseDataset([], 640, 200)
    raw = [(synth_image(g, h, w), synth_boxes(g, 3, h, w), bool(k & 1), bool(k & 2)) for k, (h, w) in enumerate(sizes)]
    batch = []
    for rgb, ann, hf, vf in raw:
        lab = ds.labels_for(ann, rgb.shape[:2], hf, vf)
        t = torch.zeros(len(lab), 6); t[:, 1:] = torch.from_numpy(lab)
        batch.append((rgb, t, (hf, vf)))
    host = ds.collate_host(batch)
    buf, offsets, shapes, flips, labels = host
    # (1) whole device half incl. the pinned upload of the decoded bytes
    for _ in range(3): ds.to_device(host, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): ds.to_device(host, dev)
    torch.cuda.synchronize(); incl = (time.perf_counter() - t0) / a.steps * 1e3
    # (2) the kernel alone, inputs resident
    src = buf.to(dev)
    tab, start = _job_tables(ds.jobs_for(shapes, flips), offsets, shapes, len(shapes), dev)
    lut = torch.from_numpy(value_table(IMAGENET_MEAN, IMAGENET_STD)).to(dev)
    out = torch.empty(len(shapes), 3, 640, 640, device=dev)
    call = lambda: _lib.call('fva_paste_resize_normalize', _p(src), _p(tab), _p(start), len(shapes), 640, 640, PAD_VALUE, _p(lut), _p(out), _stream())
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps): call()
    e1.record(); torch.cuda.synchronize()
    kern = e0.elapsed_time(e1) / a.steps
    nbytes = src.numel() + out.numel() * 4
    # (3) CPU oracle on a sample
    t0 = time.perf_counter()
    for rgb, ann, hf, vf in raw[:a.cpu_images]:
        P.library_sample(rgb, ann, 640, hf, vf)
    cpu = (time.perf_counter() - t0) / a.cpu_images * 1e3
    print(json.dumps({'workload': '32 decoded RGB images (375x500 ... 1280x1280) -> [32,3,640,640] fp32: resize + letterbox + flips + normalise + CHW',
                      'source_MB': round(src.numel() / 1e6, 1), 'output_MB': round(out.numel() * 4 / 1e6, 1),
                      'kernel_ms': round(kern, 4), 'kernel_GBps_algorithmic': round(nbytes / kern / 1e6, 1),
                      'images_per_sec_resident': round(32 / kern * 1e3), 'batch_ms_incl_pinned_upload_and_host': round(incl, 3),
                      'images_per_sec_incl_upload': round(32 / incl * 1e3),
                      'cpu_oracle': {'ms_per_image': round(cpu, 2), 'images_per_sec_1_core': round(1e3 / cpu, 1), 'images': a.cpu_images,
                                     'kind': 'port (oracle/pipeline.py, numpy)'}}))


if __name__ == '__main__':
    main()
