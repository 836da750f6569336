#!/usr/bin/env python3
"""Validation-side throughput on one MI355X: eval forward (bf16) + decode + NMS at B=32, 640x640, and the CPU oracle's
decode + NMS on a sample beside it.   tools/bench_eval.py [--batch 32] [--size 640] [--steps 10] [--bias -4.0]

A freshly initialised model leaves every one of the 25200 rows per image a candidate (sigmoid(~0) = 0.5 > 0.25: the worst
case for NMS).  --cand-frac F moves each level's objectness bias so that a fraction F of its rows passes the threshold
(0.02 -> ~500 candidates per image, like a trained model); --bias adds a plain shift."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastvision_amd
from fastvision_amd.classfication.models import darknet53
from fastvision_amd.detection.neck import yolov3neck
from fastvision_amd.detection.head import yolov3head
from fastvision_amd.detection.models import yolov3
from fastvision_amd.detection.tools import non_max_suppression_images
from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch


def timed(fn, steps, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32); ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--steps', type=int, default=10); ap.add_argument('--bias', type=float, default=0.0)
    ap.add_argument('--cpu-images', type=int, default=2); ap.add_argument('--cand-frac', type=float, default=0.0)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    fastvision_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(20220504)
    m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
               in_channels=3, num_classes=80, training=False).to(dev).eval()
    with torch.no_grad():
        for conv in m.head.heads:
            conv.bias.view(3, 85)[:, 4] += a.bias
    images, _ = synthetic_batch(a.batch, a.size)
    images = images.to(dev)
    if a.cand_frac > 0:
        # place the 0.25 objectness threshold at the (1 - cand_frac) quantile of each level's logits
        with torch.no_grad():
            head_out, _ = m(images)
            for conv, h in zip(m.head.heads, head_out):
                logit = h[..., 4].float().flatten()
                q = torch.quantile(logit[torch.randperm(logit.numel(), device=dev)[:1 << 20]], 1.0 - a.cand_frac)
                conv.bias.view(3, 85)[:, 4] += float(-1.0986123 - q)
    with torch.no_grad():
        fwd_ms, (head_out, results) = timed(lambda: m(images), a.steps)
        from fastvision_amd.detect_ops import yolo_decode
        dec_ms, _ = timed(lambda: yolo_decode(list(head_out), m._anchor_lists, m.backbone_strides_per_level), a.steps)
        nms_ms, dets = timed(lambda: non_max_suppression_images(results, 0.25, 0.45, 300), a.steps)
        all_ms, _ = timed(lambda: non_max_suppression_images(m(images)[1], 0.25, 0.45, 300), a.steps)
        from fastvision_amd.graphs import graphed_eval
        gm = graphed_eval(m, images)
        gfwd_ms, (ghead, gres) = timed(lambda: gm(images), a.steps)
        assert torch.equal(gres, results)
        gall_ms, _ = timed(lambda: non_max_suppression_images(gm(images)[1], 0.25, 0.45, 300), a.steps)
    cand = (results[..., 4] > 0.25).sum(1).float()
    out = {'workload': f'YOLOv3 eval {a.batch}x3x{a.size}x{a.size} bf16: forward + decode + NMS(0.25, 0.45, 300)', 'objectness_bias': a.bias, 'candidate_fraction_target': a.cand_frac,
           'candidates_per_image_mean': float(cand.mean()), 'candidates_per_image_max': float(cand.max()),
           'detections_per_image_mean': sum(len(d[0]) for d in dets) / a.batch,
           'forward_incl_decode_ms': round(fwd_ms, 3), 'decode_ms': round(dec_ms, 3), 'nms_ms': round(nms_ms, 3),
           'total_ms': round(all_ms, 3), 'images_per_sec': round(a.batch / all_ms * 1e3, 1),
           'hip_graph': {'forward_incl_decode_ms': round(gfwd_ms, 3), 'total_ms': round(gall_ms, 3), 'images_per_sec': round(a.batch / gall_ms * 1e3, 1)}}
    if a.cpu_images:
        from oracle import detect as D
        res_h = results[:a.cpu_images].float().cpu()
        heads_h = [h[:a.cpu_images].float().cpu().contiguous() for h in head_out]
        anchors = list(coco_anchors_px().view(3, 3, 2))
        t0 = time.perf_counter()
        D.decode_library(heads_h, m.backbone_strides_per_level, anchors)
        t1 = time.perf_counter()
        for b in range(a.cpu_images):
            D.nms_library(res_h[b], 0.25, 0.45, 300)
        t2 = time.perf_counter()
        out['cpu_oracle'] = {'images': a.cpu_images, 'decode_ms_per_image': round((t1 - t0) / a.cpu_images * 1e3, 2),
                             'nms_ms_per_image': round((t2 - t1) / a.cpu_images * 1e3, 2), 'kind': 'port (oracle/detect.py, numpy/torch CPU)'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
