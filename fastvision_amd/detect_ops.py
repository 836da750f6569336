"""Device-side validation ops (scope row f-2): eval decode and non-maximum suppression through the C ABI
(``fva_yolo_decode``, ``fva_nms_candidates``, ``fva_nms_select`` -- fastvision_amd/csrc/detect.hip).

The reference does this with torch elementwise ops, boolean-mask indexing and torchvision.ops.nms
(detection/models/yolov3.py:35-53, detection/tools/NMS.py, demos/yolov3_u/utils/nms.py, inference.py:58-120).
"""
import ctypes as C

import torch

from . import _lib
from .ops import _p, _stream, require_gpu

__all__ = ['yolo_decode', 'nms_batch', 'NMS_LIBRARY', 'NMS_DEMO', 'NMS_DEMO_BATCH']

# (box_mode, score_mode, rethreshold, class_gap, max_nms) of the three reference wrappers
NMS_LIBRARY = dict(box_mode=0, score_mode=0, rethreshold=0, class_gap=0.0, max_nms=0)        # detection/tools/NMS.py
NMS_DEMO = dict(box_mode=1, score_mode=1, rethreshold=0, class_gap=4096.0, max_nms=30000)     # utils/nms.py:5-52
NMS_DEMO_BATCH = dict(box_mode=0, score_mode=0, rethreshold=1, class_gap=4096.0, max_nms=30000)  # utils/nms.py:54-98

MASK_BYTES_PER_CALL = 2 << 30   # images are processed in groups whose suppression bitmaps fit this much HBM


def _level(h, anchors, stride):
    lv = _lib.HeadLevel()
    lv.data, lv.grad = h.data_ptr(), None
    lv.sb, lv.sa, lv.sy, lv.sx, lv.sk = h.stride()
    lv.B, lv.A, lv.H, lv.W, lv.K = h.shape
    for i, (w, hh) in enumerate(anchors):
        lv.anchor_w[i], lv.anchor_h[i] = float(w), float(hh)
    lv.stride = float(stride)
    return lv


def yolo_decode(heads, anchors, strides, variant=0, letterbox=None):
    """heads: fp32 CUDA tensors [B,A,H,W,K] (any strides) per level; anchors: per level a sequence of (w, h);
    returns [B, sum A*H*W, K] fp32.  variant / letterbox: see fva_yolo_decode in include/fastvision_amd.h."""
    require_gpu(heads[0], 'yolo_decode')
    heads = [h if h.dtype == torch.float32 else h.float() for h in heads]
    levels = (_lib.HeadLevel * len(heads))()
    rows = 0
    for i, h in enumerate(heads):
        levels[i] = _level(h, anchors[i], strides[i])
        rows += h.shape[1] * h.shape[2] * h.shape[3]
    B, K = heads[0].shape[0], heads[0].shape[4]
    out = torch.empty(B, rows, K, dtype=torch.float32, device=heads[0].device)
    lb = None
    if letterbox is not None:
        lb = C.byref(_lib.Letterbox(*[float(v) for v in letterbox]))
    _lib.call('fva_yolo_decode', levels, len(heads), variant, lb, _p(out), rows, _stream())
    return out


def nms_batch(pred, conf_thres, iou_thres, max_det, mode):
    """pred [B,R,K] fp32 CUDA (decoded rows).  Returns per image (det [n,6] = x1,y1,x2,y2,score,category; rows [n] =
    source row of each detection), highest score first.  Two host read-backs per call (candidate counts, kept counts):
    the reference's boolean-mask indexing synchronises at the same two points per IMAGE."""
    require_gpu(pred, 'nms_batch')
    assert pred.dim() == 3 and pred.dtype == torch.float32
    pred = pred.contiguous()
    B, R, K = pred.shape
    dev = pred.device
    prm = _lib.NmsParams(mode['box_mode'], mode['score_mode'], mode['rethreshold'], int(max_det), mode['max_nms'],
                         float(conf_thres), float(iou_thres), mode['class_gap'])
    lib = _lib.load()
    cb = lib.fva_nms_candidates_workspace(B, R)
    cand = torch.empty(cb, dtype=torch.uint8, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.call('fva_nms_candidates', _p(pred), B, R, K, C.byref(prm), _p(cand), cb, _p(counts), _stream())
    counts_h = counts.cpu()
    nmax = int(counts_h.max())
    empty = (torch.zeros(0, 6, device=dev), torch.zeros(0, dtype=torch.int64, device=dev))
    if nmax == 0:
        return [empty] * B
    out = torch.empty(B, max_det, 6, dtype=torch.float32, device=dev)
    rows = torch.empty(B, max_det, dtype=torch.int32, device=dev)
    kept = torch.empty(B, dtype=torch.int32, device=dev)
    wb = lib.fva_nms_select_workspace(B, nmax)
    if wb <= MASK_BYTES_PER_CALL or B == 1:
        ws = torch.empty(wb, dtype=torch.uint8, device=dev)
        _lib.call('fva_nms_select', _p(cand), _p(counts), B, R, nmax, C.byref(prm), _p(ws), wb, _p(out), _p(rows), _p(kept), _stream())
    else:
        # large candidate sets (an untrained model passes every row): one image at a time, each with its own nmax
        # (stage 1 is re-run on the image's slice so that its candidate buffer stands alone)
        for b in range(B):
            n_b = int(counts_h[b])
            if n_b == 0:
                kept[b] = 0
                continue
            cb1 = lib.fva_nms_candidates_workspace(1, R)
            cand1 = torch.empty(cb1, dtype=torch.uint8, device=dev)
            cnt1 = torch.empty(1, dtype=torch.int32, device=dev)
            _lib.call('fva_nms_candidates', _p(pred[b]), 1, R, K, C.byref(prm), _p(cand1), cb1, _p(cnt1), _stream())
            wb1 = lib.fva_nms_select_workspace(1, n_b)
            ws = torch.empty(wb1, dtype=torch.uint8, device=dev)
            _lib.call('fva_nms_select', _p(cand1), _p(cnt1), 1, R, n_b, C.byref(prm), _p(ws), wb1, _p(out[b]), _p(rows[b]),
                      _p(kept[b:b + 1]), _stream())
    kept_h = kept.cpu().tolist()
    return [(out[b, :kept_h[b]], rows[b, :kept_h[b]].long()) if kept_h[b] else empty for b in range(B)]
