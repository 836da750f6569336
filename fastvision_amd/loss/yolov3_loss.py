"""Yolov3Loss on the MI355X kernels -- API mirror of the reference's loss/yolov3_loss.py.

``forward(y_pred, y_true)`` runs target assignment, the three loss terms AND their analytic backward in the
fused HIP kernels (``fva_yolov3_loss``): no autograd graph, no boolean-mask host sync, gradients (including the
path through the non-detached IoU objectness target) land in buffers that ``loss.backward()`` hands to the head.
``build_target`` returns the reference's structures (integer outputs bit-exact); it syncs once to size them.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..ops import _p, _stream, require_gpu

__all__ = ['Yolov3Loss']


def make_level(h, grad, anchors_px, stride):
    """fva_head_level for a head tensor [B,A,H,W,K] of any strides (and a gradient buffer with the same strides)."""
    lv = _lib.HeadLevel()
    lv.data, lv.grad = h.data_ptr(), (grad.data_ptr() if grad is not None else None)
    lv.sb, lv.sa, lv.sy, lv.sx, lv.sk = h.stride()
    lv.B, lv.A, lv.H, lv.W, lv.K = h.shape
    for i, (w, hh) in enumerate(anchors_px):
        lv.anchor_w[i], lv.anchor_h[i] = w, hh
    lv.stride = float(stride)
    return lv


def _as_f32(h):
    return h if h.dtype == torch.float32 else h.float()


def job_match_counts(targets, hs, anchors, strides, group):
    """[levels] int32 on the device: matches per level summed over every rank of ``group`` (three tiny matcher launches and one
    12-byte all-reduce, enqueued on the current stream -- no host sync)."""
    import torch.distributed as dist
    dev, T = hs[0].device, targets.shape[0]
    counts = torch.zeros(len(hs), dtype=torch.int32, device=dev)
    keep = []
    for lvl, h in enumerate(hs):
        A = len(anchors[lvl])
        cap = max(T * A, 1)
        i64 = torch.empty((5, cap), dtype=torch.int64, device=dev)
        f32 = torch.empty(cap * 6, dtype=torch.float32, device=dev)
        lv = make_level(h, None, anchors[lvl], strides[lvl])
        mo = _lib.MatchOut(counts[lvl:].data_ptr(), i64[0].data_ptr(), i64[1].data_ptr(), i64[2].data_ptr(), i64[3].data_ptr(),
                           i64[4].data_ptr(), f32.data_ptr(), f32[cap * 4:].data_ptr())
        keep.append((i64, f32))
        _lib.call('fva_yolov3_match', _p(targets) if T else C.c_void_p(0), T, C.byref(lv), C.byref(mo), _stream())
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


class _Yolov3LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, targets, anchors, strides, ratios, dp, *heads):
        hs = [_as_f32(h.detach()) for h in heads]
        need = any(ctx.needs_input_grad[5:])
        grads = []
        levels = (_lib.HeadLevel * len(hs))()
        for i, h in enumerate(hs):
            g = None
            if need:
                g = torch.zeros_like(h)                     # preserve_format keeps the (dense) strides of h
                if g.stride() != h.stride():
                    hs[i] = h = h.contiguous()
                    g = torch.zeros_like(h)
            grads.append(g)
            levels[i] = make_level(h, g, anchors[i], strides[i])
        T = targets.shape[0]
        lib = _lib.load()
        wsb = lib.fva_yolov3_loss_workspace(T, levels, len(hs))
        ws = torch.empty(wsb, dtype=torch.uint8, device=hs[0].device)
        out = torch.empty(4, dtype=torch.float32, device=hs[0].device)
        if dp is None:
            _lib.call('fva_yolov3_loss', _p(targets) if T else C.c_void_p(0), T, levels, len(hs), ratios[0], ratios[1], ratios[2],
                      _p(out), _p(ws), wsb, _stream())
        else:
            group, world = dp
            counts = job_match_counts(targets, hs, anchors, strides, group)
            _lib.call('fva_yolov3_loss_dp', _p(targets) if T else C.c_void_p(0), T, levels, len(hs), ratios[0], ratios[1], ratios[2],
                      _p(counts), world * hs[0].shape[0], _p(out), _p(ws), wsb, _stream())
        ctx.grads = grads
        ctx.dtypes = [h.dtype for h in heads]
        ctx.parts = out
        parts = out[1:4].clone()
        ctx.mark_non_differentiable(parts)
        return out[0:1].clone(), parts

    @staticmethod
    def backward(ctx, gout, _gparts):
        from ..ops import scale_loss_grads
        return (None, None, None, None, None, *scale_loss_grads(ctx, gout))


class Yolov3Loss(nn.Module):
    def __init__(self, model, iou_negative_thres, ratio_box, ratio_conf, ratio_cls):
        super().__init__()
        if isinstance(model, torch.nn.DataParallel):
            model = model.module
        self.anchor_levels = model.anchors_per_level
        self.backbone_stride_levels = model.backbone_strides_per_level
        self.levels = len(self.backbone_stride_levels)
        self.iou_negative_thres = iou_negative_thres        # stored, unused -- as in the reference (:20)
        self.ratio_box, self.ratio_conf, self.ratio_cls = ratio_box, ratio_conf, ratio_cls
        self._anchors_px = [[(float(w), float(h)) for w, h in a.reshape(-1, 2).tolist()] for a in self.anchor_levels]
        self.last_parts = None
        self._dp = None

    def data_parallel(self, group=None, enable=True):
        """One process per GPU in place of the reference's nn.DataParallel: evaluate this rank's SHARE of the loss the reference
        computes once on the batch gathered from all replicas (demos/yolov3_u/cfg/_fit.py:48-51, loss/yolov3_loss.py:69-71) --
        per-match means over the job's match count, "* bs" with the job's batch.  Summing the shares (and SUM-reducing the
        gradients: parallel.GradientReducer(average=False)) reproduces the reference's step; every rank must hold the same
        number of images.  Returns self."""
        import torch.distributed as dist
        self._dp = None
        if enable and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self._dp = (group, dist.get_world_size(group))
        return self

    def _targets(self, y_true, like):
        require_gpu(like, 'Yolov3Loss')
        return y_true.detach().to(device=like.device, dtype=torch.float32).contiguous()

    def forward(self, y_pred, y_true):
        tg = self._targets(y_true, y_pred[0])
        loss, parts = _Yolov3LossFn.apply(tg, self._anchors_px, self.backbone_stride_levels,
                                          (self.ratio_box, self.ratio_conf, self.ratio_cls), self._dp, *y_pred)
        self.last_parts = parts          # (box, conf, cls) means, device tensor, no sync
        return loss

    def build_target(self, y_pred, y_true):
        """Reference structures: per level (b, gxy[M,2], a) int64, cls int64, xywh [M,4], anchors [M,2]."""
        tg = self._targets(y_true, y_pred[0])
        dev, T = tg.device, tg.shape[0]
        locs, cats, xywhs, matched, pend = [], [], [], [], []
        for lvl, pre in enumerate(y_pred):
            A = len(self._anchors_px[lvl])
            cap = max(T * A, 1)
            i64 = lambda: torch.empty(cap, dtype=torch.int64, device=dev)
            count = torch.zeros(1, dtype=torch.int32, device=dev)
            b, gx, gy, a, cls = i64(), i64(), i64(), i64(), i64()
            xywh = torch.empty((cap, 4), dtype=torch.float32, device=dev)
            anc = torch.empty((cap, 2), dtype=torch.float32, device=dev)
            shape5 = (pre.shape[0], A, pre.shape[2], pre.shape[3], pre.shape[4])
            dummy = torch.empty(0, device=dev)
            lv = _lib.HeadLevel()
            lv.data = dummy.data_ptr() or 1
            lv.B, lv.A, lv.H, lv.W, lv.K = shape5
            for i, (w, h) in enumerate(self._anchors_px[lvl]):
                lv.anchor_w[i], lv.anchor_h[i] = w, h
            lv.stride = float(self.backbone_stride_levels[lvl])
            mo = _lib.MatchOut(count.data_ptr(), b.data_ptr(), gx.data_ptr(), gy.data_ptr(), a.data_ptr(), cls.data_ptr(),
                               xywh.data_ptr(), anc.data_ptr())
            _lib.call('fva_yolov3_match', _p(tg) if T else C.c_void_p(0), T, C.byref(lv), C.byref(mo), _stream())
            pend.append((count, b, gx, gy, a, cls, xywh, anc))
        for count, b, gx, gy, a, cls, xywh, anc in pend:
            m = int(count.item())
            locs.append((b[:m], torch.stack([gx[:m], gy[:m]], dim=1), a[:m]))
            cats.append(cls[:m])
            xywhs.append(xywh[:m])
            matched.append(anc[:m])
        return locs, cats, xywhs, matched
