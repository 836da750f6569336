"""ComputeLoss on the MI355X kernels -- API mirror of the reference's demos/yolov3_u/utils/lossv3.py.

``forward(predict_layers, target_all, model)``: best-anchor assignment on every level, BCE(xy) / MSE(wh) / BCE(cls),
IoU>0.5 ignore mask, masked objectness BCE, total = 2*xy + wh + cls + conf -- value and analytic gradients from the
fused HIP kernels (``fva_demo_loss``).  The reference prints its four partial losses every call (4 host syncs,
lossv3.py:108); here they stay on the device in ``self.last_parts`` and ``verbose=True`` restores the print.
"""
import torch
import torch.nn as nn

from .... import _lib
from ....ops import _p, _stream, require_gpu

__all__ = ['ComputeLoss']


class _DemoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, targets, anchors, *layers):
        need = any(ctx.needs_input_grad[2:])
        hs, grads = [], []
        levels = (_lib.HeadLevel * len(layers))()
        for i, raw in enumerate(layers):
            h = raw.detach()
            if h.dtype != torch.float32:
                h = h.float()
            B, ch, gh, gw = h.shape
            A = len(anchors[i])
            K = ch // A
            g = None
            if need:
                g = torch.zeros_like(h)
                if g.stride() != h.stride():
                    h = h.contiguous()
                    g = torch.zeros_like(h)
            sb, sc, sy, sx = h.stride()
            lv = levels[i]
            lv.data, lv.grad = h.data_ptr(), (g.data_ptr() if g is not None else None)
            lv.sb, lv.sa, lv.sy, lv.sx, lv.sk = sb, K * sc, sy, sx, sc      # channel c = a*K + k  (lossv3.py:42)
            lv.B, lv.A, lv.H, lv.W, lv.K = B, A, gh, gw, K
            for j, (w, hh) in enumerate(anchors[i]):
                lv.anchor_w[j], lv.anchor_h[j] = w, hh
            lv.stride = 1.0
            hs.append(h)
            grads.append(g)
        T = targets.shape[0]
        lib = _lib.load()
        wsb = lib.fva_demo_loss_workspace(T, levels, len(hs))
        ws = torch.empty(wsb, dtype=torch.uint8, device=hs[0].device)
        out = torch.empty(5, dtype=torch.float32, device=hs[0].device)
        _lib.call('fva_demo_loss', _p(targets), T, levels, len(hs), _p(out), _p(ws), wsb, _stream())
        ctx.grads, ctx.dtypes = grads, [l.dtype for l in layers]
        parts = out[1:5].clone()
        ctx.mark_non_differentiable(parts)
        return out[0:1].clone(), parts

    @staticmethod
    def backward(ctx, gout, _gparts):
        from ....ops import scale_loss_grads
        return (None, None, *scale_loss_grads(ctx, gout))


class ComputeLoss(nn.Module):
    def __init__(self, verbose=False):
        super().__init__()
        self.verbose = verbose
        self.last_parts = None
        self._anchors = None                     # (key, host copy): reading model.anchors is a device sync, and not allowed inside a graph capture

    def get_model(self, model):
        return model.module if hasattr(model, 'module') else model

    def forward(self, predict_layers, target_all, model):
        model = self.get_model(model)
        require_gpu(predict_layers[0], 'ComputeLoss')
        key = tuple((a.data_ptr(), a._version) for a in model.anchors)
        if self._anchors is None or self._anchors[0] != key:
            self._anchors = (key, [[(float(w), float(h)) for w, h in a.reshape(-1, 2).tolist()] for a in model.anchors])   # feature scale
        anchors = self._anchors[1]
        if target_all.shape[0] == 0:
            raise RuntimeError('ComputeLoss needs at least one target (the reference fails on an empty image too, lossv3.py:94)')
        tg = target_all.detach().to(device=predict_layers[0].device, dtype=torch.float32).contiguous()
        loss, parts = _DemoLossFn.apply(tg, anchors, *predict_layers)
        self.last_parts = parts                  # (xy, wh, cls, conf) as the reference prints them
        if self.verbose:
            print(*parts.tolist())
        return loss
