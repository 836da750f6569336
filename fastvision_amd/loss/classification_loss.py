"""Classification losses -- API mirror of the reference's loss/classification_loss.py.

These two classes are the thin, standalone API surface (``CrossEntropyLoss`` is the CPU plumbing loss of
BASELINE config 1).  On the accelerated path ``Yolov3Loss`` does NOT call them: its class / objectness BCE
terms and their gradients are computed inside the fused HIP loss kernels (csrc/loss.hip), which restate
``BiCrossEntropyLoss`` exactly (1e-8 inside both logs, sum / numel).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['one_hot', 'CrossEntropyLoss', 'BiCrossEntropyLoss']


def one_hot(y, num_classes):
    """datasets/common/id_2_onehot.py:10-15 (torch branch)."""
    col = y.view(-1, 1).long()
    return torch.zeros((col.size(0), num_classes)).to(y).scatter_(1, col, 1)


class CrossEntropyLoss(nn.Module):
    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, y_pre, y_true, weights=None):
        target = one_hot(y_true, y_pre.size(-1)).float()
        loss = -torch.sum(target * F.log_softmax(y_pre, dim=-1), dim=1)
        if weights is not None:
            loss = loss * weights
        return torch.mean(loss) if self.reduction == 'mean' else torch.sum(loss)


class _BceFn(torch.autograd.Function):
    """fva_bce_loss: value and dl/dy in one launch (+ a one-block finish); backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, y, label, dense, weights, C, already_sigmoid, mean):
        import ctypes as Ct
        from .. import _lib
        from ..ops import _p, _stream, require_gpu
        require_gpu(y, 'BiCrossEntropyLoss')
        yf = y.detach().float().contiguous().view(-1)
        numel = yf.numel()
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(yf) if need else None
        out = torch.empty(1, dtype=torch.float32, device=y.device)
        ws = torch.empty(1024, dtype=torch.float32, device=y.device)
        w = weights.detach().float().contiguous().view(-1) if weights is not None else None
        if w is not None and w.numel() not in (1, numel):
            w = w.expand(numel // w.numel(), w.numel()).contiguous().view(-1) if numel % w.numel() == 0 else w
        lab = label.detach().long().contiguous().view(-1) if label is not None else None
        den = dense.detach().float().contiguous().view(-1) if dense is not None else None
        _lib.call('fva_bce_loss', _p(yf), _p(lab), _p(den), _p(w), w.numel() if w is not None else 0, numel, C, 1 if already_sigmoid else 0,
                  1 if mean else 0, _p(out), _p(grad), _p(ws), _stream())
        ctx.grad, ctx.shape, ctx.dtype, ctx.scale = grad, y.shape, y.dtype, (1.0 / numel if mean else 1.0)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        g = (ctx.grad * (gout * ctx.scale)).view(ctx.shape)
        return (g if g.dtype == ctx.dtype else g.to(ctx.dtype)), None, None, None, None, None, None


class BiCrossEntropyLoss(nn.Module):
    """loss/classification_loss.py:36-65 on the device: ``forward(y_pre, y_true, already_sigmoid=False, weights=None)``.  A last
    dimension > 1 means class scores with integer labels (one-hot targets); a last dimension of 1 a dense float target.  1e-8 sits
    inside both logarithms; ``mean`` divides the summed element losses by the element count.  Device tensors only (the reference's
    CPU use of this class, config 1's plumbing, goes through CrossEntropyLoss above)."""

    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, y_pre, y_true, already_sigmoid=False, weights=None):
        C = y_pre.size(-1)
        if C > 1:
            return _BceFn.apply(y_pre, y_true, None, weights, C, already_sigmoid, self.reduction == 'mean')
        return _BceFn.apply(y_pre, None, y_true, weights, 1, already_sigmoid, self.reduction == 'mean')
