"""FusedAdam: torch.optim.Adam semantics (the optimizer the reference builds, demos/yolov3_u/train.py:66-70) as
ONE multi-tensor HIP launch per step (``fva_adam_step``) instead of ~10 foreach kernels over 222 tensors.

Drop-in for ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (no amsgrad / maximize).  State lives in the
usual ``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}`` entries, so ``state_dict()`` round-trips with torch's Adam.
"""
import ctypes as C

import torch

from . import _lib
from .ops import _stream

__all__ = ['FusedAdam']


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale
        self._tables = {}

    def _table(self, gi, group):
        """Device pointer table [4][n] + sizes, rebuilt only when a gradient tensor moved."""
        ps = [p for p in group['params'] if p.grad is not None]
        for p in ps:
            st = self.state[p]
            if not st:
                st['step'] = 0
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise RuntimeError('FusedAdam needs contiguous fp32 parameters and gradients')
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        hit = self._tables.get(gi)
        if hit is None or hit[0] != key:
            n = len(ps)
            ptrs = [p.data_ptr() for p in ps] + [p.grad.data_ptr() for p in ps] + \
                   [self.state[p]['exp_avg'].data_ptr() for p in ps] + [self.state[p]['exp_avg_sq'].data_ptr() for p in ps]
            dev = ps[0].device
            # gradients are fresh tensors every step, so this small table is re-uploaded every step: pinned staging +
            # non_blocking keeps the upload asynchronous (a pageable .to(device) would stall the host on the stream)
            host = torch.tensor(ptrs + [p.numel() for p in ps], dtype=torch.int64)
            if dev.type == 'cuda':
                host = host.pin_memory()
            both = host.to(dev, non_blocking=True)
            tab, sizes = both[:4 * n], both[4 * n:]
            hit = (key, tab, sizes, n, max(p.numel() for p in ps), host)
            self._tables[gi] = hit
        return ps, hit

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .ops import join_side_stream
        join_side_stream()          # weight gradients may still be in flight on the library's side stream
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group['params']):
                continue
            ps, (_, tab, sizes, n, mx, _keep) = self._table(gi, group)
            if not ps[0].is_cuda:
                raise RuntimeError('FusedAdam: parameters must live on the GPU (no CPU path)')
            step = self.state[ps[0]]['step'] + 1
            for p in ps:
                self.state[p]['step'] = step
            b1, b2 = group['betas']
            _lib.call('fva_adam_step', C.c_void_p(tab.data_ptr()), C.c_void_p(sizes.data_ptr()), n, mx, group['lr'], b1, b2,
                      group['eps'], group['weight_decay'], step, self.grad_scale, _stream())
            torch.autograd.graph.increment_version(ps)      # the kernel wrote p in place: invalidate packed-weight caches
        return loss
