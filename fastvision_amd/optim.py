"""FusedAdam: torch.optim.Adam semantics (the optimizer the reference builds, demos/yolov3_u/train.py:66-70) as
ONE multi-tensor HIP launch per step (``fva_adam_step``) instead of ~10 foreach kernels over 222 tensors.

Drop-in for ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (no amsgrad / maximize).  State lives in the
usual ``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}`` entries; ``state_dict()`` / ``load_state_dict()`` exchange
checkpoints with torch's Adam in both directions (torch stores ``step`` as a float tensor: it is coerced on use).

``capturable=True`` keeps the step count and the learning rate in device memory (``fva_adam_step_dev``) so that the
whole training step can be captured in a HIP graph (graphs.GraphedTrainStep): a replay advances the device counter
itself and reads the current learning rate from a device scalar that ``step()`` / the graph wrapper refresh whenever
``param_groups[i]['lr']`` changes (LR schedules keep working).
"""
import ctypes as C

import torch

from . import _lib
from .ops import _stream

__all__ = ['FusedAdam']


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale
        self.capturable = capturable
        self._tables = {}
        self._dev = {}            # group index -> {'state': double[3], 'lr': float[1], 'lr_host': float, 'pinned': int64[5n], 'table': int64[5n]}

    # ---- checkpoints -------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables.clear()                      # the moment tensors were replaced: pointer tables are stale
        for gi, d in self._dev.items():           # device step counters follow the loaded step
            ps = [p for p in self.param_groups[gi]['params'] if p in self.state and self.state[p]]
            if ps:
                d['state'][0] = float(self._step_of(ps[0]))
                d['lr_host'] = None

    def _step_of(self, p):
        st = self.state[p].get('step', 0)
        return int(st.item()) if torch.is_tensor(st) else int(st)

    # ---- pointer table -----------------------------------------------------------------------------------------------
    def _table(self, gi, group):
        """Device pointer table [4][n] + sizes, rebuilt only when a parameter, gradient or moment tensor moved."""
        ps = [p for p in group['params'] if p.grad is not None]
        for p in ps:
            st = self.state[p]
            if not st:
                st['step'] = 0
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise RuntimeError('FusedAdam needs contiguous fp32 parameters and gradients')
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr())
                    for p in ps)
        hit = self._tables.get(gi)
        if hit is None or hit[0] != key:
            n = len(ps)
            vals = [k[0] for k in key] + [k[1] for k in key] + [k[2] for k in key] + [k[3] for k in key] + [p.numel() for p in ps]
            dev = ps[0].device
            if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                # inside a graph capture nothing may be allocated on the host side: the staging buffers were made by an eager
                # (warm-up) step; the captured copy node re-reads the same pinned words on every replay
                d = self._dev.get(gi)
                if d is None or d.get('pinned') is None or d['pinned'].numel() != len(vals):
                    raise RuntimeError('FusedAdam: run at least one eager step() (capturable=True) and begin_capture() before capturing a graph')
                host, both = d['pinned'], d['table']
                host.copy_(torch.tensor(vals, dtype=torch.int64))
                both.copy_(host, non_blocking=True)
            else:
                # gradients are fresh tensors every step, so this small table is re-uploaded every step: pinned staging +
                # non_blocking keeps the upload asynchronous (a pageable .to(device) would stall the host on the stream)
                host = torch.tensor(vals, dtype=torch.int64)
                if dev.type == 'cuda':
                    host = host.pin_memory()
                both = host.to(dev, non_blocking=True)
                if self.capturable and dev.type == 'cuda':
                    d = self._dev_state(gi, group, dev)
                    if d.get('pinned') is None or d['pinned'].numel() != len(vals):
                        d['pinned'] = torch.empty(len(vals), dtype=torch.int64).pin_memory()
                        d['table'] = torch.empty(len(vals), dtype=torch.int64, device=dev)
            tab, sizes = both[:4 * n], both[4 * n:]
            hit = (key, tab, sizes, n, max(p.numel() for p in ps), host)
            self._tables[gi] = hit
        return ps, hit

    def begin_capture(self):
        """Fresh staging buffers for ONE graph capture (call right before ``torch.cuda.graph``, after a warm-up step).  The captured
        copy node re-reads the pinned words it was recorded with on every replay, so every graph needs words of its own: with one
        shared buffer a second capture on the same optimizer (another batch size, a re-capture) rewrote the pointers the first
        graph's Adam launch reads.  Returns the buffers: the graph's owner keeps them alive as long as the graph."""
        owned = []
        for gi, d in self._dev.items():
            if d.get('pinned') is not None:
                n = d['pinned'].numel()
                d['pinned'] = torch.empty(n, dtype=torch.int64).pin_memory()
                d['table'] = torch.empty(n, dtype=torch.int64, device=d['state'].device)
                owned.append((d['pinned'], d['table']))
        self._tables.clear()              # the capture-time step must write (and record the upload of) its own table
        return owned

    def _dev_state(self, gi, group, dev):
        d = self._dev.get(gi)
        if d is None:
            d = self._dev[gi] = {'state': torch.zeros(3, dtype=torch.float64, device=dev),
                                 'lr': torch.zeros(1, dtype=torch.float32, device=dev), 'lr_host': None}
            started = [p for p in group['params'] if p in self.state and self.state[p]]
            if started:
                d['state'][0] = float(self._step_of(started[0]))
        return d

    def sync_lr(self):
        """Refresh the device learning-rate scalars from param_groups (capturable mode; call between graph replays -- the graph
        wrapper does -- after an LR scheduler changed them).  One tiny fill kernel per changed group, no host sync."""
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            if d is not None and d['lr_host'] != group['lr']:
                d['lr'].fill_(float(group['lr']))
                d['lr_host'] = group['lr']

    def note_replayed_steps(self, k=1):
        """Host mirrors of the step count after ``k`` replays of a captured step (the device counter advanced by itself)."""
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                if st:
                    st['step'] = self._step_of(p) + k

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
            step = self._step_of(ps[0]) + 1
            for p in ps:
                self.state[p]['step'] = step
            b1, b2 = group['betas']
            if self.capturable:
                d = self._dev_state(gi, group, ps[0].device)
                if not torch.cuda.is_current_stream_capturing() and d['lr_host'] != group['lr']:
                    d['lr'].fill_(float(group['lr']))
                    d['lr_host'] = group['lr']
                _lib.call('fva_adam_step_dev', C.c_void_p(tab.data_ptr()), C.c_void_p(sizes.data_ptr()), n, mx, C.c_void_p(d['lr'].data_ptr()),
                          b1, b2, group['eps'], group['weight_decay'], C.c_void_p(d['state'].data_ptr()), self.grad_scale, _stream())
            else:
                _lib.call('fva_adam_step', C.c_void_p(tab.data_ptr()), C.c_void_p(sizes.data_ptr()), n, mx, group['lr'], b1, b2,
                          group['eps'], group['weight_decay'], step, self.grad_scale, _stream())
            torch.autograd.graph.increment_version(ps)      # the kernel wrote p in place: invalidate packed-weight caches
        return loss
