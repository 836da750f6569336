"""Data parallelism for the hot path: one process per GPU, gradients all-reduced over RCCL/xGMI in buckets that are launched
while the backward pass is still running.

The reference only has single-process ``nn.DataParallel`` (demos/yolov3_u/train.py:85), which re-broadcasts the 248 MB of
parameters and gathers the outputs to GPU 0 every step, where ONE loss is evaluated on the gathered batch
(demos/yolov3_u/cfg/_fit.py:48-51).  This replacement keeps those semantics -- replicated parameters, per-GPU BatchNorm statistics
(no SyncBN), and the gradient of the loss over the WHOLE batch -- and changes the mechanics: each rank evaluates the loss on its own
32 images; the library's ``Yolov3Loss`` (means x batch size, loss/yolov3_loss.py:69-71) normalises its per-match sums by the match
counts of the whole job (``Yolov3Loss.data_parallel()``, a 12-byte all-reduce) and the ranks' gradients are SUMMED
(``GradientReducer(average=False)``); the demo's ``ComputeLoss`` is a plain mean, so its gradients are AVERAGED
(``average=True``).  Checked against the CPU oracle evaluated on the gathered batch
(tests/test_gpu_model.py::test_data_parallel_reproduces_the_reference_dataparallel_step).

xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound by one link: buckets are kept large (16 MiB on the wire in
bench.py, ~16 buckets for the 61.9 M parameters) to amortise latency, and are filled in reverse registration order (head -> neck ->
backbone), i.e. in the order backward produces gradients, so the first buckets fly while the backbone is still in backward.  The
wire dtype is the parameters' fp32 unless ``bucket_dtype=torch.bfloat16`` is asked for (half the bytes; the reduced buckets are
widened back into the fp32 ``.grad`` views, the optimizer never sees bf16).  ``torch.distributed`` is the transport (backend "nccl"
is RCCL on ROCm; "gloo" for CPU tests).

Streams (audited in round 3): weight gradients are computed on the library's low-priority side stream; a
bucket's gather/narrow launch is enqueued THERE, behind the gradients it reads; its all-reduce is enqueued from the MAIN stream
one bucket later, after ``wait_event(filled)`` -- so the only cross-stream barrier the process group's stream ever carries waits
for the main stream's recent past, never for the lagging side stream.
"""
import contextlib

import torch
import torch.distributed as dist

__all__ = ['init_from_env', 'GradientReducer', 'broadcast_parameters', 'shard_targets', 'refuse_dataparallel_replica']


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* / LOCAL_RANK (torchrun contract).
    Returns (rank, world, local_rank).  A no-op (0, 1, 0) when WORLD_SIZE is absent or 1."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('FVA_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)     # several ranks may share a GPU in tests (gloo only)
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters and buffers (DataParallel replicates from GPU 0).
    One flat broadcast per (dtype, device) -- two collectives for YOLOv3 (fp32 tensors, the int64 batch counters) instead of
    one per tensor (438 + 72)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    groups = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault((t.dtype, t.device), []).append(t.data)
    for ts in groups.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def refuse_dataparallel_replica(module):
    """Called at the top of the detection models' forward.  The reference wraps its model in ``nn.DataParallel``
    (demos/yolov3_u/train.py:85, generate/template-yolov3/train.py:81-83).  With ONE visible device that wrapper calls the module
    directly and everything works; with more it replicates the module onto the other devices every step and runs the replicas from
    threads -- but this package's kernels, side stream, packed-weight caches and BatchNorm buffers are per process and per device (one
    process per GPU is the model, SURVEY 8b: "refuse DP-wrap and provide its own DP").  A replica (``_is_replica``, set by
    nn.Module._replicate_for_data_parallel) therefore refuses loudly instead of computing on state it does not own."""
    if getattr(module, '_is_replica', False):
        raise RuntimeError(
            'fastvision_amd: nn.DataParallel over more than one device is not supported -- the HIP kernels, the weight-gradient side '
            'stream and the packed-weight caches are per process and per device.  Run one process per GPU instead (python -m '
            'torch.distributed.run --nproc-per-node N ...; fastvision_amd.parallel.init_from_env() + GradientReducer; for the library '
            'loss also Yolov3Loss.data_parallel()), or make one device visible (HIP_VISIBLE_DEVICES=0): INTEGRATION.md section 1.')


def shard_targets(targets, rank, per_rank_batch):
    """Rows of a global [T,6] target table that belong to this rank's images, re-based to local image indices."""
    lo = rank * per_rank_batch
    keep = (targets[:, 0] >= lo) & (targets[:, 0] < lo + per_rank_batch)
    out = targets[keep].clone()
    out[:, 0] -= lo
    return out


class GradientReducer:
    """Bucketed, overlapped gradient averaging.

        reducer = GradientReducer(model.parameters())
        loss.backward()          # hooks copy each finished gradient into its bucket; full buckets start reducing
        reducer.finish()         # wait for the collectives; p.grad are now views of the averaged buckets
        optimizer.step()

    Buckets are launched strictly in index order so that every rank issues the same sequence of collectives.

    ``average``: True divides the summed gradient by the world size (right for a loss that is a MEAN over the batch, the demo's
    ComputeLoss); False leaves the SUM (right for the library's Yolov3Loss, which multiplies its means by the batch size,
    loss/yolov3_loss.py:69-71: under the reference's nn.DataParallel that loss sees the gathered batch of N*b images, whose
    gradient is the sum of the per-rank ``* b`` losses' gradients -- an average would be 1/N of it and make Adam's folded-in weight
    decay and eps N times stronger; ``Yolov3Loss.data_parallel()`` supplies the job-wide match counts).
    ``bucket_dtype``: dtype the gradients travel in (default: the parameters' own, fp32 -- what the reference's DataParallel
    reduces).  torch.bfloat16 halves the bytes on the xGMI links (124 MB instead of 248 MB per step for YOLOv3); finish() widens the
    reduced wire buffers back into the fp32 buckets that ``p.grad`` views, so the optimizer always reads fp32.
    ``late``: how many buckets behind its fill a bucket's all-reduce is issued from the main stream (default 1).  The main stream
    waits for the fill event, which is recorded on the low-priority side stream; issued one bucket late that event is normally long
    past.  Where the side stream lags further (its own hardware queue: DESIGN.md section 6, wait (4)) the main stream stalls there --
    ``stats()['main_stream_wait_ms']`` measures exactly that, per step, from timing events around the wait -- and ``late=2`` trades it
    for a longer tail in ``finish()``.
    ``stats()`` reports what the last step did (backend, world size, buckets launched before backward ended, bytes on the wire, and
    how long the main stream sat waiting for fill events).
    """

    def __init__(self, params, bucket_bytes=32 << 20, group=None, average=True, bucket_dtype=None, world=None, late=1):
        if late not in (1, 2):
            raise ValueError('GradientReducer: late must be 1 or 2')
        self.late = late
        self.group, self.average, self.bucket_dtype = group, average, bucket_dtype
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)   # world: tests only
        if self.world == 1:
            self.bucket_dtype = None                      # nothing travels
        self.params = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))               # backward produces gradients roughly in reverse order
        self.buckets, self.where = [], {}
        cur, cur_bytes = [], 0
        for p in order:
            nbytes = p.numel() * (p.element_size() if self.bucket_dtype is None else torch.empty(0, dtype=self.bucket_dtype).element_size())
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self.pending = [0] * len(self.buckets)
        self.handles = [None] * len(self.buckets)
        self.next_launch = 0
        self.filled = 0               # GPU buckets: filled up to here (their collectives follow one bucket later)
        self._gpu = {}
        self.hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.last = {'launched_in_backward': 0, 'collectives': 0, 'waits': []}
        self._waits = []              # (event before the main stream's wait, the fill event): this step's, timed in stats()
        self.reset()

    def stats(self):
        """What the reducer did in the step that finish() last closed: for bench.py's JSON line, so that a driver can check that the
        backend really saw N ranks and that buckets overlapped the backward pass."""
        ini = dist.is_available() and dist.is_initialized()
        return {'backend': dist.get_backend(self.group) if ini else None,
                'world_size_seen_by_backend': dist.get_world_size(self.group) if ini else 1,
                'buckets': len(self.buckets),
                'buckets_launched_before_backward_ended': self.last['launched_in_backward'],
                'collectives_per_step': self.last['collectives'],
                'wire_dtype': str(self.buckets[0][3].dtype).replace('torch.', '') if self.buckets else None,
                'wire_bytes_per_step': sum(b[3].numel() * b[3].element_size() for b in self.buckets) if self.world > 1 else 0,
                'reduce_op': 'avg' if self.average else 'sum', 'late': self.late,
                'main_stream_wait_ms': self._wait_ms(self.last.get('waits', []))}

    @staticmethod
    def _wait_ms(pairs):
        """Time the main stream spent in wait_event(filled) over one step: for every bucket, fill time minus the time the main
        stream reached the wait, where positive (both are timing events; synchronises on them -- call outside a timed region)."""
        total = 0.0
        for pre, filled in pairs:
            try:
                pre.synchronize()
                filled.synchronize()
                total += max(0.0, pre.elapsed_time(filled))
            except RuntimeError:
                return None
        return round(total, 3)

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        # the buffer that travels: the fp32 bucket itself, or a narrower copy of it (p.grad must keep the parameter's dtype, so
        # the optimizer-facing views stay fp32 and finish() widens the reduced wire buffer back into them: 0.1 ms for 62 M values)
        wire = flat if self.bucket_dtype in (None, flat.dtype) else torch.zeros(n, dtype=self.bucket_dtype, device=flat.device)
        views, wviews, off = [], [], 0
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            wviews.append(wire[off:off + p.numel()].view_as(p))
            self.where[p] = (len(self.buckets), len(views) - 1)
            off += p.numel()
        self.buckets.append((flat, plist, views, wire, wviews))

    def reset(self):
        self.pending = [len(b[1]) for b in self.buckets]
        self.handles = [None] * len(self.buckets)
        self.next_launch = 0
        self.filled = 0

    def _side(self, p):
        """Context in which a gradient of ``p`` may be read: weight gradients are computed on the library's low-priority
        side stream (ops.wgrad_stream), so the bucket copies and the all-reduce launches are enqueued there too, behind
        them -- the backward pass on the main stream is not stalled.  finish() joins."""
        if p.is_cuda:
            from .ops import fork_side_stream
            st = fork_side_stream()
            if st is not None:
                return torch.cuda.stream(st)
        return contextlib.nullcontext()

    # ---- GPU buckets: one gather launch per bucket ----------------------------------------------------------------------------
    # Per parameter the hook only counts.  When a bucket is complete, ONE fva_gather_cast launch copies (and narrows) the gradients
    # of all its parameters into the wire buffer, from a pointer table that is re-uploaded only when a gradient moved; then the
    # parameters' .grad become views of the bucket; the all-reduce follows from the main stream (_launch_gpu).  Before: a torch copy,
    # a stream context and a fork per parameter in the hooks.
    def _gpu_state(self, bi):
        st = self._gpu.get(bi)
        if st is None:
            flat, plist, views, wire, wviews = self.buckets[bi]
            n = len(plist)
            offs, off = [], 0
            for p in plist:
                offs.append(off)
                off += p.numel()
            static = [p.numel() for p in plist] + offs
            st = self._gpu[bi] = {'n': n, 'key': None, 'flip': 0, 'max': max(p.numel() for p in plist),
                                  'pinned': [torch.zeros(3 * n, dtype=torch.int64).pin_memory() for _ in range(2)],
                                  'event': [torch.cuda.Event(), torch.cuda.Event()], 'used': [False, False],
                                  'filled': torch.cuda.Event(enable_timing=True), 'pre': torch.cuda.Event(enable_timing=True),
                                  'table': torch.zeros(3 * n, dtype=torch.int64, device=flat.device)}
            for h in st['pinned']:
                h[n:] = torch.tensor(static, dtype=torch.int64)
        return st

    def _fill_gpu(self, bi):
        import ctypes as C
        from . import _lib
        from .ops import _code, fork_side_stream, hold_for_side_stream
        flat, plist, views, wire, wviews = self.buckets[bi]
        st = self._gpu_state(bi)
        ptrs, missing = [], []
        for p, v, wv in zip(plist, views, wviews):
            g = p.grad
            if g is None:
                ptrs.append(0)
                missing.append(wv)
            elif g.data_ptr() == v.data_ptr():
                ptrs.append(0 if wire is flat else v.data_ptr())          # already in the bucket: only the wire copy is missing
            else:
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise RuntimeError('GradientReducer: gradients must be contiguous fp32 tensors')
                ptrs.append(g.data_ptr())
        side = fork_side_stream()
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            cur = torch.cuda.current_stream(flat.device)
            key = tuple(ptrs)
            if key != st['key']:
                k = st['flip']
                if st['used'][k]:
                    st['event'][k].synchronize()                          # the upload that last read this staging buffer has run
                st['pinned'][k][:st['n']] = torch.tensor(ptrs, dtype=torch.int64)
                st['table'].copy_(st['pinned'][k], non_blocking=True)
                st['event'][k].record(cur)
                st['used'][k], st['flip'], st['key'] = True, k ^ 1, key
            for wv in missing:
                wv.zero_()
            if any(ptrs):
                _lib.call('fva_gather_cast', C.c_void_p(st['table'].data_ptr()), st['n'], st['max'], C.c_void_p(wire.data_ptr()),
                          _code(wire.dtype), C.c_void_p(cur.cuda_stream))
            for p, v in zip(plist, views):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    if p.grad is not None:
                        hold_for_side_stream(p.grad)                      # read by the gather launch on the side stream
                    p.grad = v                                            # the optimizer reads the (soon reduced) bucket
            st['filled'].record(cur)

    def _launch_gpu(self, bi):
        """All-reduce of a filled bucket, enqueued from the MAIN stream once it has waited for the fill.  Never from the side stream:
        the collective's own stream would then carry a barrier that waits for the low-priority side stream, which lags the main
        stream by milliseconds -- and HIP streams share a handful of hardware queues, so such a barrier can sit in front of the
        main stream's kernels.  Measured on one MI355X with a one-rank RCCL group: 43.8 ms per step instead of 31.4 (44 / 34 / 31 ms
        with GPU_MAX_HW_QUEUES = 4 / 8 / 2: pure queue aliasing).  A barrier that waits for the MAIN stream's recent past is harmless
        wherever it lands."""
        cur = torch.cuda.current_stream(self.buckets[bi][0].device)
        st = self._gpu[bi]
        st['pre'].record(cur)                         # when the main stream reaches the wait
        cur.wait_event(st['filled'])
        self._waits.append((st['pre'], st['filled']))
        self._launch(bi)

    def _on_grad(self, p):
        bi, vi = self.where[p]
        if p.is_cuda:
            self.pending[bi] -= 1
            while self.filled < len(self.buckets) and self.pending[self.filled] <= 0:
                self._fill_gpu(self.filled)
                self.filled += 1
            # a bucket's collective goes out one bucket late: by then the side stream has (almost always) passed its fill, and
            # the main stream's wait for it costs nothing
            while self.next_launch < self.filled - self.late:
                self._launch_gpu(self.next_launch)
                self.next_launch += 1
            return
        view, wview = self.buckets[bi][2][vi], self.buckets[bi][4][vi]
        with self._side(p):
            if p.grad.data_ptr() != view.data_ptr():
                wview.copy_(p.grad)
                if p.is_cuda:
                    from .ops import hold_for_side_stream
                    hold_for_side_stream(p.grad)           # still being written / read on the side stream
                p.grad = view                              # the optimizer reads the (soon reduced) bucket
            elif wview.data_ptr() != view.data_ptr():
                wview.copy_(view)
            self.pending[bi] -= 1
            self._launch_ready()

    def _launch_ready(self):
        while self.next_launch < len(self.buckets) and self.pending[self.next_launch] <= 0:
            self._launch(self.next_launch)
            self.next_launch += 1

    def _launch(self, bi):
        if self.world == 1:
            return
        flat = self.buckets[bi][3]
        backend = dist.get_backend(self.group)
        if self.average and backend == 'nccl':
            self.handles[bi] = (dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), False)
        else:
            self.handles[bi] = (dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), self.average)

    def finish(self):
        """Launch what is still outstanding (parameters that received no gradient contribute zeros), wait for every
        collective and re-arm for the next step."""
        self.last = {'launched_in_backward': self.next_launch, 'collectives': 0, 'waits': []}
        if self.params and self.params[0].is_cuda:
            from .ops import join_side_stream
            join_side_stream(force=True)
        for bi in range(self.next_launch, len(self.buckets)):
            flat, plist, views, wire, wviews = self.buckets[bi]
            if flat.is_cuda:
                if bi >= self.filled:
                    self._fill_gpu(bi)
                self._launch_gpu(bi)
                continue
            for p, v, wv in zip(plist, views, wviews):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    if p.grad is None:
                        wv.zero_()
                    else:
                        wv.copy_(p.grad)
                    p.grad = v
            self._launch(bi)
        self.next_launch = len(self.buckets)
        for bi, h in enumerate(self.handles):
            if h is not None:
                self.last['collectives'] += 1
                h[0].wait()
                flat, wire = self.buckets[bi][0], self.buckets[bi][3]
                if h[1]:
                    wire.div_(self.world)
                if wire.data_ptr() != flat.data_ptr():
                    flat.copy_(wire)
        self.last['waits'], self._waits = self._waits, []
        self.reset()

    def remove(self):
        for h in self.hooks:
            h.remove()
