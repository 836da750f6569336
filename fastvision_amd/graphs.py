"""HIP-graph capture of shape-static, host-bound call sequences (inference).

A YOLOv3 eval forward is ~230 kernel launches for ~6 ms of GPU work at B=32: Python cannot issue them that fast, so the
eager loop is host-bound.  ``GraphedCallable`` runs the callable once under ``torch.cuda.graph`` (hipGraph underneath: the
C ABI only launches kernels on the current stream, which is the capture stream) and afterwards replays the whole sequence
with one launch.  Inputs are copied into the captured buffers; outputs are the captured tensors (consume or clone them
before the next call).  ``GraphedTrainStep`` captures the whole training step: its Adam step count and learning rate live on the
device (FusedAdam(capturable=True)) and its target table has a fixed capacity.
"""
import torch

__all__ = ['GraphedCallable', 'graphed_eval', 'GraphedTrainStep']


class GraphedCallable:
    def __init__(self, fn, *example_inputs, warmup=2):
        self.inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():       # lazy one-time work (weight packing, kernel attributes) happens here
            for _ in range(warmup):
                fn(*self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.outputs = fn(*self.inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs


def graphed_eval(model, example_images):
    """Capture ``model(images)`` of a model in eval mode (library Yolov3: returns (head_out, decoded rows))."""
    if model.training:
        raise RuntimeError('graphed_eval: put the model in eval mode first (model.eval())')
    return GraphedCallable(lambda im: model(im), example_images)


class GraphedTrainStep:
    """The reference's per-batch training step (utils/fit.py:52-66: forward, zero_grad, loss, backward, optimizer step) captured
    ONCE in a HIP graph and replayed with one launch per batch.

    Eager, a YOLOv3 step is ~900 C-ABI calls issued from Python (15-23 ms of host time for ~31 ms of GPU work at B=32, 640 px);
    replayed, the host needs ~0.1 ms and the kernels run back to back.  What used to be host scalars lives on the device:
      * the Adam step count and learning rate (FusedAdam(capturable=True): fva_adam_step_dev; an LR scheduler may keep rewriting
        ``param_groups[i]['lr']`` -- the wrapper refreshes the device scalar before each replay);
      * the target table: a fixed-capacity [max_targets, 6] buffer whose unused rows are zero.  A zero-size box matches no anchor
        (loss/yolov3_loss.py:98-99: max(r, 1/r) = inf), so the library loss sees exactly the rows the reference would; the demo
        loss assigns EVERY row to its best anchor, so it must be captured with its exact target count (max_targets=None).
    The captured sequence includes the weight re-pack, the weight gradients, BatchNorm running statistics and the optimizer.
    Results are bit-identical with the eager step issued under the same weight-gradient plan (same kernels, same order, same
    split-K factors; ``tests/test_gpu_graph.py`` checks that).  The plan (``ops.get_wgrad_plan()``) fixes the fp32 summation order
    of dW and in its default 'auto' mode follows the side-stream switch: a single-stream capture therefore runs the 'alone' plan and
    equals the eager SINGLE-stream step bit for bit, while the eager two-stream step ('beside' plan) differs from it in the last bits
    of dW; ``ops.set_wgrad_plan('alone' | 'beside')`` pins one plan for both.  ``self.wgrad_plan`` records what was captured.
    ``side_stream`` (default False): capture on ONE stream.  A second graph branch for the weight gradients replays at equal
    priority and takes CUs from the critical path -- 43-44 ms instead of 32 per YOLOv3 step (the eager
    step's side stream is low-priority, which graph kernel nodes cannot be on this runtime) -- so the library's side stream is
    switched off around warm-up and capture and put back afterwards; pass True to capture the two-branch form anyway.
    Several graphs may be captured on one optimizer (another batch size, a re-capture): each owns its pointer-table staging.

        step = GraphedTrainStep(model, lambda pred, tg: criterion(pred, tg), optimizer, images, targets, max_targets=1280)
        for images, targets in loader:
            loss = step(images, targets)          # a device tensor, overwritten by the next call
    """

    def __init__(self, model, loss_fn, optimizer, images, targets, max_targets=None, warmup=2, pre_step=None, on_capture=None,
                 side_stream=False):
        if not getattr(optimizer, 'capturable', False):
            raise RuntimeError('GraphedTrainStep needs FusedAdam(..., capturable=True): step count and LR must live on the device')
        if not model.training:
            raise RuntimeError('GraphedTrainStep: put the model in train mode first')
        self.model, self.loss_fn, self.optimizer, self.pre_step = model, loss_fn, optimizer, pre_step
        self.params = [p for g in optimizer.param_groups for p in g['params']]
        T = int(targets.shape[0])
        self.capacity = int(max_targets) if max_targets is not None else T
        if T > self.capacity:
            raise ValueError(f'{T} targets exceed max_targets={self.capacity}')
        self.images = images.detach().clone()
        self.targets = torch.zeros((self.capacity, targets.shape[1]), dtype=torch.float32, device=images.device)
        self.targets[:T].copy_(targets)
        # warm-up steps do the lazy one-time work (kernel attributes, pointer tables, allocator growth, Adam state) -- on a copy of
        # the training state that is put back afterwards, so that capture leaves parameters, statistics and moments untouched
        from . import ops
        if side_stream:
            # Diagnosed in round 3 (tools/hwq_capture_check.py under faulthandler): with GPU_MAX_HW_QUEUES=2 a graph that holds a second
            # branch captures, then segfaults on the HOST inside hipGraphLaunch (torch/cuda/graphs.py replay) -- the runtime maps
            # parallel branches onto hardware queues it was told not to create; 4 queues replay fine, and so does the single-stream
            # capture with 2.  Refuse the combination instead of crashing the process.
            import os
            q = os.environ.get('GPU_MAX_HW_QUEUES')
            if q is not None and q.strip().isdigit() and int(q) < 4:
                raise RuntimeError(f'GraphedTrainStep(side_stream=True): a two-branch HIP graph crashes in hipGraphLaunch with GPU_MAX_HW_QUEUES={q} '
                                   '(< 4) on ROCm 7.2; capture single-stream (the default) or allow at least 4 hardware queues')
        side_was = ops.set_wgrad_side_stream(bool(side_stream) and ops._SIDE['on'])
        self.wgrad_plan = ops.get_wgrad_plan()
        try:
            snap = self._snapshot()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._restore(snap)
            optimizer.zero_grad(set_to_none=True)
            torch.autograd.graph.increment_version(self.params)      # the captured sequence must start with the weight re-pack
            steps_before = [optimizer._step_of(p) if optimizer.state.get(p) else None for p in self.params]
            self._adam_staging = optimizer.begin_capture()           # this graph's own pointer-table words (kept alive with it)
            if on_capture is not None:
                on_capture()                                         # e.g. arm the library's event spans: they become graph nodes
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._step()
        finally:
            ops.set_wgrad_side_stream(side_was)
        for p, st in zip(self.params, steps_before):                 # capture ran the host side of optimizer.step() once
            if st is not None:
                optimizer.state[p]['step'] = st
        torch.autograd.graph.increment_version(self.params)
        self.replays = 0

    def _step(self):
        pred = self.model(self.images)
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(pred, self.targets)
        loss.backward()
        if self.pre_step is not None:
            self.pre_step()
        self.optimizer.step()
        return loss.detach()

    def _snapshot(self):
        opt = self.optimizer
        bufs = [b for b in self.model.buffers()]
        started = [p for p in self.params if opt.state.get(p)]
        return {'params': [p.detach().clone() for p in self.params], 'bufs': [b.detach().clone() for b in bufs],
                'moments': {p: (opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone(), opt._step_of(p)) for p in started},
                'dev': {gi: d['state'].clone() for gi, d in opt._dev.items()}}

    def _restore(self, snap):
        opt = self.optimizer
        with torch.no_grad():
            for p, v in zip(self.params, snap['params']):
                p.copy_(v)
            for b, v in zip(self.model.buffers(), snap['bufs']):
                b.copy_(v)
            for p in self.params:
                st = opt.state.get(p)
                if not st:
                    continue
                if p in snap['moments']:
                    m, v, k = snap['moments'][p]
                    st['exp_avg'].copy_(m)
                    st['exp_avg_sq'].copy_(v)
                    st['step'] = k
                else:                                   # state born in the warm-up: back to a fresh optimizer's zeros
                    st['exp_avg'].zero_()
                    st['exp_avg_sq'].zero_()
                    st['step'] = 0
            for gi, d in opt._dev.items():
                if gi in snap['dev']:
                    d['state'].copy_(snap['dev'][gi])
                else:
                    d['state'].zero_()

    def __call__(self, images=None, targets=None):
        if images is not None:
            self.images.copy_(images, non_blocking=True)
        if targets is not None:
            T = int(targets.shape[0])
            if T > self.capacity:
                raise ValueError(f'{T} targets exceed the captured capacity {self.capacity}')
            if T < self.capacity:
                self.targets[T:].zero_()
            self.targets[:T].copy_(targets, non_blocking=True)
        self.optimizer.sync_lr()
        from . import ops
        ops.settle_accumulators()                # the captured launches assume zero forward accumulators (ops._AccState)
        self.graph.replay()
        ops.accumulators_after_replayed_step()
        self.replays += 1
        self.optimizer.note_replayed_steps(1)
        torch.autograd.graph.increment_version(self.params)          # eager code that follows must see the parameters as changed
        return self.loss
