#!/usr/bin/env python3
"""Benchmark of the hot path: YOLOv3 (Darknet-53 + FPN + 3-scale head) train step, 640x640, 32 images per GPU.

    python bench.py --gpus N --steps K --warmup W

ONE command for N devices, like the reference's nn.DataParallel wrapper (demos/yolov3_u/train.py:85): with N > 1 and no
torchrun environment, this process -- before it makes any GPU call -- starts `python -m torch.distributed.run` with N ranks of
itself (one process per GPU, RCCL), relays rank 0's JSON line and exits with the children's status.  Started by
torch.distributed.run directly (RANK / WORLD_SIZE set) it is simply one of the ranks.

A step = forward + target assignment + yolov3_loss + backward + gradient all-reduce (N > 1) + Adam on one synthetic
batch that is resident in HBM before the timed region.  Rank 0 prints ONE JSON line: BASELINE.json's metric
(images/sec, whole job), plus
  roofline     -- the dominant kernel class (MFMA implicit-GEMM convolution), algorithmic FLOPs per launch divided by
                  its average launch duration, measured live with HIP events on the launch stream in the timed region;
  cpu_baseline -- the CPU oracle (a port of the reference's CPU path: same model / loss / Adam / step contract of
                  utils/fit.py:52-66) timed on this host's cores on a bounded sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The host configuration is whatever the caller's environment says: bench.py sets no runtime variables of its own (round 3 defaulted
# HSA_ENABLE_INTERRUPT=0 here as an unverified mitigation for a rare host stall; a benchmark must not run a host configuration the
# library's users do not get).  The variables that change how the host waits / how streams map to hardware queues are reported in
# config.host_env as found.  The stall itself is made self-diagnosing instead: see `stall_watch` in the timed loop.
import torch  # noqa: E402

METRIC = 'images/sec (640×640, bs=32/GPU) YOLOv3 train step, 1/2/4/8 MI355X'
PEAK_BF16_TFLOPS = 2500.0       # dense MFMA bf16 peak of MI355X (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
TRAIN_GFLOP_PER_IMAGE_640 = 466.97   # fwd + dgrad (not conv0) + wgrad conv FLOPs


def usable_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(size, batch, budget_s=100.0):
    """Time the CPU oracle (port of the reference's CPU path, utils/fit.py:47-71) on this host's cores, BASELINE.md section 4:
    the bench workload itself -- ``batch`` x 3 x size x size, fp32, 1 warm-up + up to 2 timed steps on all cores this process may
    use -- plus a 1-thread figure on one image of the same size.  Bounded: when the warm-up step says a timed step would not fit
    the budget, the warm-up is the sample; when even that is out of reach (a small host), one image per step is timed instead
    and flagged ``port-sample``."""
    from oracle import train as otrain
    from fastvision_amd.synthetic import synthetic_batch
    cores = usable_cores()

    def run(b, threads, max_timed, budget):
        torch.set_num_threads(threads)
        images, tg = synthetic_batch(b, size)
        net, crit = otrain.make_library(20220504)
        opt = otrain.make_adam(net)
        print(f'[bench] cpu_baseline: oracle on {threads} thread(s), {b}x3x{size}x{size} ...', file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        _, tw = otrain.train_steps(net, crit, opt, images, tg, 1)          # warm-up (also bounds the cost of a step)
        print(f'[bench] cpu_baseline: warm-up step {tw[0]:.2f} s', file=sys.stderr, flush=True)
        times = []
        while len(times) < max_timed and time.perf_counter() - t0 + tw[0] < budget:
            _, t = otrain.train_steps(net, crit, opt, images, tg, 1)
            times.append(t[0])
            print(f'[bench] cpu_baseline: step {len(times)} {t[0]:.2f} s', file=sys.stderr, flush=True)
        warm_only = not times
        med = sorted(times or tw)[len(times or tw) // 2]
        return med, ('warm-up step only' if warm_only else f'1 warm-up + {len(times)} timed steps')
    try:
        mem_gb = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES') / 2 ** 30
    except (ValueError, OSError):
        mem_gb = 0.0
    full = mem_gb >= 2.5 * batch * (size / 640.0) ** 2 + 8          # the fp32 autograd state of the oracle is ~2 GB per 640 px image
    b = batch if full else 1
    med, what = run(b, cores, 2, budget_s)
    out = {'value': round(b / med, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port' if full else 'port-sample',
           'sample': f'CPU oracle (port of the reference CPU path: same model, yolov3_loss, Adam, utils/fit.py step), fp32, {what} of '
                     f'{b}x3x{size}x{size}, median {med:.2f} s/step, torch threads = {cores}'}
    med1, what1 = run(1, 1, 1, 25.0)
    out['one_thread'] = {'value': round(1.0 / med1, 4), 'unit': 'images/sec', 'cores': 1,
                         'sample': f'{what1} of 1x3x{size}x{size}, median {med1:.2f} s/step, 1 torch thread'}
    torch.set_num_threads(cores)
    return out


def secondary_fp32(dev, with_oracle, steps=10):
    """BASELINE config 2 -- YOLOv3 8x3x416x416, fp32 end to end, the reference's own precision (no autocast anywhere in the reference:
    classfication/models/darknet53.py:5-17) -- as a block of the default line: step time, the convolution classes against the 157.3
    TFLOP/s f32-MFMA peak, and its own parity line: the first step's loss against the CPU oracle on the same seeded batch and
    initialisation (1e-3 relative is north_star's bar).  About 10 s of GPU time plus one oracle forward pass."""
    import fastvision_amd
    from fastvision_amd import FusedAdam, ops as fva_ops
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.profiler import KernelTimer, count_calls
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    B, S = 8, 416
    images_c, targets_c = synthetic_batch(B, S)
    out = {'config': {'workload': f'YOLOv3 Darknet-53 {B}x3x{S}x{S} fp32 train step (BASELINE config 2), COCO-80, random init', 'surface': 'lib'},
           'dtype': 'f32', 'steps': steps}
    with fastvision_amd.compute_dtype(torch.float32):
        torch.manual_seed(20220504)
        net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                     in_channels=3, num_classes=80, training=True).to(dev).train()
        crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4)
        images, targets = images_c.to(dev), targets_c.to(dev)

        def step():
            pred = net(images)
            opt.zero_grad()
            loss = crit(pred, targets)
            loss.backward()
            opt.step()
            return loss
        with count_calls() as cc:
            first = float(step().detach())                # the loss of the seeded initialisation: the parity line below
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        side_was = fva_ops.set_wgrad_side_stream(False)   # exclusive per-class figures, as in the primary line's probe
        with KernelTimer(pool=cc.calls * 3 + 8) as probe:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        fva_ops.set_wgrad_side_stream(side_was)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    ms = el / steps * 1e3
    summ = probe.summary()
    flop = sum(v['flop_total'] for v in summ.values())
    tms = sum(v['ms_total'] for v in summ.values())
    tf = flop / (tms * 1e-3) / 1e12 if tms > 0 else 0.0
    out.update({'value': round(B * steps / el, 2), 'unit': 'images/sec', 'ms_per_step': round(ms, 3),
                'roofline': {'bound': 'mfma', 'kernel': 'igemm_kernel<float> / wgrad_kernel<float>: v_mfma_f32_32x32x2_f32, all three convolution classes, exclusive',
                             'achieved': round(tf, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_F32_TFLOPS, 4), 'traffic': None},
                'kernels': {k: {'tflops': round(v['tflops'], 2), 'ms_per_step': round(v['ms_total'] / 3, 3)} for k, v in summ.items()},
                'loss_first_step': round(first, 6)})
    if with_oracle:
        from oracle import train as otrain
        ref, ref_crit = otrain.make_library(20220504)
        t0 = time.perf_counter()
        with torch.no_grad():
            want = float(ref_crit(ref(images_c), targets_c))
        out['parity'] = {'oracle_loss_first_step': round(want, 6), 'rel_deviation': float(f'{abs(first - want) / abs(want):.3e}'),
                         'tolerance': 1e-3, 'oracle_forward_s': round(time.perf_counter() - t0, 2),
                         'note': 'CPU oracle (port of the reference CPU path) forward + yolov3_loss on the same seeded batch and initialisation'}
    return out


def spawn_ranks(n):
    """Re-launch this script as ``n`` ranks under torch.distributed.run.  Runs in a parent that has not touched the GPU
    (importing torch does not initialise HIP; nothing here asks for a device).  Returns the launcher's exit status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] starting {n} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode            # the ranks inherit stdout: rank 0 prints the one JSON line


def dry_run(args):
    """No GPU: the launcher, the rendezvous, the bucketed gradient reduction (gloo) and the JSON contract, with the model's
    real parameter set on the CPU and stand-in gradients -- what a container without a device can check of the N > 1 path.
    ``value`` is null: nothing of the hot path is measured."""
    import torch.distributed as dist
    from fastvision_amd import parallel
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.synthetic import coco_anchors_px
    rank, world, _ = parallel.init_from_env('gloo')
    torch.manual_seed(20220504 + rank)
    net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(), num_anchors_per_level=[3, 3, 3],
                 in_channels=3, num_classes=80, training=True)
    parallel.broadcast_parameters(net)
    params = [p for p in net.parameters() if p.requires_grad]
    reducer = parallel.GradientReducer(params, average=False, bucket_dtype=torch.bfloat16 if args.wire == 'bf16' else None) if world > 1 else None

    def step():
        for p in params:
            p.grad = None
        sum((p * float(rank + 1)).sum() for p in params).backward()      # d/dp = rank + 1 everywhere
        if reducer is not None:
            reducer.finish()

    def fence():
        if world > 1:
            dist.barrier()
    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    want = float(world * (world + 1) // 2)                               # SUM over ranks of (rank + 1)
    ok = all(bool((p.grad == want).all()) for p in params[::17])
    if rank == 0:
        print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'dry-run (no GPU, gloo, stand-in gradients)',
                          'dry_run': True, 'reduced_gradients_ok': ok, 'n_params': sum(p.numel() for p in params),
                          'dp': reducer.stats() if reducer is not None else None,
                          'config': {'workload': 'launcher / rendezvous / gradient-bucket all-reduce only', 'parallelism': f'dp{world}'}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true', help='issue every step eagerly from Python instead of replaying a captured HIP graph')
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--surface', default='lib', choices=['lib', 'demo'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shapes', action='store_true', help='also print per-layer-shape conv timings to stderr')
    ap.add_argument('--model', default='yolov3', choices=['yolov3', 'faster_rcnn'],
                    help='faster_rcnn: the train step of the reference\'s Faster R-CNN demo at BASELINE config 5 (4x3x800x1333, one GPU)')
    ap.add_argument('--dry-run', action='store_true', help='no GPU: launcher + rendezvous + gradient reduction over gloo only')
    ap.add_argument('--wire', default='fp32', choices=['fp32', 'bf16'],
                    help='dtype of the gradient buckets on the wire for --gpus N > 1: fp32 is what the reference\'s DataParallel reduces (default); '
                         'bf16 halves the bytes over xGMI (recorded in config.grad_wire_dtype)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the Faster R-CNN (BASELINE config 5) block of the default one-GPU line')
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        sys.exit(spawn_ranks(args.gpus))
    if args.dry_run:
        sys.exit(dry_run(args))
    if args.model == 'faster_rcnn':
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_faster
        bench_faster.main(steps=min(args.steps, 20), warmup=args.warmup, cpu_baseline=not args.no_cpu_baseline)
        return

    import torch.distributed as dist
    import fastvision_amd
    from fastvision_amd import FusedAdam, parallel
    from fastvision_amd.profiler import KernelTimer, PyKernelTimer
    from fastvision_amd.synthetic import coco_anchors_feature, coco_anchors_px, synthetic_batch

    rank, world, local = parallel.init_from_env()
    args.gpus = world
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    fastvision_amd.set_compute_dtype(dtype)

    torch.manual_seed(20220504)
    if args.surface == 'lib':
        from fastvision_amd.classfication.models import darknet53
        from fastvision_amd.detection.head import yolov3head
        from fastvision_amd.detection.models import yolov3
        from fastvision_amd.detection.neck import yolov3neck
        from fastvision_amd.loss import Yolov3Loss
        net = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                     num_anchors_per_level=[3, 3, 3], in_channels=3, num_classes=80, training=True).to(dev).train()
        crit = Yolov3Loss(net, 0.5, 0.05, 1.0, 0.5)
        loss_fn = lambda pred, tg: crit(pred, tg)
    else:
        from fastvision_amd.demos.yolov3_u.models import YoloV3
        from fastvision_amd.demos.yolov3_u.utils import ComputeLoss
        net = YoloV3(anchors=tuple(a.to(dev) for a in coco_anchors_feature())).to(dev).train()
        cl = ComputeLoss()
        loss_fn = lambda pred, tg: cl(pred, tg, net)
    parallel.broadcast_parameters(net)
    # One GPU: the whole step is captured in a HIP graph and replayed (graphs.GraphedTrainStep) -- the host issues one launch per
    # step instead of ~900 C-ABI calls.  N > 1 stays eager: the gradient buckets' collectives are launched from autograd hooks.
    use_graph = world == 1 and not args.no_graph and not args.shapes
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.937, 0.999), weight_decay=5e-4, capturable=use_graph)
    reducer = None
    if world > 1:
        # the library loss is "(means) * batch" (loss/yolov3_loss.py:69-71): under the reference's DataParallel it sees the gathered
        # batch, so the ranks' shares are SUMMED (job-wide match counts in the loss kernel); the demo loss is a plain mean: AVG
        lib_surface = args.surface == 'lib'
        if lib_surface:
            crit.data_parallel()
        reducer = parallel.GradientReducer(net.parameters(), bucket_bytes=16 << 20, average=not lib_surface,
                                           bucket_dtype=torch.bfloat16 if args.wire == 'bf16' else None)

    images, targets = synthetic_batch(args.batch, args.size, rank=rank)     # per-rank shard of the global batch (weak scaling)
    images, targets = images.to(dev), targets.to(dev)

    def eager_step():
        pred = net(images)
        opt.zero_grad()
        loss = loss_fn(pred, targets)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss.detach()       # NOT the attached loss: see below
    step = eager_step
    # (Round 4, found by the stall watchdog: a caller that keeps the attached `loss` of step k alive while it issues step k + 1 -- `loss =
    # step()` does -- used to keep step k's autograd nodes, and with them this package's saved activations, alive: 11 GB more live memory,
    # which torch's caching allocator went to the driver for inside the first timed steps -- 105 hipMalloc calls, one of which now and
    # then took 300 ms: the "second timed step" stall of rounds 2 and 3.  The nodes now release their buffers in backward, as torch's own
    # saved tensors do, and the benchmark keeps a detached loss.)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from fastvision_amd.profiler import count_calls
    calls_per_step = 0
    for i in range(max(args.warmup, 1)):
        if i == 0:
            with count_calls() as cc:
                step()
            calls_per_step = cc.calls
        else:
            step()
    fence()
    # untimed probe for the informative `kernels` table: all three convolution classes bracketed with HIP events inside the
    # library, issued eagerly with the weight gradients on the launch stream like everything else, so that every launch has the
    # GPU to itself (exclusive durations).  In the product configuration the wgrad launches run on the library's low-priority
    # side stream, concurrently with the rest of the backward pass: event-to-event durations of wgrad and dgrad launches then
    # include time spent sharing the GPU.
    from fastvision_amd import ops as fva_ops
    probe_steps = 3
    side_was = fva_ops.set_wgrad_side_stream(False)
    step()
    with KernelTimer(pool=calls_per_step * probe_steps + 8) as probe:
        for _ in range(probe_steps):
            step()
        fence()
    # the shader clock under load: MI355X runs its matrix pipes well below the 2.4 GHz the 2.5 PFLOP/s peak is quoted at.
    # One more untimed step with the 8-phase kernel's diagnostic stamps on: cycle counter / wall clock over each block's
    # k-loop (bf16 only -- the 8-phase kernel is a bf16 kernel).
    clock_mhz = None
    import ctypes as C
    from fastvision_amd import _lib as fva_lib
    if args.dtype == 'bf16':
        nstamp = 8192
        stamps = torch.zeros(8 * nstamp, dtype=torch.int64, device=dev)
        fva_lib.call('fva_conv_debug_stamps', C.c_void_p(stamps.data_ptr()), nstamp)
        step()
        fence()
        fva_lib.call('fva_conv_debug_stamps', C.c_void_p(0), 0)
        st = stamps.view(-1, 8).cpu()
        st = st[st[:, 3] > 0]
        if len(st):
            wall_us = (st[:, 2] - st[:, 1]).double() / 100.0
            clock_mhz = float(((st[:, 6] - st[:, 5]).double() / wall_us).median())
    fva_ops.set_wgrad_side_stream(side_was)
    probe_summ = probe.summary()
    dom = 'conv_fwd'

    # The side stream is the product default, but whether two HIP streams (or two branches of a graph) share the GPU to advantage
    # depends on the box and on the collective backend: time three steps each way and keep the faster setting (collectively).
    graph_info = None
    if use_graph:
        from fastvision_amd.graphs import GraphedTrainStep

        def capture(side_on, spans=False):
            fva_ops.set_wgrad_side_stream(side_on)

            def arm():      # the roofline class bracketed by event-record nodes inside the captured sequence
                fva_lib.call('fva_profile_classes', 1, 1)
                fva_lib.call('fva_profile_start', calls_per_step + 8)
            return GraphedTrainStep(net, loss_fn, opt, images, targets, warmup=1, on_capture=arm if spans else None, side_stream=side_on)

        def window(fn, n=3):
            fence()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            fence()
            return (time.perf_counter() - t) / n * 1e3
        side_check, cand = {}, {}
        t_cap = time.perf_counter()
        refused = None
        for name, on in (('off', False), ('on', True)):
            if on and not side_was:
                continue
            try:
                cand[name] = capture(on)
            except RuntimeError as e:          # e.g. the two-branch form below four hardware queues (graphs.GraphedTrainStep refuses it)
                if not on:
                    raise
                refused = f'refused: {e}'
                continue
            cand[name]()
            side_check[name] = round(window(cand[name]), 3)
        side_use = 'on' in cand and side_check['on'] <= side_check['off']
        gstep = cand['on' if side_use else 'off']
        cand.clear()
        graph_ms = side_check['on' if side_use else 'off']
        graph_info = {'capture_s': round(time.perf_counter() - t_cap, 2), 'replay_ms_per_step': graph_ms, 'wgrad_plan': gstep.wgrad_plan}
        if refused:
            graph_info['graph_on'] = refused
        # A captured step has no host work left, but on ROCm 7.2 a graph with a second branch (the weight gradients on the side
        # stream) replays far slower than the same work issued eagerly on two streams, so its replays are single-stream.  When the
        # host keeps up, the eager two-stream step can therefore still be the faster one: time it too and use the winner.
        eager = fva_ops.autotune_wgrad_side_stream(eager_step, fence, steps=3)
        eager_use = eager.pop('use')
        eager_ms = eager['on' if eager_use else 'off']
        graph_info['eager_ms_per_step'] = eager
        if os.environ.get('FVA_BENCH_MODE', 'auto') == 'graph' or (os.environ.get('FVA_BENCH_MODE', 'auto') == 'auto' and graph_ms <= eager_ms):
            fva_ops.set_wgrad_side_stream(side_use)
            step = gstep
            graph_info['used'] = 'graph'
        else:
            use_graph, side_use, step = False, eager_use, eager_step
            fva_ops.set_wgrad_side_stream(side_use)
            del gstep
            graph_info['used'] = 'eager'
            side_check = dict(eager, graph_off=side_check.get('off'), graph_on=side_check.get('on'))
    else:
        side_check = fva_ops.autotune_wgrad_side_stream(step, fence, steps=3)
        side_use = side_check.pop('use')
        fva_ops.set_wgrad_side_stream(side_use)
    step()
    fence()
    # settle: a freshly started process can run its first steps at a different pace (allocator growth, clocks, a busy
    # host); keep stepping, untimed, until two consecutive 3-step windows agree within 5 % (at most 6 windows)
    prev = None
    settle_log = []
    for _ in range(6):
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        issue = time.perf_counter() - t0
        fence()
        w = time.perf_counter() - t0
        settle_log.append((round(w / 3 * 1e3, 2), round(issue / 3 * 1e3, 2)))
        if world > 1:                              # every rank must take the same decision: the steps contain collectives
            tw = torch.tensor([w], device=dev, dtype=torch.float64)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            w = tw.item()
        if prev is not None and abs(w - prev) <= 0.05 * min(w, prev):
            break
        prev = w
    # The dominant kernel is the implicit-GEMM convolution (igemm8_kernel / igemm_kernel: forward + dgrad instances = 2/3 of the
    # convolution time); its roofline figure is measured live with HIP events on its FORWARD launches -- the instances that run
    # alone on the device (dgrad and wgrad launches share the GPU in the product configuration, so their event-to-event
    # durations say nothing about the kernel).  Eager: every 7th forward launch is bracketed inside the timed region (a span
    # costs two event packets on the stream; 7 is coprime to the 74 launches of a step, so the sample walks through all layers).
    # Graph: the timed region replays the product graph untouched; straight after it a second capture of the same step, with an
    # event-record node before and after every forward launch, is replayed and read out.
    SAMPLE_STRIDE = 7
    if use_graph:
        timer = None
    else:
        timer = PyKernelTimer(pool=calls_per_step * args.steps + 8) if args.shapes else \
            KernelTimer(pool=calls_per_step * args.steps + 8, classes=[dom], stride=SAMPLE_STRIDE)
    fence()
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one event per step: where the time goes
    for e in step_marks:             # torch creates the HIP event at the first record(): do that here, not between timed steps
        e.record()                   # (event creation is a driver call; driver calls have stalled for 0.2-0.3 s right after another
    fence()                          # GPU process on the box exited -- the one timed step of 180-310 ms seen twice in this round's runs)
    # Python's cyclic garbage collector: a generation-2 pass over everything this process has built (modules, the oracle's
    # imports, autograd graphs) stops the launching thread for 0.2-0.3 s -- ten steps' worth of queued GPU work runs dry.
    # Collect now, then freeze the survivors into the permanent generation so that later passes only look at new objects;
    # pauses that still happen inside the timed region are recorded.
    import contextlib
    import gc
    gc_log, gc_t = [], [0.0]

    def gc_watch(phase, info):
        if phase == 'start':
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((info['generation'], round((time.perf_counter() - gc_t[0]) * 1e3, 2)))
    gc.collect()
    gc.freeze()
    gc.callbacks.append(gc_watch)
    with (timer if timer is not None else contextlib.nullcontext()) as kt:
        # pre-roll: three steps issued exactly like the timed ones (armed spans, no fence in between), then the fence the contract
        # asks for.  Three of ~70 eager runs of round 2 had ONE timed step -- always the second -- of 180-770 ms: the host blocked
        # in some call while issuing it (host_issue_ms), no garbage-collector pass involved.  Whatever grows lazily the first time
        # the host runs a step ahead in this configuration now does so here.
        if not args.shapes:
            for _ in range(3):
                step()
            fence()
            if kt is not None:
                fva_lib.call('fva_profile_classes', kt.mask, kt.stride)
                fva_lib.call('fva_profile_start', kt.pool)       # span counters back to zero
        # stall_watch: in 4 of ~120 eager runs of rounds 2 and 3 the HOST stopped issuing inside one timed step for 180-770 ms (cause
        # unknown; `host_issue_ms` shows which step).  A watchdog armed per step writes every thread's Python stack to stderr when a step
        # has not returned after 150 ms (five times its normal duration), so the next natural occurrence names the blocking call.
        import faulthandler
        ms0 = torch.cuda.memory_stats(dev)
        t0 = time.perf_counter()
        step_marks[0].record()
        host_marks = [t0]
        for i in range(args.steps):
            faulthandler.dump_traceback_later(0.15, repeat=False, file=sys.stderr, exit=False)
            loss = step()
            faulthandler.cancel_dump_traceback_later()
            step_marks[i + 1].record()
            host_marks.append(time.perf_counter())
        host_s = time.perf_counter() - t0          # time the host needed to ISSUE the steps (no sync inside a step)
        ms1 = torch.cuda.memory_stats(dev)
        fence()
        elapsed = time.perf_counter() - t0
    gc.callbacks.remove(gc_watch)
    final_loss = float(loss.detach())
    roof_how = 'every launch' if args.shapes else f'every {SAMPLE_STRIDE}th launch of the class over the timed region'
    if use_graph:
        # the same step captured with event-record nodes around every forward convolution launch, replayed right after the
        # timed region; falls back to eagerly issued steps if this runtime does not time events recorded by graph nodes
        summ = None
        try:
            if os.environ.get('FVA_BENCH_GRAPH_SPANS', '1') == '0':
                raise RuntimeError('switched off (FVA_BENCH_GRAPH_SPANS=0)')
            gspan = capture(side_use, spans=True)
            for _ in range(3):
                gspan()
            fence()
            kt = KernelTimer(pool=calls_per_step + 8, classes=[dom], stride=1)
            kt._collect()
            cand_summ = kt.summary()
            d0 = cand_summ.get(dom)
            if d0 and d0['launches'] >= 10 and 0.2 * probe_summ[dom]['ms_total'] / probe_steps < d0['ms_total'] < 5 * probe_summ[dom]['ms_total'] / probe_steps:
                summ = cand_summ
                roof_how = 'event-record nodes around every forward launch of the captured step, replay right after the timed region'
            del gspan
        except RuntimeError as e:
            print(f'[bench] spans inside the graph unavailable: {e}', file=sys.stderr)
            fva_lib.load().fva_profile_stop(None, None, None, 0)
        if summ is None:
            fva_ops.set_wgrad_side_stream(side_use)
            with KernelTimer(pool=calls_per_step * 3 + 8, classes=[dom], stride=1) as kt:
                for _ in range(3):
                    eager_step()
                fence()
            summ = kt.summary()
            roof_how = 'every forward launch of 3 eagerly issued steps right after the timed region (graph replays carry no events)'
    else:
        summ = kt.summary()
    params_identical = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        # every rank must hold the same parameters after the same steps (replicated model, reduced gradients): one checksum per rank
        cs = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        params_identical = bool(lo.item() == hi.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        ips = args.batch * world * args.steps / elapsed
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else PEAK_F32_TFLOPS
        d = summ[dom]
        kernel_name = {'conv_fwd': 'igemm8_kernel / igemm_kernel <EPI_STATS>: implicit-GEMM convolution, forward launches (fva_conv_fwd[_acc]; the 13 1x1 launches that also carry the apply pass of the block before them are a class of their own: kernels.conv_fwd_fused)',
                       'conv_dgrad': 'igemm_kernel<EPI_PLAIN> (fva_conv_dgrad)',
                       'conv_wgrad': 'wgrad_kernel (+reduce) (fva_conv_wgrad)'}[dom]
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            traffic = (json.load(open(tpath)).get(dom) or {}).get('bytes_per_launch')   # measured offline with PMC counters
        # north_star's own yardstick: MFMA utilisation on the 3x3 convolutions (exclusive probe figures, all three classes)
        p3 = probe.summary(ksize=3)
        f3, t3 = sum(v['flop_total'] for v in p3.values()), sum(v['ms_total'] for v in p3.values())
        tf3 = f3 / (t3 * 1e-3) / 1e12 if t3 > 0 else 0.0
        conv3x3 = {'tflops': round(tf3, 2), 'frac_of_peak': round(tf3 / peak, 4), 'ms_per_step': round(t3 / probe_steps, 3),
                   'frac_at_measured_clock': None if clock_mhz is None else round(tf3 / (peak * clock_mhz / 2400.0), 4),
                   'per_class_tflops': {k: round(v['tflops'], 2) for k, v in p3.items()},
                   'note': 'all 3x3 convolution launches of the step (forward, dgrad, wgrad incl. its reduce; thin and stride-2 '
                           'layers included), exclusive timings of the probe pass'}
        out = {
            'metric': METRIC, 'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'YOLOv3 Darknet-53 {args.batch}x3x{args.size}x{args.size}/GPU {args.dtype} train step '
                                   '(fwd + target assignment + yolov3_loss + bwd + grad all-reduce + Adam), COCO-80, random init',
                       'surface': args.surface, 'global_batch': args.batch * world, 'image_size': args.size,
                       'parallelism': f'dp{world}', 'wgrad_plan': fva_ops.get_wgrad_plan(),
                       'bn_statistics': 'accumulators' if fva_ops._BN_ACC[0] else 'tables', 'apply_fused_into_1x1': bool(fva_ops._DEFER['on']),
                       'bn_backward_statistics_in_dgrad': bool(fva_ops._BN_FUSE[0]),
                       'host_env': {k: os.environ.get(k) for k in ('HSA_ENABLE_INTERRUPT', 'GPU_MAX_HW_QUEUES', 'FVA_WGRAD_STREAM', 'FVA_WGRAD_PLAN', 'FVA_BN_ACC', 'FVA_FUSE_APPLY', 'FVA_BN_FUSE')
                                    if os.environ.get(k) is not None}},
            'roofline': {'bound': 'mfma', 'kernel': kernel_name, 'achieved': round(d['tflops'], 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(d['tflops'] / peak, 4), 'traffic': traffic,
                         'launches_per_step': probe_summ[dom]['launches'] // probe_steps, 'launches_timed': d['launches'],
                         'sampling': roof_how,
                         # rounds 1-3 pooled ALL forward launches; the 13 launches that now also carry an apply pass left that class in
                         # round 4 -- the same pooled figure, from the probe pass, for comparison across rounds
                         'pooled_with_fused_launches': (lambda a, b: None if b is None else {
                             'achieved': round((a['flop_total'] + b['flop_total']) / ((a['ms_total'] + b['ms_total']) * 1e-3) / 1e12, 2),
                             'frac': round((a['flop_total'] + b['flop_total']) / ((a['ms_total'] + b['ms_total']) * 1e-3) / 1e12 / peak, 4),
                             'launches_per_step': (a['launches'] + b['launches']) // probe_steps,
                             'note': 'probe pass (exclusive), plain + fused forward launches'})(probe_summ[dom], probe_summ.get('conv_fwd_fused')) if dom == 'conv_fwd' else None,
                         'avg_launch_ms': round(d['ms_avg'], 4), 'gflop_per_launch': round(d['flop_per_launch'] / 1e9, 3),
                         'ms_per_step': round(probe_summ[dom]['ms_total'] / probe_steps, 3),
                         'shader_clock_mhz_under_load': None if clock_mhz is None else round(clock_mhz),
                         'peak_at_that_clock': None if clock_mhz is None else round(peak * clock_mhz / 2400.0, 1),
                         'frac_at_that_clock': None if clock_mhz is None else round(d['tflops'] / (peak * clock_mhz / 2400.0), 4),
                         'clock_note': 'peak is the nominal dense figure at 2400 MHz; the clock is the median, over the blocks of the 8-phase '
                                       'convolution launches of one untimed step, of cycle counter / wall clock across the k-loop'},
            'kernels': {k: {'tflops': round(v['tflops'], 2), 'ms_per_step': round(v['ms_total'] / probe_steps, 3),
                            'launches_per_step': v['launches'] // probe_steps} for k, v in probe_summ.items()},
            'conv3x3': conv3x3,
            'kernels_note': f'exclusive per-class figures: all classes bracketed on {probe_steps} untimed steps just before the '
                            'timed region, with the weight gradients on the launch stream (in the timed region they run on a '
                            'low-priority side stream beside the rest of backward, and only the roofline class is bracketed)',
            'wgrad_side_stream': bool(side_use), 'side_stream_check_ms_per_step': side_check,
            'hip_graph': graph_info if graph_info is not None else False,
            'step_tflops': round(TRAIN_GFLOP_PER_IMAGE_640 * (args.size / 640.0) ** 2 * args.batch / 1e3 / (ms_step * 1e-3), 2),
            'loss': round(final_loss, 5), 'host_ms_per_step': round(host_s / args.steps * 1e3, 3),
            'step_ms': [round(step_marks[i].elapsed_time(step_marks[i + 1]), 2) for i in range(args.steps)],
            'host_issue_ms': [round((host_marks[i + 1] - host_marks[i]) * 1e3, 2) for i in range(args.steps)],
            'settle_windows_ms_per_step': settle_log,
            'gc_passes_in_timed_region': [g for g in gc_log if g[1] >= 1.0] or len(gc_log),
            # the one host stall on record that the watchdog has caught so far (round 4: 330 ms in `dx = torch.empty(...)` inside a
            # backward node) was torch's caching allocator on its slow path: segments it had to get from the driver inside the timed steps
            'allocator': {'device_mallocs_in_timed_region': int(ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)),
                          'device_frees_in_timed_region': int(ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0)),
                          'alloc_retries': int(ms1.get('num_alloc_retries', 0)),
                          'reserved_gb': round(ms1.get('reserved_bytes.all.current', 0) / 2 ** 30, 2),
                          'active_peak_gb': round(ms1.get('active_bytes.all.peak', 0) / 2 ** 30, 2)},
        }
        if args.shapes:
            for k, v in sorted(kt.by_shape().items(), key=lambda kv: -kv[1][1]):
                print(f'{k}: launches {v[0]} ms_total {v[1]:.3f} tflops {v[2]:.1f}', file=sys.stderr)
        if reducer is not None:
            # what the driver needs to check that the collective backend really saw N ranks and that buckets overlapped backward
            out['dp'] = reducer.stats()
            out['dp']['parameters_identical_across_ranks'] = params_identical
            out['config']['grad_wire_dtype'] = out['dp']['wire_dtype']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.size, args.batch)
        default_line = world == 1 and not args.no_secondary and not args.shapes and args.surface == 'lib' and args.dtype == 'bf16' and args.batch == 32 and args.size == 640
        if default_line:
            # BASELINE config 2 at the reference's own precision (fp32 end to end) in the same driver-run line
            try:
                del net, opt, images, targets
                gc.collect()
                torch.cuda.empty_cache()
                out['secondary_fp32'] = secondary_fp32(dev, with_oracle=not args.no_cpu_baseline)
            except Exception as e:                                   # never lose the primary line to a secondary workload
                out['secondary_fp32'] = {'error': f'{type(e).__name__}: {e}'}
        if default_line:
            # BASELINE config 5 (the reference's Faster R-CNN demo, demos/faster_rcnn/cfg/_fit.py:6-53) in the same driver-run line:
            # its own step time, convolution classes and CPU baseline (tools/bench_faster.py = bench.py --model faster_rcnn)
            try:
                gc.collect()
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import bench_faster
                sec = bench_faster.main(steps=10, warmup=3, cpu_baseline=not args.no_cpu_baseline, emit=False)
                out['secondary'] = {k: sec[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'dtype', 'config', 'roofline', 'kernels',
                                                         'loss', 'cpu_baseline') if k in sec}
            except Exception as e:                                   # never lose the primary line to the secondary workload
                out['secondary'] = {'error': f'{type(e).__name__}: {e}'}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
