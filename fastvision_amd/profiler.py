"""Live per-kernel timing with HIP events recorded on the stream each call is launched on (its `stream` argument).

Two implementations with one interface: ``KernelTimer`` arms the spans built into the library (fva_profile_start / _stop:
no per-call host work, which matters when the host has little headroom over the GPU), ``PyKernelTimer`` brackets the calls
from Python with torch events and also keeps each call's layer shape (``by_shape``, for tuning).

    with KernelTimer() as kt:
        ... training steps ...
    kt.summary()  ->  {class: {'launches', 'ms_total', 'ms_avg', 'flop_per_launch', 'tflops'}}

Only the MFMA convolution entry points are bracketed (they carry >99 % of the step's FLOPs); their algorithmic
FLOPs come from the call's own descriptor: 2 * (B*OH*OW) * Cout * Cin * k*k for forward, dgrad and wgrad alike.
"""
import torch

from . import _lib

CONV_CALLS = {'fva_conv_fwd': 'conv_fwd', 'fva_conv_fwd_acc': 'conv_fwd', 'fva_conv1x1_fwd_apply': 'conv_fwd_fused', 'fva_conv1x1_fwd_apply_acc': 'conv_fwd_fused', 'fva_conv_dgrad': 'conv_dgrad', 'fva_conv_dgrad_bnstats': 'conv_dgrad', 'fva_conv_wgrad': 'conv_wgrad',
              'fva_head_fwd': 'conv_fwd'}


_streams = {}


def _stream_of(arg):
    """torch handle for the raw hipStream_t an entry point was given (its last argument)."""
    raw = getattr(arg, 'value', arg) or 0
    st = _streams.get(raw)
    if st is None:
        st = _streams[raw] = torch.cuda.ExternalStream(raw) if raw else torch.cuda.default_stream()
    return st


class _Span:
    def __init__(self, rec, stream):
        self.rec, self.stream = rec, stream

    def __enter__(self):
        self.rec[1].record(self.stream)

    def __exit__(self, *a):
        self.rec[2].record(self.stream)


class _Counter:
    """Tracer that only counts the bracketed calls (used on a warm-up step to size a KernelTimer's event pool)."""

    def __init__(self):
        self.calls = 0

    def _trace(self, name, args):
        if name in CONV_CALLS:
            self.calls += 1
        return None

    def __enter__(self):
        self.prev = _lib.tracer
        _lib.tracer = self._trace
        return self

    def __exit__(self, *a):
        _lib.tracer = self.prev


def count_calls():
    return _Counter()


class PyKernelTimer:
    def __init__(self, pool=0):
        """pool: number of bracketed calls expected.  Their HIP events are created (and recorded once, which is what makes
        the runtime allocate them) HERE, outside the region being timed: creating thousands of timing events costs a kernel-
        driver call each, and those calls can stall for seconds while the driver is still tearing down a previous GPU
        process -- it must not be charged to the step."""
        self.records = []
        self.pool = []
        for _ in range(pool):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            self.pool.append((e0, e1))
        if pool:
            torch.cuda.synchronize()

    def _trace(self, name, args):
        cls = CONV_CALLS.get(name)
        if cls is None:
            return None
        d = args[0]._obj
        oh, ow = (d.H - 1) // d.stride + 1, (d.W - 1) // d.stride + 1
        flop = 2.0 * d.B * oh * ow * d.Cout * d.Cin * d.ksize * d.ksize
        e0, e1 = self.pool.pop() if self.pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        rec = (cls, e0, e1, flop, (d.Cin, d.Cout, d.ksize, d.stride, oh))
        self.records.append(rec)
        return _Span(rec, _stream_of(args[-1]))

    def __enter__(self):
        self.prev = _lib.tracer
        _lib.tracer = self._trace
        return self

    def __exit__(self, *a):
        _lib.tracer = self.prev

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for cls, e0, e1, flop, _ in self.records:
            s = out.setdefault(cls, {'launches': 0, 'ms_total': 0.0, 'flop_total': 0.0})
            s['launches'] += 1
            s['ms_total'] += e0.elapsed_time(e1)
            s['flop_total'] += flop
        for s in out.values():
            s['ms_avg'] = s['ms_total'] / s['launches']
            s['flop_per_launch'] = s['flop_total'] / s['launches']
            s['tflops'] = s['flop_total'] / (s['ms_total'] * 1e-3) / 1e12 if s['ms_total'] > 0 else 0.0
        return out

    def by_shape(self):
        """{(class, Cin, Cout, k, stride, OH): (launches, ms_total, tflops)} for tuning."""
        torch.cuda.synchronize()
        out = {}
        for cls, e0, e1, flop, shape in self.records:
            s = out.setdefault((cls,) + shape, [0, 0.0, 0.0])
            s[0] += 1
            s[1] += e0.elapsed_time(e1)
            s[2] += flop
        return {k: (v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0) for k, v in out.items()}


class KernelTimer:
    """Spans recorded inside the library around fva_conv_fwd / fva_head_fwd / fva_conv_fwd_bnact (class conv_fwd),
    fva_conv_dgrad (conv_dgrad), fva_conv_wgrad incl. its reduce (conv_wgrad) and fva_conv1x1_fwd_apply[_acc] (conv_fwd_fused: a 1x1
    forward launch that also carries the BatchNorm + SiLU apply pass of the block before it; its FLOPs are the convolution's only).  ``pool`` = number of calls expected
    while armed (more are simply not timed); the events are created in __enter__, before anything the caller times."""
    CLASSES = ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'conv_fwd_fused')

    def __init__(self, pool=4096, classes=None, stride=1):
        """classes: bracket only these (default all); stride: of their calls, only every stride-th one.  A span costs
        two event packets on the stream and ~7 us of GPU time, so a throughput run brackets a sample of the class it reports."""
        self.pool = int(pool)
        self.spans = None
        self.mask = sum(1 << self.CLASSES.index(c) for c in (classes or self.CLASSES))
        self.stride = int(stride)

    def __enter__(self):
        _lib.call('fva_profile_classes', self.mask, self.stride)
        _lib.call('fva_profile_start', self.pool)
        return self

    def __exit__(self, *a):
        self._collect()

    def _collect(self):
        if self.spans is None:
            import ctypes as C
            n = self.pool
            cls, flop, ms = (C.c_int32 * n)(), (C.c_double * n)(), (C.c_float * n)()
            got = _lib.load().fva_profile_stop(cls, flop, ms, n)
            _lib.call('fva_profile_classes', 0xffffffff, 1)
            self.spans = [(self.CLASSES[cls[i] & 0xff], flop[i], ms[i]) for i in range(got)]
            self.ksizes = [cls[i] >> 8 for i in range(got)]
        return self.spans

    def summary(self, ksize=None):
        """ksize: only the launches of layers with that kernel size (3 = the 3x3 convolutions)."""
        out = {}
        spans = self._collect()
        if ksize is not None:
            spans = [s for s, k in zip(spans, self.ksizes) if k == ksize]
        for c, flop, ms in spans:
            s = out.setdefault(c, {'launches': 0, 'ms_total': 0.0, 'flop_total': 0.0})
            s['launches'] += 1
            s['ms_total'] += ms
            s['flop_total'] += flop
        for s in out.values():
            s['ms_avg'] = s['ms_total'] / s['launches']
            s['flop_per_launch'] = s['flop_total'] / s['launches']
            s['tflops'] = s['flop_total'] / (s['ms_total'] * 1e-3) / 1e12 if s['ms_total'] > 0 else 0.0
        return out
