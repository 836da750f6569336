"""Live per-kernel timing with HIP events recorded on the stream each call is launched on (its `stream` argument).

    with KernelTimer() as kt:
        ... training steps ...
    kt.summary()  ->  {class: {'launches', 'ms_total', 'ms_avg', 'flop_per_launch', 'tflops'}}

Only the MFMA convolution entry points are bracketed (they carry >99 % of the step's FLOPs); their algorithmic
FLOPs come from the call's own descriptor: 2 * (B*OH*OW) * Cout * Cin * k*k for forward, dgrad and wgrad alike.
"""
import torch

from . import _lib

CONV_CALLS = {'fva_conv_fwd': 'conv_fwd', 'fva_conv_dgrad': 'conv_dgrad', 'fva_conv_wgrad': 'conv_wgrad',
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


class KernelTimer:
    def __init__(self):
        self.records = []

    def _trace(self, name, args):
        cls = CONV_CALLS.get(name)
        if cls is None:
            return None
        d = args[0]._obj
        oh, ow = (d.H - 1) // d.stride + 1, (d.W - 1) // d.stride + 1
        flop = 2.0 * d.B * oh * ow * d.Cout * d.Cin * d.ksize * d.ksize
        rec = (cls, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), flop,
               (d.Cin, d.Cout, d.ksize, d.stride, oh))
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
