"""Host-side glue between torch tensors / autograd and the C ABI (include/fastvision_amd.h).

PyTorch is plumbing here: device memory, streams and the autograd tape.  All arithmetic on the hot path runs
in the HIP kernels of libfastvision_amd.so; there is no PyTorch or CPU fallback (CPU tensors raise).

Activation convention: a feature map is an ordinary ``torch.Tensor`` of logical shape [B,C,H,W] that is a
strided *view* of a halo NHWC buffer [B,H+2,W+2,C] (zero border).  Modules hand these views to each other
zero-copy; foreign tensors (any strides, fp32/bf16) are packed on entry.
"""
import ctypes as C
import os
import weakref
import threading

import torch

from . import _lib
from ._lib import BF16, F32, ConvDesc

_state = threading.local()
_default_dtype = torch.bfloat16


def set_compute_dtype(dtype):
    """torch.bfloat16 (default: bf16 storage, fp32 accumulate) or torch.float32 (exact fp32 MFMA path)."""
    global _default_dtype
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError('compute dtype must be torch.bfloat16 or torch.float32')
    _default_dtype = dtype


def get_compute_dtype():
    return getattr(_state, 'dtype', None) or _default_dtype


class compute_dtype:
    """Context manager: ``with compute_dtype(torch.float32): ...``"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = getattr(_state, 'dtype', None)
        _state.dtype = self.dtype

    def __exit__(self, *a):
        _state.dtype = self.prev


def _code(dtype):
    return BF16 if dtype == torch.bfloat16 else F32


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~7 us in Python; the raw query does not."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError(f'fastvision_amd.{who}: tensors must live on the GPU -- this package has no CPU path '
                           '(the CPU oracle lives under oracle/ and is test infrastructure only)')


# ------------------------------------------------------------------------------------------------ layouts
def halo_alloc(B, Cc, H, W, dtype, device, pad=1):
    """Uninitialised halo buffer [B,H+2p,W+2p,C] and its logical [B,C,H,W] view (kernels write the border)."""
    buf = torch.empty((B, H + 2 * pad, W + 2 * pad, Cc), dtype=dtype, device=device)
    return buf, buf[:, pad:pad + H, pad:pad + W, :].permute(0, 3, 1, 2)


def halo_info(x, dtype):
    """If x [B,C,H,W] is a view of a halo buffer of ``dtype`` return (base_ptr, pad); else None."""
    if x.dim() != 4 or x.dtype != dtype:
        return None
    B, Cc, H, W = x.shape
    item = x.element_size()
    for pad in (1, 0):
        Hp, Wp = H + 2 * pad, W + 2 * pad
        if tuple(x.stride()) == (Hp * Wp * Cc, 1, Wp * Cc, Cc):
            lead = pad * (Wp + 1) * Cc
            tail = (B * Hp * Wp * Cc) - lead
            st = x.untyped_storage()
            off = x.storage_offset()
            if off >= lead and (off + tail) * item <= st.nbytes():
                return x.data_ptr() - lead * item, pad
    return None


def to_halo(x, dtype, min_pad):
    """Return (keepalive, base_ptr, pad) for x in halo layout of ``dtype`` with pad >= min_pad (packing if needed)."""
    info = halo_info(x, dtype)
    if info is not None and info[1] >= min_pad:
        return x, info[0], info[1]
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    B, Cc, H, W = x.shape
    buf, _ = halo_alloc(B, Cc, H, W, dtype, x.device, 1)
    sb, sc, sh, sw = x.stride()
    _lib.call('fva_pack_nchw', _code(dtype), _p(x), 1 if x.dtype == torch.bfloat16 else 0, sb, sc, sh, sw, _p(buf), 1,
              B, Cc, H, W, _stream())
    return buf, buf.data_ptr(), 1


def to_dense(g, dtype):
    """Gradient [B,C,H,W] -> (keepalive, ptr) of a dense NHWC buffer of ``dtype`` (zero-copy when it already is one)."""
    info = halo_info(g, dtype)
    if info is not None and info[1] == 0:
        return g, info[0]
    B, Cc, H, W = g.shape
    if g.dtype not in (torch.float32, torch.bfloat16):
        g = g.float()
    buf = torch.empty((B, H, W, Cc), dtype=dtype, device=g.device)
    sb, sc, sh, sw = g.stride()
    _lib.call('fva_pack_nchw', _code(dtype), _p(g), 1 if g.dtype == torch.bfloat16 else 0, sb, sc, sh, sw, _p(buf), 0,
              B, Cc, H, W, _stream())
    return buf, buf.data_ptr()


def dense_view(buf):
    """Dense NHWC buffer [B,H,W,C] -> logical [B,C,H,W] view."""
    return buf.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ conv + BN + SiLU
class _BNState:
    """Tensors of one nn.BatchNorm2d that the kernels update in place."""

    def __init__(self, bn):
        self.rm, self.rv, self.nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
        self.momentum = 0.1 if bn.momentum is None else bn.momentum
        self.eps = bn.eps


class _PackRegistry:
    """All conv weights seen so far, so that the per-step re-pack (fp32 OIHW -> MFMA operand layouts) is ONE launch.

    The first stale weight met in a step triggers ``fva_conv_pack_weights_multi`` over every registered layer (the
    optimizer updates them all at once); the other layers then find their cache fresh."""

    def __init__(self):
        self.entries = {}          # id(weight) -> (weakref(weight), desc fields, wf, wd); one entry per parameter
        self.table = None

    def add(self, weight, desc, wf, wd):
        import weakref
        self.entries[id(weight)] = (weakref.ref(weight), (desc.Cout, desc.Cin, desc.ksize, desc.dtype, desc.stride), wf, wd)
        self.table = None

    def repack_all(self, device):
        dead = [k for k, (r, *_) in self.entries.items() if r() is None]
        for k in dead:
            del self.entries[k]
        if dead:
            self.table = None
        live = [(r(), m, wf, wd) for r, m, wf, wd in self.entries.values() if r().device == device]
        if not live:
            return
        if self.table is None or self.table[0] != len(live) or self.table[3] != device:
            arr = (_lib.PackEntry * len(live))()
            mx = tiles = 0
            for i, (w, (co, ci, k, dt, st), wf, wd) in enumerate(live):
                e = arr[i]
                e.w, e.w_fwd, e.w_dgrad = w.data_ptr(), wf.data_ptr(), wd.data_ptr()
                e.Cout, e.Cin, e.ksize, e.dtype = co, ci, k, dt
                e.taps_fwd, e.taps_dgrad = wf.numel() // (co * ci), wd.numel() // (co * ci)
                e.dgrad_paired = 1 if (st == 2 and e.taps_dgrad == 12) else 0
                e.tile_start = tiles                         # one block per 32 x 32 weight tile (fva_conv_pack_weights_tiled)
                tiles += ((co + 31) // 32) * ((ci + 31) // 32)
                mx = max(mx, wf.numel(), wd.numel())
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            self.table = (len(live), raw, mx, device, [w.data_ptr() for w, *_ in live], tiles)
        if self.table[4] != [w.data_ptr() for w, *_ in live]:      # a parameter moved: rebuild
            self.table = None
            return self.repack_all(device)
        if os.environ.get('FVA_PACK_TILED', '1') != '0':
            _lib.call('fva_conv_pack_weights_tiled', _p(self.table[1]), self.table[0], self.table[5], _stream())
        else:
            _lib.call('fva_conv_pack_weights_multi', _p(self.table[1]), self.table[0], self.table[2], _stream())
        for w, (co, ci, k, dt, st), wf, wd in live:
            w._fva_packed = ((w._version, w.data_ptr(), dt), wf, wd)


_registry = _PackRegistry()


def packed_weights(weight, desc, dtype, cache=True, owner=None):
    """(w_fwd, w_dgrad) in the MFMA operand layouts.  The pair is cached ON the parameter object and re-packed
    only when its autograd version counter (bumped by every in-place update, incl. FusedAdam) or data moved.
    ``owner``: the parameter that ``weight`` is a same-memory view of (an nn.Linear weight seen as a 1x1 filter): the cache and
    the one-launch re-pack registry then hang on the parameter."""
    holder = weight if owner is None else owner
    key = (holder._version, weight.data_ptr(), desc.dtype)
    hit = getattr(holder, '_fva_packed', None) if cache else None
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    if hit is not None and hit[0][1:] == key[1:] and isinstance(holder, torch.nn.Parameter):
        _registry.repack_all(holder.device)            # stale registered parameter: refresh every layer in one launch
        hit = holder._fva_packed
        if hit[0] == key:
            return hit[1], hit[2]
    lib = _lib.load()
    wf = torch.empty(lib.fva_conv_packed_elems(C.byref(desc), 0), dtype=dtype, device=weight.device)
    wd = torch.empty(lib.fva_conv_packed_elems(C.byref(desc), 1), dtype=dtype, device=weight.device)
    _lib.call('fva_conv_pack_weights', C.byref(desc), _p(weight), _p(wf), _p(wd), _stream())
    if cache:
        holder._fva_packed = (key, wf, wd)
        if isinstance(holder, torch.nn.Parameter):
            _registry.add(holder, desc, wf, wd)
    return wf, wd


class _Saved:
    pass


def _released(what):
    return RuntimeError(f'fastvision_amd: {what}: the buffers saved for backward have already been released (a second backward pass through the same '
                        'graph; as with torch\'s own saved tensors, this package keeps them for ONE backward pass)')


# ---- BatchNorm-backward statistics in the epilogue of the dgrad launch that produces dz --------------------------------------
# The first pass of a block's BatchNorm backward (sum dU, sum dU * xhat over the batch, dU = dz * SiLU') needs dz complete.  When
# the block's output z has ONE consumer and that consumer is one of our convolutions, the dgrad launch of the consumer writes
# exactly dz (incl. the residual addend): it then also reads the block's y and leaves the partial sums (fva_conv_dgrad_bnstats),
# and the block's own backward skips fva_bn_silu_bwd_reduce -- one read of dz and one launch less per layer (65 of the 72
# BatchNorm layers of YOLOv3).  Forward tags every block output with the state of the block that produced it (z._fva_prod) and
# counts the consumers it sees; backward trusts the sums only if the gradient it receives IS the consumer's buffer (pointer
# identity; the consumer keeps a second reference to that buffer so that autograd cannot accumulate another gradient into it in
# place): with a consumer this module does not know, or several, the block falls back to its own reduce pass.
_BN_FUSE = [os.environ.get('FVA_BN_FUSE', '1') != '0']


def set_bn_backward_fusion(on):
    """Switch the fused BatchNorm-backward statistics on or off (default on; env FVA_BN_FUSE=0).  Returns the previous setting."""
    prev = _BN_FUSE[0]
    _BN_FUSE[0] = bool(on)
    return prev


def _note_consumer(x):
    """A consumer of x that will hand back a gradient: called by every autograd node of this module in forward."""
    src = getattr(x, '_fva_prod', None)
    if src is not None:
        src.consumers += 1
    return src


def _fuse_struct(src, part, acc=None, replicas=1):
    return _lib.BnBwdFuse(src.y.data_ptr(), src.scale.data_ptr(), src.shift.data_ptr(), src.mean.data_ptr(), src.rstd.data_ptr(),
                          None if part is None else part.data_ptr(), None if acc is None else acc.data_ptr(), replicas)


def _dgrad(d, dy, wd, dx, addend_ptr, src, dtype):
    """fva_conv_dgrad, with the producer's BatchNorm-backward statistics in its epilogue when dx is that producer's whole dz."""
    if (_BN_FUSE[0] and src is not None and src.consumers == 1 and src.training and src.dtype == dtype
            and src.M == d.B * d.H * d.W and src.d.Cout == d.Cin and not torch.is_grad_enabled()):
        rows = _lib.load().fva_conv_dgrad_stat_rows(C.byref(d))
        if rows > 0 and getattr(src, 'acc', None) is not None:
            # the producer layer keeps its statistics in accumulators: the epilogue ADDS the sums to its backward one
            st = src.acc
            st.produce(1)
            fs = _fuse_struct(src, None, st.buf[1], st.replicas)
            _lib.call('fva_conv_dgrad_bnstats', C.byref(d), _p(dy), _p(wd), _p(dx), C.c_void_p(addend_ptr or 0), C.byref(fs), _stream())
            src.fused = (None, 0, dx.data_ptr(), dx)
            return
        if rows > 0:
            part = torch.empty((_lib.load().fva_bn_partial_rows(rows), 2, d.Cin), dtype=torch.float32, device=dx.device)
            fs = _fuse_struct(src, part)
            _lib.call('fva_conv_dgrad_bnstats', C.byref(d), _p(dy), _p(wd), _p(dx), C.c_void_p(addend_ptr or 0), C.byref(fs), _stream())
            src.fused = (part, rows, dx.data_ptr(), dx)
            return
    _lib.call('fva_conv_dgrad', C.byref(d), _p(dy), _p(wd), _p(dx), C.c_void_p(addend_ptr or 0), _stream())


# ---- the forward apply pass of a block, deferred into the 1x1 convolution that consumes it -------------------------------------------
# z = SiLU(BN(y)) (+ identity) of a block is an HBM round trip of its own (read y [+ identity], write z), and the residual blocks' conv1
# (1x1) then reads z straight back.  Inside ``defer_apply_scope()`` -- Darknet.forward opens one per stage, around its own modules only --
# a block may leave its apply pass PENDING (its z buffer allocated, not yet written); when the very next launch is a 1x1 layer that
# the fused kernel serves (fva_conv1x1_fwd_apply: bf16 training, Cin % 64 == 0, Cout <= 128), that launch produces z on the way
# to its own MFMAs -- same arithmetic, same bits, z written once, never read back by conv1.  Anything else that comes next, and the end
# of the scope, run the ordinary apply launch first: a pending buffer never leaves the scope.
_DEFER = {'depth': 0, 'pending': None, 'on': os.environ.get('FVA_FUSE_APPLY', '1') != '0', 'fused': 0}


class defer_apply_scope:
    def __enter__(self):
        if _DEFER['depth'] == 0:
            _DEFER['pending'] = None          # a scope that died with an exception leaves nothing behind
        _DEFER['depth'] += 1
        return self

    def __exit__(self, et, ev, tb):
        _DEFER['depth'] -= 1
        if _DEFER['depth'] == 0:
            if et is None:
                flush_pending_apply()
            else:
                _DEFER['pending'] = None
        return False


def set_apply_fusion(on):
    """Switch the deferred / fused forward apply on or off (default on; env FVA_FUSE_APPLY=0).  Returns the previous setting."""
    prev, _DEFER['on'] = _DEFER['on'], bool(on)
    return prev


def flush_pending_apply():
    """Run the apply pass that a block left pending, as its own launch."""
    pd, _DEFER['pending'] = _DEFER['pending'], None
    if pd is not None:
        if pd['fin'] is not None:
            _lib.call('fva_bn_silu_apply_acc', _code(pd['dtype']), _p(pd['y']), C.byref(pd['fin'].desc()), C.c_void_p(pd['res_ptr'] or 0), pd['res_pad'],
                      C.c_void_p(pd['z_ptr']), 1, pd['B'], pd['H'], pd['W'], pd['C'], _stream())
            pd['fin'].state.consumed(0)
            return
        _lib.call('fva_bn_silu_apply', _code(pd['dtype']), _p(pd['y']), _p(pd['scale']), _p(pd['shift']), C.c_void_p(pd['res_ptr'] or 0), pd['res_pad'],
                  C.c_void_p(pd['z_ptr']), 1, pd['B'], pd['H'], pd['W'], pd['C'], _stream())


# ---- BatchNorm statistics without a launch of their own ------------------------------------------------------------------------------
# The table form (every tile of the convolution stores its partial sums, fva_bn_finalize folds them) costs one or two small launches
# per layer and direction, each a dependent step of the chain: 197 launches = 2 ms of a 28 ms step (measured by leaving them out,
# profiles/r04_experiments.md).  In the accumulator form every tile ADDS its partial sums to a per-layer fixed-point accumulator (two
# int64 words per sum: integer addition is associative, so the result does not depend on the order of the atomics and stays
# run-to-run bit-identical), and the launch that consumes the statistics -- the apply pass, the fused 1x1 convolution that carries it,
# the second backward pass -- turns them into its coefficients in its prologue; block 0 also writes mean / rstd / scale / shift (or
# dgamma / dbeta) and updates the running statistics.  The accumulators hang on the BatchNorm weight (one per layer and direction,
# allocated once); who returns which to zero, and what the host tracks about it, is _AccState's docstring.  The stem (its own kernels)
# and eval mode keep the table form / the running statistics; FVA_BN_ACC=0 selects the table form everywhere (what the accumulator form
# is tested against, tests/test_gpu_bn_acc.py).
_BN_ACC = [os.environ.get('FVA_BN_ACC', '1') != '0']
_ACC_STATES = weakref.WeakSet()


def set_bn_accumulators(on):
    """Switch the accumulator form of the BatchNorm statistics on or off (default on; env FVA_BN_ACC=0).  Returns the previous setting."""
    prev, _BN_ACC[0] = _BN_ACC[0], bool(on)
    return prev


class _AccState:
    """The two accumulators of one BatchNorm layer (buf[0]: forward sums, buf[1]: backward sums; int64 [replicas][5][C] each) and what
    the host knows of them: 0 = zero, 1 = a producer has added to it, 2 = consumed (its sums are still there).  The normal sequence needs
    no clearing launch: each direction's consumer zeroes the other direction's accumulator.  ``replicas``: one copy per 65536 output
    pixels (a power of two, at most 32) -- the atomics on one address are served one after the other, ~10 ns each, and a layer at 320 x 320
    runs 12800 tiles."""
    __slots__ = ('buf', 'state', 'replicas', '__weakref__')

    def __init__(self, Cc, M, device):
        self.replicas = _replicas(M)
        self.buf = torch.zeros((2, self.replicas * 5 * Cc), dtype=torch.int64, device=device)
        self.state = [0, 0]
        _ACC_STATES.add(self)

    def produce(self, which):
        if self.state[which] != 0:          # left over from a pass whose other half never ran
            self.buf[which].zero_()
        self.state[which] = 1

    def consumed(self, which):
        self.state[which] = 2
        self.state[1 - which] = 0           # the consumer returned the other direction's accumulator to zero


def settle_accumulators():
    """Bring every forward accumulator to zero whose sums nobody cleared (a training-mode forward pass whose backward never ran, an
    aborted step).  Eager producers do this for themselves; a captured graph replays the launches of a NORMAL step and cannot look, so
    graphs.GraphedTrainStep calls this before each replay (a host loop over the layers; a launch only for a dirty accumulator)."""
    for st in list(_ACC_STATES):
        if st.state[0] != 0:
            st.buf[0].zero_()
            st.state[0] = 0


def accumulators_after_replayed_step():
    """What a replayed step leaves behind on the device, noted on the host: forward accumulators zero (the backward consumers cleared
    them), backward accumulators holding the step's sums (the next forward consumers clear them)."""
    for st in list(_ACC_STATES):
        st.state[0], st.state[1] = 0, 2


def _replicas(M):
    r = 1
    while r < 32 and r * 65536 < M:
        r *= 2
    return r


def _acc_state(gamma, M):
    """The layer's accumulators for a launch over M output pixels.  One _AccState per (device, replica count) is kept for as long as the
    parameter lives -- never replaced: a captured graph holds the addresses of the one it was recorded with, and an eager step of another
    batch size (another replica count) in between must not free it."""
    Cc, R = gamma.numel(), _replicas(M)
    cur = getattr(gamma, '_fva_acc', None)
    if cur is not None and cur.replicas == R and cur.buf.device == gamma.device and cur.buf.shape[1] == R * 5 * Cc:
        return cur
    table = gamma.__dict__.setdefault('_fva_accs', {})
    key = (gamma.device, R, Cc)
    st = table.get(key)
    if st is None:
        st = table[key] = _AccState(Cc, M, gamma.device)
    gamma._fva_acc = st
    return st


class _Fin:
    """What the consumer of a layer's accumulator needs to finalise it (struct fva_bn_fwd_acc)."""

    def __init__(self, state, gamma, beta, bn, mean, rstd, scale, shift):
        self.state, self.keep = state, (gamma, beta, bn, mean, rstd, scale, shift)

    def desc(self):
        gamma, beta, bn, mean, rstd, scale, shift = self.keep
        opt = lambda t: None if t is None else t.data_ptr()
        return _lib.BnFwdAcc(self.state.buf[0].data_ptr(), self.state.buf[1].data_ptr(), self.state.replicas, gamma.data_ptr(), beta.data_ptr(),
                             opt(bn.rm), opt(bn.rv), opt(bn.nbt), bn.momentum, bn.eps,
                             mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())


def conv_block_fwd(x, x_ptr, x_pad, weight, gamma, beta, bn, training, stride, dtype, residual=None, need_ctx=True, may_defer=False):
    """SiLU(BN(conv(x))) [+ residual].  x: logical [B,Cin,H,W]; x_ptr/x_pad describe its halo buffer.
    Returns (z_view, saved).  ``residual`` = (ptr, pad) of a halo buffer with the output's shape.
    ``may_defer``: the caller allows this block's apply pass to stay pending for the next launch (see defer_apply_scope)."""
    B, Cin, H, W = x.shape
    Cout, _, k, _ = weight.shape
    lib = _lib.load()
    # an apply pass left pending by the block before: taken over by this launch when this is the 1x1 layer that reads exactly that
    # buffer and the fused kernel serves it; run on its own otherwise
    pend = _DEFER['pending']
    if pend is not None:
        if (pend['z_ptr'] == x_ptr and k == 1 and stride == 1 and training and dtype == torch.bfloat16 and pend['dtype'] == dtype
                and Cin % 64 == 0 and Cin <= 512 and Cout <= 128 and Cout % 8 == 0 and x_pad == 1
                and (pend['B'], pend['C'], pend['H'], pend['W']) == (B, Cin, H, W)
                and not (pend['fin'] is not None and pend['fin'].keep[0] is gamma)):      # one layer applied twice in a row: its accumulator would be read and written by one launch
            _DEFER['pending'] = None
        else:
            flush_pending_apply()
            pend = None
    x_src = _note_consumer(x) if need_ctx else None
    d = ConvDesc(_code(dtype), B, H, W, Cin, Cout, k, stride, x_pad, 1)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    M = B * OH * OW
    dev = x.device
    wf, wd = packed_weights(weight, d, dtype)
    scale = torch.empty(Cout, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    mean = rstd = None
    if not training and not need_ctx and (residual is None or residual[1] == 1):
        # inference: running statistics fold into a per-channel affine, and affine + SiLU (+ residual) run in the conv
        # epilogue -- one pass over the output instead of conv-out, apply-in, apply-out.  The affine is cached on the
        # running-mean buffer until any of the four BatchNorm tensors changes (version counters / storage).
        key = (gamma._version, beta._version, bn.rm._version, bn.rv._version, gamma.data_ptr(), beta.data_ptr(), bn.rm.data_ptr(),
               bn.rv.data_ptr(), bn.eps)
        hit = getattr(bn.rm, '_fva_eval_affine', None)
        if hit is not None and hit[0] == key:
            scale, shift = hit[1], hit[2]
        else:
            _lib.call('fva_bn_eval_coeffs', Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), bn.eps, _p(scale), _p(shift), _stream())
            bn.rm._fva_eval_affine = (key, scale, shift)
        zbuf, z = halo_alloc(B, Cout, OH, OW, dtype, dev, 1)
        _lib.call('fva_conv_fwd_bnact', C.byref(d), C.c_void_p(x_ptr), _p(wf), _p(scale), _p(shift),
                  C.c_void_p(residual[0] if residual is not None else 0), _p(zbuf), 1, _stream())
        return z, None
    y = torch.empty((M, Cout), dtype=dtype, device=dev)
    fin = None
    if training:
        nblk = lib.fva_conv_stat_blocks(C.byref(d))
        mean = torch.empty_like(scale)
        rstd = torch.empty_like(scale)
        if _BN_ACC[0] and gamma.is_cuda and need_ctx:      # (a forward pass that no backward will follow keeps the table form: nobody would clear its sums)
            st = _acc_state(gamma, M)
            st.produce(0)
            fin = _Fin(st, gamma, beta, bn, mean, rstd, scale, shift)
            stats = None
        else:
            stats = torch.empty((lib.fva_bn_partial_rows(nblk), 2, Cout), dtype=torch.float32, device=dev)
        if pend is not None:
            pf = pend['fin']
            if fin is not None:
                # the block before: its accumulator is finalised by this launch (pf), or its coefficients are final already
                prev = pf.desc() if pf is not None else _lib.BnFwdAcc(None, None, 1, None, None, None, None, None, 0.0, 0.0, None, None,
                                                                      pend['scale'].data_ptr(), pend['shift'].data_ptr())
                _lib.call('fva_conv1x1_fwd_apply_acc', C.byref(d), _p(pend['y']), C.byref(prev), C.c_void_p(pend['res_ptr'] or 0),
                          pend['res_pad'], C.c_void_p(x_ptr), _p(wf), _p(y), _p(st.buf[0]), st.replicas, _stream())
            else:
                if pf is not None:
                    _lib.call('fva_bn_acc_finalize', C.byref(pf.desc()), M, Cin, _stream())
                _lib.call('fva_conv1x1_fwd_apply', C.byref(d), _p(pend['y']), _p(pend['scale']), _p(pend['shift']), C.c_void_p(pend['res_ptr'] or 0),
                          pend['res_pad'], C.c_void_p(x_ptr), _p(wf), _p(y), _p(stats), _stream())
            if pf is not None:
                pf.state.consumed(0)
            _DEFER['fused'] += 1
            pend = None
        elif fin is not None:
            _lib.call('fva_conv_fwd_acc', C.byref(d), C.c_void_p(x_ptr), _p(wf), _p(y), _p(st.buf[0]), st.replicas, _stream())
        else:
            _lib.call('fva_conv_fwd', C.byref(d), C.c_void_p(x_ptr), _p(wf), _p(y), _p(stats), _stream())
        if fin is None:
            _lib.call('fva_bn_finalize', _p(stats), nblk, stats.shape[0], M, Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), _p(bn.nbt),
                      bn.momentum, bn.eps, _p(mean), _p(rstd), _p(scale), _p(shift), _stream())
    else:
        _lib.call('fva_conv_fwd', C.byref(d), C.c_void_p(x_ptr), _p(wf), _p(y), C.c_void_p(0), _stream())
        _lib.call('fva_bn_eval_coeffs', Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), bn.eps, _p(scale), _p(shift), _stream())
    zbuf, z = halo_alloc(B, Cout, OH, OW, dtype, dev, 1)
    rp, rpad = (C.c_void_p(residual[0]), residual[1]) if residual is not None else (C.c_void_p(0), 0)
    if may_defer and training and _DEFER['on'] and _DEFER['depth'] > 0 and dtype == torch.bfloat16:
        # the references keep y / scale / shift / the identity alive until the pass has run (they are saved for backward anyway)
        _DEFER['pending'] = {'z_ptr': zbuf.data_ptr(), 'zbuf': zbuf, 'y': y, 'scale': scale, 'shift': shift, 'dtype': dtype,
                             'res_ptr': residual[0] if residual is not None else None, 'res_pad': rpad, 'keep': x,
                             'B': B, 'H': OH, 'W': OW, 'C': Cout, 'fin': fin}
    elif fin is not None:
        _lib.call('fva_bn_silu_apply_acc', _code(dtype), _p(y), C.byref(fin.desc()), rp, rpad, _p(zbuf), 1, B, OH, OW, Cout, _stream())
        fin.state.consumed(0)
    else:
        _lib.call('fva_bn_silu_apply', _code(dtype), _p(y), _p(scale), _p(shift), rp, rpad, _p(zbuf), 1, B, OH, OW, Cout, _stream())
    s = None
    if need_ctx:
        s = _Saved()
        s.d, s.x, s.x_ptr, s.y, s.scale, s.shift, s.mean, s.rstd, s.wd = d, x, x_ptr, y, scale, shift, mean, rstd, wd
        s.gamma, s.dtype, s.OH, s.OW, s.M, s.wshape = gamma, dtype, OH, OW, M, tuple(weight.shape)
        s.weight = weight
        s.training = training
        s.x_src, s.consumers, s.fused = x_src, 0, None
        s.acc = fin.state if fin is not None else None
        if training:
            z._fva_prod = s
    return z, s


def scale_loss_grads(ctx, gout):
    """Chain rule of the fused losses (loss.Yolov3Loss, the demo's ComputeLoss): their forward pass already filled ctx.grads with
    d loss / d head; backward multiplies by the upstream scalar.  For fp32 buffers that this package allocated the multiply runs IN
    PLACE and is an empty launch when the scalar is exactly 1 (what ``loss.backward()`` passes): `g * gout` was a 274 MB
    read-modify-write at the very start of the backward pass.  The buffers are spent afterwards: a second backward pass raises."""
    grads, ctx.grads = ctx.grads, None
    if grads is None:
        raise _released('the loss')
    res = []
    for g, dt in zip(grads, ctx.dtypes):
        if g is None:
            res.append(None)
            continue
        if (gout.numel() == 1 and gout.dtype == torch.float32 and gout.is_cuda and g.is_cuda and g.dtype == torch.float32 and not gout.requires_grad
                and g.storage_offset() == 0 and g.untyped_storage().nbytes() == g.numel() * 4 and g.data_ptr() % 16 == 0):
            _lib.call('fva_scale_by_device_scalar', _p(g), g.numel(), _p(gout), _stream())
        else:
            g = g * gout
        res.append(g if g.dtype == dt else g.to(dt))
    return res


# ---- weight gradients beside the rest of the backward pass ----------------------------------------------------------------
# Nothing reads a layer's dW before the optimizer (or the gradient all-reduce), so the wgrad launches go to the library's
# low-priority side stream: their blocks fill the CUs that the partly empty last round of the dgrad launches (800 / 400 / 200
# tiles of 256x256 on 256 CUs) and the HBM-bound BatchNorm passes leave idle.  The fork/join events are recorded inside the
# library (two HIP calls per layer); the buffers the side stream touches are simply kept referenced until the join, which a
# callback queued on the autograd engine performs at the end of the backward pass (anything that reads .grad earlier -- the
# gradient-bucket hooks of parallel.GradientReducer -- joins first through join_side_stream()).
_SIDE = {'on': os.environ.get('FVA_WGRAD_STREAM', '1') != '0', 'keep': [], 'queued': False, 'torch': None}

# The split-K plan of the weight gradients (fva_conv_wgrad_plan) fixes the fp32 summation order of dW.  It is a process-wide, sticky
# setting: 'auto' (default) follows the side-stream switch -- 'beside' while the weight gradients run on the side stream, 'alone'
# otherwise --, 'alone' / 'beside' pin it.  It never depends on which stream an individual launch is given, so a layer that falls back
# to the launch stream (gradient accumulation) sums in the same order as its neighbours, and two runs under the same plan are
# bit-identical whether they are issued eagerly on two streams or replayed from a single-stream graph.
_PLAN = {'mode': os.environ.get('FVA_WGRAD_PLAN', 'auto'), 'applied': None}


def _apply_wgrad_plan():
    want = 1 if (_PLAN['mode'] == 'beside' or (_PLAN['mode'] == 'auto' and _SIDE['on'])) else 0
    if _PLAN['applied'] != want:
        _lib.call('fva_conv_wgrad_plan', want)
        _PLAN['applied'] = want


def set_wgrad_plan(mode):
    """'auto' | 'alone' | 'beside' (see above).  Returns the previous mode."""
    if mode not in ('auto', 'alone', 'beside'):
        raise ValueError("wgrad plan must be 'auto', 'alone' or 'beside'")
    prev, _PLAN['mode'] = _PLAN['mode'], mode
    _apply_wgrad_plan()
    return prev


def get_wgrad_plan():
    """The plan in force: 'alone' or 'beside' (what the bench line records as config.wgrad_plan)."""
    _apply_wgrad_plan()
    return 'beside' if _PLAN['applied'] else 'alone'


def join_side_stream(force=False):
    """Make the current stream wait for everything launched on the side stream so far; release the buffers held for it."""
    if force or _SIDE['keep'] or _SIDE['queued']:
        _lib.call('fva_side_stream_join', _stream())
        _SIDE['keep'].clear()
        _SIDE['queued'] = False


def fork_side_stream():
    """torch handle of the side stream, after making it wait for the current stream -- or None when the side stream is off.
    For consumers of a gradient that is still in flight there (the bucket copies / all-reduces of parallel.GradientReducer):
    work they enqueue on the handle runs after the weight gradients launched so far, without stalling the backward pass."""
    if not _SIDE['on']:
        return None
    side = C.c_void_p()
    _lib.call('fva_side_stream_fork', _stream(), C.byref(side))
    if _SIDE['torch'] is None or _SIDE['torch'].cuda_stream != side.value:
        _SIDE['torch'] = torch.cuda.ExternalStream(side.value)
    return _SIDE['torch']


def set_wgrad_side_stream(on):
    """Switch the side stream for weight gradients on or off (default on; env FVA_WGRAD_STREAM=0 turns it off).  Returns the
    previous setting.  Off = every kernel of the step runs on the caller's stream, one after the other (exclusive timings)."""
    prev = _SIDE['on']
    if prev and not on:
        join_side_stream(force=True)
    _SIDE['on'] = bool(on)
    _apply_wgrad_plan()
    return prev


def autotune_wgrad_side_stream(step, sync=None, steps=3, retries=2):
    """Time ``steps`` calls of ``step()`` with the weight gradients on the launch stream and on the side stream and keep the
    faster setting.  Whether two HIP streams really share the GPU to advantage depends on what else runs (with a host-staged
    gloo all-reduce the side stream LOSES: 108 vs 81 ms per step on a 2-rank test; with one GPU it wins by 4-9 %).  ``sync()``
    must drain the device (and, in a process group, be a barrier); in a process group the decision is taken on the slowest
    rank's times, so every rank switches the same way.  Returns {'off': ms, 'on': ms, 'use': bool}."""
    import time
    import torch.distributed as dist
    sync = sync or torch.cuda.synchronize
    if not _SIDE['on']:
        return {'use': False}

    def window():
        sync()
        t = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        w = (time.perf_counter() - t) / steps * 1e3
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            tw = torch.tensor([w], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            w = tw.item()
        return w
    out = {}
    for name, on in (('off', False), ('on', True)):
        set_wgrad_side_stream(on)
        step()
        out[name] = round(window(), 3)
    # A low-priority stream that landed badly among the hardware queues does not yield to the launch stream (32-35 ms per step
    # instead of 28.9 on some boxes / processes): ask the library for a fresh stream, twice at most, before giving the side stream up.
    tries = 0
    while out['on'] > out['off'] and tries < retries:
        tries += 1
        join_side_stream()
        sync()
        _lib.call('fva_side_stream_renew')
        step()
        t = round(window(), 3)
        out.setdefault('on_first_streams', []).append(out['on'])
        out['on'] = t
    out['use'] = out['on'] <= out['off']
    set_wgrad_side_stream(out['use'])
    return out


def hold_for_side_stream(*buffers):
    """Keep tensors that work on the side stream still reads or writes referenced until the next join (their memory would
    otherwise return to the allocator, which hands it to the next main-stream allocation)."""
    if _SIDE['on']:
        _SIDE['keep'].append(buffers)


def wgrad_stream(buffers, weight=None):
    """Stream argument for a weight-gradient launch (called from inside a backward pass).  A parameter that already holds a
    gradient (accumulation over micro-batches) gets its dW added by autograd on the main stream right after this backward
    returns, so that layer stays on the main stream; so does everything under create_graph."""
    if _PLAN['applied'] is None:
        _apply_wgrad_plan()
    if not _SIDE['on'] or torch.is_grad_enabled() or (weight is not None and (not weight.is_leaf or weight.grad is not None)):
        return _stream()          # (a derived weight, e.g. a channel-padded filter: autograd's next node reads dW on the main stream)
    side = C.c_void_p()
    try:
        _lib.call('fva_side_stream_fork', _stream(), C.byref(side))
    except RuntimeError:                          # e.g. a second device in this process: everything stays on the launch stream
        _SIDE['on'] = False
        _apply_wgrad_plan()
        return _stream()
    _SIDE['keep'].append(buffers)
    # one callback per launch, not one per backward pass: a pass that dies with an exception never runs its callbacks, and a
    # "queued" flag left behind by it would leave the next pass without a join (all but the first callback find nothing to do)
    _SIDE['queued'] = True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
    except RuntimeError:                          # not inside an engine-driven backward pass: the caller joins
        pass
    return side


def conv_block_bwd(s, dz_ptr, need_dx, addend_ptr=None):
    """Backward of conv_block_fwd given dz (dense NHWC, dtype).  Returns (dx_buf or None, dw, dgamma, dbeta)."""
    if not s.training:
        raise RuntimeError('fastvision_amd: backward through an eval-mode BatchNorm block is not supported')
    lib = _lib.load()
    d, dtype, dev = s.d, s.dtype, s.y.device
    Cout, code = d.Cout, _code(s.dtype)
    fused, s.fused = s.fused, None
    have = fused is not None and fused[2] == dz_ptr     # the consumer's dgrad epilogue has already summed dU and dU * xhat of exactly this dz
    dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
    dbeta = torch.empty_like(dgamma)
    dy = torch.empty((d.B, s.OH + 2, s.OW + 2, Cout), dtype=dtype, device=dev)
    st = getattr(s, 'acc', None)
    if st is not None:
        # accumulator form: the sums are in (or now go to) the layer's backward accumulator, and the second pass finalises them itself
        if not have:
            st.produce(1)                                # (also clears sums that a consumer added for another gradient buffer)
            _lib.call('fva_bn_silu_bwd_reduce_acc', code, C.c_void_p(dz_ptr), _p(s.y), _p(s.scale), _p(s.shift), _p(s.mean), _p(s.rstd),
                      _p(st.buf[1]), st.replicas, s.M, Cout, _stream())
        ba = _lib.BnBwdAcc(st.buf[1].data_ptr(), st.buf[0].data_ptr(), st.replicas, s.gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0)
        _lib.call('fva_bn_silu_bwd_apply_acc', code, C.c_void_p(dz_ptr), _p(s.y), _p(s.scale), _p(s.shift), _p(s.mean), _p(s.rstd),
                  C.byref(ba), _p(dy), 1, d.B, s.OH, s.OW, Cout, _stream())
        st.consumed(1)
    else:
        if have:
            part, nb = fused[0], fused[1]
        else:
            nb = lib.fva_bn_bwd_blocks(code, s.M, Cout)
            part = torch.empty((lib.fva_bn_partial_rows(nb), 2, Cout), dtype=torch.float32, device=dev)
            _lib.call('fva_bn_silu_bwd_reduce', code, C.c_void_p(dz_ptr), _p(s.y), _p(s.scale), _p(s.shift), _p(s.mean), _p(s.rstd),
                      _p(part), nb, s.M, Cout, _stream())
        coef = torch.empty((3, Cout), dtype=torch.float32, device=dev)
        _lib.call('fva_bn_bwd_finalize', _p(part), nb, part.shape[0], s.M, Cout, _p(s.gamma), _p(s.rstd), _p(dgamma), _p(dbeta), 0, _p(coef), _stream())
        _lib.call('fva_bn_silu_bwd_apply', code, C.c_void_p(dz_ptr), _p(s.y), _p(s.scale), _p(s.shift), _p(s.mean), _p(s.rstd),
                  _p(coef), _p(dy), 1, d.B, s.OH, s.OW, Cout, _stream())
    del fused
    dw = torch.empty(s.wshape, dtype=torch.float32, device=dev)
    ws_bytes = lib.fva_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(s.x_ptr), _p(dy), _p(dw), 0, _p(ws), ws_bytes, wgrad_stream((s.x, getattr(s, 'keep', None), dy, ws), s.weight))     # NOT dw: a second reference makes AccumulateGrad clone it (on the main stream)
    dx = None
    if need_dx:
        dx = torch.empty((d.B, d.H, d.W, d.Cin), dtype=dtype, device=dev)
        _dgrad(d, dy, s.wd, dx, addend_ptr, s.x_src, dtype)
    # this block's saved state is spent: drop what it references (its output tensor may outlive the step in the caller's hands and
    # still points here through _fva_prod; the side stream's launches keep their own references until the join)
    s.__dict__.clear()
    s.training, s.consumers, s.fused = False, 0, None
    return dx, dw, dgamma, dbeta


def _grad_like(dx_buf, x):
    """dense NHWC gradient buffer -> tensor shaped and typed like the forward input x."""
    g = dense_view(dx_buf)
    return g if g.dtype == x.dtype else g.to(x.dtype)


class ConvBNSiLUFn(torch.autograd.Function):
    """One ConvBlock3x3 / ConvBlock1x1 (reference classfication/models/darknet53.py:22-44)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, training, stride, dtype):
        require_gpu(x, 'ConvBlock')
        keep, x_ptr, x_pad = to_halo(x, dtype, weight.shape[2] // 2)
        need = _GRAD_ON[0] and any(ctx.needs_input_grad)
        z, s = conv_block_fwd(x, x_ptr, x_pad, weight, gamma, beta, bn, training, stride, dtype, need_ctx=need, may_defer=True)
        if s is not None:
            s.keep = keep
        ctx.s = s
        ctx.x_like = x
        return z

    @staticmethod
    def backward(ctx, dz):
        s = ctx.s
        if s is None:
            raise _released('ConvBlock')
        keep, dz_ptr = to_dense(dz, s.dtype)
        dx, dw, dg, db = conv_block_bwd(s, dz_ptr, ctx.needs_input_grad[0])
        x_like, ctx.s, ctx.x_like = ctx.x_like, None, None       # like torch's saved tensors: released by the backward pass that used them
        return (_grad_like(dx, x_like) if dx is not None else None), dw, dg, db, None, None, None, None


class StemFn(torch.autograd.Function):
    """conv0 of Darknet-53 on the caller's fp32 NCHW images (darknet53.py:73) + BN + SiLU."""

    @staticmethod
    def forward(ctx, images, weight, gamma, beta, bn, training, dtype):
        require_gpu(images, 'stem')
        flush_pending_apply()
        img = images.detach()
        if img.dtype != torch.float32 or not img.is_contiguous():
            img = img.float().contiguous()
        B, Cin, H, W = img.shape
        Cout = weight.shape[0]
        lib = _lib.load()
        dev, code = img.device, _code(dtype)
        M = B * H * W
        scale = torch.empty(Cout, dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        mean = rstd = None
        wsb = lib.fva_stem_fwd_workspace(code, B, H, W)           # bf16 NHWC4 copy of the images for the MFMA kernel
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        if ws is not None and os.environ.get('FVA_STEM_FUSED', '1') != '0':
            # bf16: conv0's pre-BN output is never stored; the statistics pass and the apply pass each recompute it on MFMA
            # from the packed image (839 MB of HBM traffic saved per pass at B = 32, 640 px)
            _lib.call('fva_stem_pack', _p(img), _p(ws), wsb, B, Cin, H, W, _stream())
            if training:
                nblk = lib.fva_stem_fused_blocks(B, H, W)
                stats = torch.empty((lib.fva_bn_partial_rows(nblk), 2, Cout), dtype=torch.float32, device=dev)
                _lib.call('fva_stem_fused', 0, _p(ws), _p(weight), None, None, None, None, None, None, None, _p(stats), B, Cin, H, W, _stream())
                mean, rstd = torch.empty_like(scale), torch.empty_like(scale)
                _lib.call('fva_bn_finalize', _p(stats), nblk, stats.shape[0], M, Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), _p(bn.nbt),
                          bn.momentum, bn.eps, _p(mean), _p(rstd), _p(scale), _p(shift), _stream())
            else:
                _lib.call('fva_bn_eval_coeffs', Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), bn.eps, _p(scale), _p(shift), _stream())
            zbuf, z = halo_alloc(B, Cout, H, W, dtype, dev, 1)
            _lib.call('fva_stem_fused', 1, _p(ws), _p(weight), None, _p(scale), _p(shift), None, None, None, _p(zbuf), None, B, Cin, H, W,
                      _stream())
            ctx.saved = (img, None, scale, shift, mean, rstd, gamma, dtype, training, tuple(weight.shape))
            ctx.img4, ctx.weight = ws, weight.detach()
            return z
        y = torch.empty((M, Cout), dtype=dtype, device=dev)
        if training:
            nblk = lib.fva_stem_stat_blocks(code, B, H, W)
            stats = torch.empty((lib.fva_bn_partial_rows(nblk), 2, Cout), dtype=torch.float32, device=dev)
            _lib.call('fva_stem_fwd', code, _p(img), _p(weight), _p(y), _p(stats), _p(ws), wsb, B, Cin, H, W, Cout, _stream())
            mean, rstd = torch.empty_like(scale), torch.empty_like(scale)
            _lib.call('fva_bn_finalize', _p(stats), nblk, stats.shape[0], M, Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), _p(bn.nbt),
                      bn.momentum, bn.eps, _p(mean), _p(rstd), _p(scale), _p(shift), _stream())
        else:
            _lib.call('fva_stem_fwd', code, _p(img), _p(weight), _p(y), C.c_void_p(0), _p(ws), wsb, B, Cin, H, W, Cout, _stream())
            _lib.call('fva_bn_eval_coeffs', Cout, _p(gamma), _p(beta), _p(bn.rm), _p(bn.rv), bn.eps, _p(scale), _p(shift), _stream())
        zbuf, z = halo_alloc(B, Cout, H, W, dtype, dev, 1)
        _lib.call('fva_bn_silu_apply', code, _p(y), _p(scale), _p(shift), C.c_void_p(0), 0, _p(zbuf), 1, B, H, W, Cout, _stream())
        ctx.saved = (img, y, scale, shift, mean, rstd, gamma, dtype, training, tuple(weight.shape))
        ctx.img4 = ws                     # bf16 NHWC4 copy of the images (MFMA path): the weight gradient reads it again
        return z

    @staticmethod
    def backward(ctx, dz):
        if ctx.saved is None:
            raise _released('stem')
        img, y, scale, shift, mean, rstd, gamma, dtype, training, wshape = ctx.saved
        img4, w_saved = ctx.img4, getattr(ctx, 'weight', None)
        ctx.saved = ctx.img4 = ctx.weight = None                  # released by the backward pass that uses them
        if not training:
            raise RuntimeError('fastvision_amd: backward through an eval-mode BatchNorm block is not supported')
        if ctx.needs_input_grad[0]:
            raise RuntimeError('fastvision_amd: the stem does not produce a gradient for the input images')
        lib = _lib.load()
        B, Cin, H, W = img.shape
        Cout, code, dev = wshape[0], _code(dtype), img.device
        M = B * H * W
        keep, dz_ptr = to_dense(dz, dtype)
        if y is None:
            # fused bf16 path: both BatchNorm-backward passes recompute conv0 from the packed image, dY lands in a halo buffer
            w = w_saved
            nb = lib.fva_stem_fused_blocks(B, H, W)
            part = torch.empty((lib.fva_bn_partial_rows(nb), 2, Cout), dtype=torch.float32, device=dev)
            _lib.call('fva_stem_fused', 2, _p(img4), _p(w), C.c_void_p(dz_ptr), _p(scale), _p(shift), _p(mean), _p(rstd), None, None,
                      _p(part), B, Cin, H, W, _stream())
            dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
            dbeta = torch.empty_like(dgamma)
            coef = torch.empty((3, Cout), dtype=torch.float32, device=dev)
            _lib.call('fva_bn_bwd_finalize', _p(part), nb, part.shape[0], M, Cout, _p(gamma), _p(rstd), _p(dgamma), _p(dbeta), 0, _p(coef), _stream())
            dy = torch.empty((B, H + 2, W + 2, Cout), dtype=dtype, device=dev)
            _lib.call('fva_stem_fused', 3, _p(img4), _p(w), C.c_void_p(dz_ptr), _p(scale), _p(shift), _p(mean), _p(rstd), _p(coef),
                      _p(dy), None, B, Cin, H, W, _stream())
            raw = torch.empty((Cout, 4, 4, 3), dtype=torch.float32, device=dev)        # [co][kw][ci][kh]
            wsb = lib.fva_stem_wgrad_mfma_workspace()
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.call('fva_stem_wgrad_mfma', _p(img4), _p(dy), _p(raw), _p(ws), wsb, B, H, W, _stream())
            return None, raw[:, :3, :Cin, :].permute(0, 2, 3, 1).contiguous(), dgamma, dbeta, None, None, None
        nb = lib.fva_bn_bwd_blocks(code, M, Cout)
        part = torch.empty((lib.fva_bn_partial_rows(nb), 2, Cout), dtype=torch.float32, device=dev)
        _lib.call('fva_bn_silu_bwd_reduce', code, C.c_void_p(dz_ptr), _p(y), _p(scale), _p(shift), _p(mean), _p(rstd), _p(part),
                  nb, M, Cout, _stream())
        dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
        dbeta = torch.empty_like(dgamma)
        coef = torch.empty((3, Cout), dtype=torch.float32, device=dev)
        _lib.call('fva_bn_bwd_finalize', _p(part), nb, part.shape[0], M, Cout, _p(gamma), _p(rstd), _p(dgamma), _p(dbeta), 0, _p(coef), _stream())
        if img4 is not None and os.environ.get('FVA_STEM_WGRAD_MFMA', '1') != '0':
            # MFMA path: dY as a halo buffer, conv0 seen as 3 vertical taps over 4-pixel windows of the NHWC4 image
            dy = torch.empty((B, H + 2, W + 2, Cout), dtype=dtype, device=dev)
            _lib.call('fva_bn_silu_bwd_apply', code, C.c_void_p(dz_ptr), _p(y), _p(scale), _p(shift), _p(mean), _p(rstd), _p(coef),
                      _p(dy), 1, B, H, W, Cout, _stream())
            raw = torch.empty((Cout, 4, 4, 3), dtype=torch.float32, device=dev)        # [co][kw][ci][kh]
            wsb = lib.fva_stem_wgrad_mfma_workspace()
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.call('fva_stem_wgrad_mfma', _p(img4), _p(dy), _p(raw), _p(ws), wsb, B, H, W, _stream())
            dw = raw[:, :3, :Cin, :].permute(0, 2, 3, 1).contiguous()
            return None, dw, dgamma, dbeta, None, None, None
        dy = torch.empty((M, Cout), dtype=dtype, device=dev)
        _lib.call('fva_bn_silu_bwd_apply', code, C.c_void_p(dz_ptr), _p(y), _p(scale), _p(shift), _p(mean), _p(rstd), _p(coef),
                  _p(dy), 0, B, H, W, Cout, _stream())
        dw = torch.empty(wshape, dtype=torch.float32, device=dev)
        wsb = lib.fva_stem_wgrad_workspace(B, Cin, H, W, Cout)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call('fva_stem_wgrad', code, _p(img), _p(dy), _p(dw), 0, _p(ws), wsb, B, Cin, H, W, Cout, _stream())
        return None, dw, dgamma, dbeta, None, None, None


class ResidualFn(torch.autograd.Function):
    """x + CB3x3(CB1x1(x)) as one node (darknet53.py:46-63): the add is fused into the second block's
    apply pass, its gradient sum into the first block's dgrad epilogue."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, bn1, w2, g2, b2, bn2, training, dtype):
        require_gpu(x, 'ResidualBlock')
        keep, x_ptr, x_pad = to_halo(x, dtype, 1)
        need = _GRAD_ON[0] and any(ctx.needs_input_grad)
        z1, s1 = conv_block_fwd(x, x_ptr, x_pad, w1, g1, b1, bn1, training, 1, dtype, need_ctx=need)
        z1_ptr, z1_pad = halo_info(z1, dtype)
        z2, s2 = conv_block_fwd(z1, z1_ptr, z1_pad, w2, g2, b2, bn2, training, 1, dtype, residual=(x_ptr, x_pad), need_ctx=need, may_defer=True)
        if need:
            s1.keep = keep
        ctx.s1, ctx.s2, ctx.x_like = s1, s2, x
        return z2

    @staticmethod
    def backward(ctx, dout):
        s1, s2 = ctx.s1, ctx.s2
        if s1 is None:
            raise _released('ResidualBlock')
        keep, dout_ptr = to_dense(dout, s1.dtype)
        dz1, dw2, dg2, db2 = conv_block_bwd(s2, dout_ptr, True)
        dx, dw1, dg1, db1 = conv_block_bwd(s1, dz1.data_ptr(), ctx.needs_input_grad[0], addend_ptr=dout_ptr)
        gx = _grad_like(dx, ctx.x_like) if dx is not None else None
        ctx.s1 = ctx.s2 = ctx.x_like = None                       # released by the backward pass that used them
        return gx, dw1, dg1, db1, None, dw2, dg2, db2, None, None, None


class UpsampleConcatFn(torch.autograd.Function):
    """cat(upsample2(up), skip) (library order, yolov3neck.py:105,110) or cat(skip, upsample2(up)) (demo order)."""

    @staticmethod
    def forward(ctx, up, skip, up_first, dtype):
        require_gpu(up, 'UpSampling')
        flush_pending_apply()
        _note_consumer(up)            # their gradients come from upcat_bwd, not from a dgrad epilogue
        _note_consumer(skip)
        ku, up_ptr, up_pad = to_halo(up, dtype, 0)
        ks, sk_ptr, sk_pad = to_halo(skip, dtype, 0)
        B, Cup, h, w = up.shape
        Cs = skip.shape[1]
        if tuple(skip.shape) != (B, Cs, 2 * h, 2 * w):
            raise RuntimeError(f'upsample/concat: skip {tuple(skip.shape)} does not match 2x of {tuple(up.shape)}')
        buf, out = halo_alloc(B, Cup + Cs, 2 * h, 2 * w, dtype, up.device, 1)
        _lib.call('fva_upsample2_concat_fwd', _code(dtype), C.c_void_p(up_ptr), up_pad, C.c_void_p(sk_ptr), sk_pad, _p(buf),
                  B, h, w, Cup, Cs, 1 if up_first else 0, _stream())
        ctx.meta = (B, h, w, Cup, Cs, up_first, dtype, up.dtype, skip.dtype)
        return out

    @staticmethod
    def backward(ctx, dcat):
        B, h, w, Cup, Cs, up_first, dtype, udt, sdt = ctx.meta
        keep, dptr = to_dense(dcat, dtype)
        dup = torch.empty((B, h, w, Cup), dtype=dtype, device=dcat.device)
        dsk = torch.empty((B, 2 * h, 2 * w, Cs), dtype=dtype, device=dcat.device)
        _lib.call('fva_upsample2_concat_bwd', _code(dtype), C.c_void_p(dptr), _p(dup), _p(dsk), B, h, w, Cup, Cs,
                  1 if up_first else 0, _stream())
        gu, gs = dense_view(dup), dense_view(dsk)
        return (gu if gu.dtype == udt else gu.to(udt)), (gs if gs.dtype == sdt else gs.to(sdt)), None, None


NPAD = 256  # head channels (255) padded for the MFMA backward kernels


class HeadFn(torch.autograd.Function):
    """Biased 1x1 conv to A*(5+C) channels (yolov3head.py:50,60).  Returns the fp32 buffer [B,H,W,N]."""

    @staticmethod
    def forward(ctx, x, weight, bias, dtype):
        require_gpu(x, 'head')
        flush_pending_apply()
        ctx.x_src = _note_consumer(x) if ctx.needs_input_grad[0] else None
        keep, x_ptr, x_pad = to_halo(x, dtype, 0)
        B, Cin, H, W = x.shape
        N = weight.shape[0]
        d = ConvDesc(_code(dtype), B, H, W, Cin, N, 1, 1, x_pad, 1)
        wf, _ = packed_weights(weight, d, dtype)
        out = torch.empty((B, H, W, N), dtype=torch.float32, device=x.device)
        _lib.call('fva_head_fwd', C.byref(d), C.c_void_p(x_ptr), _p(wf), _p(bias), _p(out), _stream())
        ctx.saved = (keep, x_ptr, x_pad, weight, dtype, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise _released('head')
        keep, x_ptr, x_pad, weight, dtype, x_like = ctx.saved
        ctx.saved = None                                          # released by the backward pass that uses them
        lib = _lib.load()
        B, Cin, H, W = x_like.shape
        N = weight.shape[0]
        npad = (N + 63) // 64 * 64
        dev, code = dout.device, _code(dtype)
        if dout.dtype != torch.float32 or not dout.is_contiguous():
            dout = dout.float().contiguous()
        dy = torch.empty((B, H + 2, W + 2, npad), dtype=dtype, device=dev)
        dbias = torch.empty(N, dtype=torch.float32, device=dev)
        ws = torch.empty(4096 * N, dtype=torch.float32, device=dev)
        _lib.call('fva_head_bwd_prepare', code, _p(dout), C.c_void_p(0), _p(dy), _p(dbias), 0, _p(ws), B, H, W, N, npad, _stream())
        # weight padded to npad rows so that dgrad / wgrad run as an ordinary Cout = npad convolution
        wpad = torch.zeros((npad, Cin, 1, 1), dtype=torch.float32, device=dev)
        wpad[:N].copy_(weight.detach())
        d = ConvDesc(code, B, H, W, Cin, npad, 1, 1, x_pad, 1)
        _, wd = packed_weights(wpad, d, dtype, cache=False)
        dwp = torch.empty((npad, Cin, 1, 1), dtype=torch.float32, device=dev)
        wsb = lib.fva_conv_wgrad_workspace(C.byref(d))
        wsw = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call('fva_conv_wgrad', C.byref(d), C.c_void_p(x_ptr), _p(dy), _p(dwp), 0, _p(wsw), wsb, _stream())
        dx = None
        if ctx.needs_input_grad[0]:
            dxb = torch.empty((B, H, W, Cin), dtype=dtype, device=dev)
            _dgrad(d, dy, wd, dxb, None, ctx.x_src, dtype)
            ctx.x_src = None
            dx = _grad_like(dxb, x_like)
        return dx, dwp[:N].contiguous(), dbias, None


# ------------------------------------------------------------------------------------------------ thin functional API
# autograd.Function.forward always runs with grad mode off, and ctx.needs_input_grad ignores torch.no_grad(): the wrappers
# below note the caller's grad mode here so that inference (no_grad) takes the context-free fused path
_GRAD_ON = [True]


def conv_bn_silu(x, conv, bn, stride=None, dtype=None):
    dtype = dtype or get_compute_dtype()
    stride = conv.stride[0] if stride is None else stride
    _GRAD_ON[0] = torch.is_grad_enabled()
    return ConvBNSiLUFn.apply(x, conv.weight, bn.weight, bn.bias, _BNState(bn), bn.training, stride, dtype)


def stem(images, conv, bn, dtype=None):
    dtype = dtype or get_compute_dtype()
    return StemFn.apply(images, conv.weight, bn.weight, bn.bias, _BNState(bn), bn.training, dtype)


def residual(x, cb1, cb2, dtype=None):
    dtype = dtype or get_compute_dtype()
    _GRAD_ON[0] = torch.is_grad_enabled()
    return ResidualFn.apply(x, cb1.conv.weight, cb1.bn.weight, cb1.bn.bias, _BNState(cb1.bn),
                            cb2.conv.weight, cb2.bn.weight, cb2.bn.bias, _BNState(cb2.bn), cb1.bn.training, dtype)


def upsample2_concat(up, skip, up_first, dtype=None):
    return UpsampleConcatFn.apply(up, skip, up_first, dtype or get_compute_dtype())


def head_conv(x, conv, dtype=None):
    return HeadFn.apply(x, conv.weight, conv.bias, dtype or get_compute_dtype())
