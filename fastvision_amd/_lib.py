"""ctypes binding of libfastvision_amd.so (include/fastvision_amd.h).  Fails loudly: no CPU fallback.

The library is built in-tree by ``python fastvision_amd/csrc/build.py`` (or ``__graft_entry__.build()``).
Every wrapper raises ``RuntimeError`` with ``fva_last_error()`` when the C call returns non-zero.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'csrc', 'libfastvision_amd.so')
if os.environ.get('FVA_LIB_PATH'):
    # kernel A/Bs against another build of the same sources: an explicit opt-in, announced (it replaces the whole library)
    if os.environ.get('FVA_EXPERIMENTS') == '1':
        LIB_PATH = os.environ['FVA_LIB_PATH']
        import sys
        print(f'fastvision_amd: EXPERIMENT -- loading {LIB_PATH} instead of the in-tree library (FVA_LIB_PATH)', file=sys.stderr)
    else:
        raise RuntimeError('fastvision_amd: FVA_LIB_PATH is set but FVA_EXPERIMENTS=1 is not -- refusing to load another library silently')

F32, BF16 = 0, 1


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('dtype', 'B', 'H', 'W', 'Cin', 'Cout', 'ksize', 'stride', 'in_pad', 'dy_pad')]


class HeadLevel(C.Structure):
    _fields_ = [('data', C.c_void_p), ('grad', C.c_void_p),
                ('sb', C.c_int64), ('sa', C.c_int64), ('sy', C.c_int64), ('sx', C.c_int64), ('sk', C.c_int64),
                ('B', C.c_int32), ('A', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('K', C.c_int32),
                ('anchor_w', C.c_float * 8), ('anchor_h', C.c_float * 8), ('stride', C.c_float)]


class PackEntry(C.Structure):
    _fields_ = [('w', C.c_void_p), ('w_fwd', C.c_void_p), ('w_dgrad', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('Cout', 'Cin', 'ksize', 'taps_fwd', 'taps_dgrad', 'dtype', 'dgrad_paired', 'tile_start')]


class Letterbox(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('resize_ratio', 'pad_left', 'pad_top', 'ori_w', 'ori_h', 'min_wh')]


class NmsParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('box_mode', 'score_mode', 'rethreshold', 'max_det', 'max_nms')] + \
               [(n, C.c_float) for n in ('conf_thres', 'iou_thres', 'class_gap')]


class PasteJob(C.Structure):
    _fields_ = [('src_offset', C.c_int64)] + [(n, C.c_int32) for n in ('src_h', 'src_w', 'dst_h', 'dst_w', 'top', 'left', 'flip_h', 'flip_v', 'src_pitch', 'reserved')] + \
               [('scale_x', C.c_double), ('scale_y', C.c_double)]


class BnBwdFuse(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('y', 'scale', 'shift', 'mean', 'rstd', 'partial', 'acc')] + [('acc_replicas', C.c_int32)]


class BnFwdAcc(C.Structure):
    """fva_bn_fwd_acc: BatchNorm statistics in a fixed-point accumulator, finalised by the consuming launch."""
    _fields_ = [('acc', C.c_void_p), ('zero', C.c_void_p), ('replicas', C.c_int32)] + [(n, C.c_void_p) for n in ('gamma', 'beta', 'running_mean', 'running_var', 'num_batches_tracked')] + \
               [('momentum', C.c_float), ('eps', C.c_float)] + [(n, C.c_void_p) for n in ('save_mean', 'save_rstd', 'scale', 'shift')]


class BnBwdAcc(C.Structure):
    """fva_bn_bwd_acc: the backward sums (dU, dU * xhat) in an accumulator, finalised by the second backward pass."""
    _fields_ = [('acc', C.c_void_p), ('zero', C.c_void_p), ('replicas', C.c_int32)] + [(n, C.c_void_p) for n in ('gamma', 'dgamma', 'dbeta')] + [('accumulate', C.c_int32)]


class ColourJob(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('h', 'w', 'clahe', 'hsv', 'blur')] + [('perm', C.c_int32 * 3)]


class MatchOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('count', 'b', 'gx', 'gy', 'a', 'cls', 'xywh', 'anc')]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_D = C.POINTER(ConvDesc)
_H = C.POINTER(HeadLevel)

# name -> (restype, argtypes); status-returning functions have restype int and are checked
PROTOTYPES = {
    'fva_last_error': (C.c_char_p, []),
    'fva_version': (_I, []),
    'fva_profile_start': (_I, [_I]),
    'fva_profile_classes': (_I, [C.c_uint32, _I]),
    'fva_side_stream_fork': (_I, [_P, C.POINTER(C.c_void_p)]),
    'fva_side_stream_join': (_I, [_P]),
    'fva_side_stream_renew': (_I, []),
    'fva_conv_debug_stamps': (_I, [_P, _I]),
    'fva_conv_patch_kernel': (_I, [_I]),
    'fva_profile_stop': (_I, [_P, _P, _P, _I]),
    'fva_conv_pack_weights': (_I, [_D, _P, _P, _P, _P]),
    'fva_conv_packed_elems': (_L, [_D, _I]),
    'fva_conv_pack_weights_multi': (_I, [_P, _I, _L, _P]),
    'fva_conv_pack_weights_tiled': (_I, [_P, _I, _I, _P]),
    'fva_conv_last_kernel': (C.c_char_p, []),
    'fva_conv_fwd': (_I, [_D, _P, _P, _P, _P, _P]),
    'fva_conv1x1_fwd_apply': (_I, [_D, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    'fva_conv_fwd_acc': (_I, [_D, _P, _P, _P, _P, _I, _P]),
    'fva_bn_acc_finalize': (_I, [C.POINTER(BnFwdAcc), _L, _I, _P]),
    'fva_bn_silu_apply_acc': (_I, [_I, _P, C.POINTER(BnFwdAcc), _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'fva_conv1x1_fwd_apply_acc': (_I, [_D, _P, C.POINTER(BnFwdAcc), _P, _I, _P, _P, _P, _P, _I, _P]),
    'fva_conv_fwd_bnact': (_I, [_D, _P, _P, _P, _P, _P, _P, _I, _P]),
    'fva_conv_stat_blocks': (_I, [_D]),
    'fva_conv_dgrad': (_I, [_D, _P, _P, _P, _P, _P]),
    'fva_conv_dgrad_bnstats': (_I, [_D, _P, _P, _P, _P, C.POINTER(BnBwdFuse), _P]),
    'fva_conv_dgrad_stat_rows': (_I, [_D]),
    'fva_conv_wgrad': (_I, [_D, _P, _P, _P, _I, _P, _L, _P]),
    'fva_conv_wgrad_workspace': (_L, [_D]),
    'fva_conv_wgrad_plan': (_I, [_I]),
    'fva_stem_fwd': (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    'fva_stem_fwd_workspace': (_L, [_I, _I, _I, _I]),
    'fva_stem_stat_blocks': (_I, [_I, _I, _I, _I]),
    'fva_stem_fused_blocks': (_I, [_I, _I, _I]),
    'fva_stem_wgrad': (_I, [_I, _P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P]),
    'fva_stem_wgrad_workspace': (_L, [_I, _I, _I, _I, _I]),
    'fva_stem_wgrad_mfma': (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _P]),
    'fva_stem_wgrad_mfma_workspace': (_L, []),
    'fva_stem_pack': (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    'fva_stem_fused': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'fva_head_fwd': (_I, [_D, _P, _P, _P, _P, _P]),
    'fva_head_bwd_prepare': (_I, [_I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'fva_bn_finalize': (_I, [_P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    'fva_bn_partial_rows': (_I, [_I]),
    'fva_bn_eval_coeffs': (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    'fva_bn_silu_apply': (_I, [_I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'fva_bn_silu_bwd_reduce': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P]),
    'fva_bn_silu_bwd_reduce_acc': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P]),
    'fva_bn_bwd_blocks': (_I, [_I, _L, _I]),
    'fva_bn_bwd_finalize': (_I, [_P, _I, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P]),
    'fva_bn_silu_bwd_apply': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'fva_bn_silu_bwd_apply_acc': (_I, [_I, _P, _P, _P, _P, _P, _P, C.POINTER(BnBwdAcc), _P, _I, _I, _I, _I, _I, _P]),
    'fva_upsample2_concat_fwd': (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'fva_upsample2_concat_bwd': (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'fva_pack_nchw': (_I, [_I, _P, _I, _L, _L, _L, _L, _P, _I, _I, _I, _I, _I, _P]),
    'fva_cast_nhwc': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'fva_yolov3_match': (_I, [_P, _I, _H, C.POINTER(MatchOut), _P]),
    'fva_yolov3_loss': (_I, [_P, _I, _H, _I, _F, _F, _F, _P, _P, _L, _P]),
    'fva_yolov3_loss_workspace': (_L, [_I, _H, _I]),
    'fva_yolov3_loss_dp': (_I, [_P, _I, _H, _I, _F, _F, _F, _P, _I, _P, _P, _L, _P]),
    'fva_scale_by_device_scalar': (_I, [_P, _L, _P, _P]),
    'fva_bce_loss': (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _P]),
    'fva_row_loss': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    'fva_smooth_l1': (_I, [_P, _P, _L, _P, _P, _P, _P]),
    'fva_rows_relu_bwd_rows': (_I, [_I]),
    'fva_rows_relu_bwd': (_I, [_I, _P, _P, _P, _L, _P, _I, _I, _I, _P]),
    'fva_demo_loss': (_I, [_P, _I, _H, _I, _P, _P, _L, _P]),
    'fva_demo_loss_workspace': (_L, [_I, _H, _I]),
    'fva_iou_pairwise': (_I, [_I, _I, _I, _P, _P, _P, _P, _L, _F, _P]),
    'fva_iou_batch': (_I, [_I, _I, _I, _P, _P, _P, _L, _L, _F, _P]),
    'fva_adam_step': (_I, [_P, _P, _I, _L, _F, _F, _F, _F, _F, _L, _F, _P]),
    'fva_adam_step_dev': (_I, [_P, _P, _I, _L, _P, _F, _F, _F, _F, _P, _F, _P]),
    'fva_gather_cast': (_I, [_P, _I, _L, _P, _I, _P]),
    'fva_paste_resize_normalize': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'fva_paste_resize_u8': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'fva_colour_workspace': (_L, [_I]),
    'fva_colour_clahe_hsv': (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    'fva_colour_blur_shuffle_normalize': (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    'fva_yolo_decode': (_I, [_H, _I, _I, C.POINTER(Letterbox), _P, _L, _P]),
    'fva_nms_candidates_workspace': (_L, [_I, _I]),
    'fva_nms_candidates': (_I, [_P, _I, _I, _I, C.POINTER(NmsParams), _P, _L, _P, _P]),
    'fva_nms_select_workspace': (_L, [_I, _I]),
    'fva_conv_fwd_bias_act': (_I, [_D, _P, _P, _P, _I, _P, _I, _P]),
    'fva_bias_relu_bwd_rows': (_I, [_I, _I, _I]),
    'fva_bias_relu_bwd': (_I, [_I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    'fva_colsum_scratch_rows': (_I, [_I]),
    'fva_colsum': (_I, [_P, _I, _I, _P, _P, _P]),
    'fva_maxpool2_fwd': (_I, [_I, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'fva_maxpool2_bwd': (_I, [_I, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    'fva_rpn_decode': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'fva_rpn_match': (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    'fva_fast_match': (_I, [_P, _I, _P, _I, _I, _F, _F, _F, _P, _P]),
    'fva_roi_align_fwd': (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    'fva_roi_align_bwd': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    'fva_nms_select': (_I, [_P, _P, _I, _I, _I, C.POINTER(NmsParams), _P, _L, _P, _P, _P, _P]),
}
UNCHECKED = {'fva_conv_patch_kernel', 'fva_rows_relu_bwd_rows', 'fva_colour_workspace', 'fva_conv_dgrad_stat_rows', 'fva_bias_relu_bwd_rows', 'fva_colsum_scratch_rows', 'fva_last_error', 'fva_version', 'fva_profile_stop', 'fva_conv_last_kernel', 'fva_conv_packed_elems', 'fva_conv_stat_blocks', 'fva_conv_wgrad_workspace', 'fva_conv_wgrad_plan',
             'fva_stem_stat_blocks', 'fva_stem_fused_blocks', 'fva_stem_wgrad_workspace', 'fva_stem_fwd_workspace', 'fva_stem_wgrad_mfma_workspace', 'fva_bn_bwd_blocks', 'fva_bn_partial_rows', 'fva_yolov3_loss_workspace',
             'fva_demo_loss_workspace', 'fva_nms_candidates_workspace', 'fva_nms_select_workspace'}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is missing: build it first."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'fastvision_amd: {LIB_PATH} is missing -- the HIP extension is REQUIRED (no CPU fallback). '
                           'Build it with `python fastvision_amd/csrc/build.py`.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


tracer = None   # set by fastvision_amd.profiler.KernelTimer: callable(name, args) -> context manager or None


_fns = {}


def call(name, *args):
    """Call a status-returning entry point; raise RuntimeError(fva_last_error()) on failure."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    lib = _lib
    if tracer is not None:
        span = tracer(name, args)
        if span is not None:
            with span:
                rc = fn(*args)
        else:
            rc = fn(*args)
    else:
        rc = fn(*args)
    if name not in UNCHECKED and rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib.fva_last_error().decode()}')
    return rc
