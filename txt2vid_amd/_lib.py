"""ctypes binding of libt2v_hip.so (C ABI declared in include/t2v_hip.h).

The library holds every hand-written gfx950 kernel of the hot path. There is NO fallback: if the
shared object is missing the import of this module raises, and every product op that reaches
`lib()` on a machine without it fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('T2V_LIB') or os.path.join(_HERE, 'csrc', 'libt2v_hip.so')     # T2V_LIB: developer A/B builds only

MAX_TAPS = 27


class ConvGeom(C.Structure):
    """Mirror of `t2v_conv_geom` (include/t2v_hip.h)."""
    _fields_ = [('N', C.c_int32), ('Cin', C.c_int32), ('D', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Cout', C.c_int32), ('ntaps', C.c_int32),
                ('dz', C.c_int8 * MAX_TAPS), ('dy', C.c_int8 * MAX_TAPS), ('dx', C.c_int8 * MAX_TAPS),
                ('pad_', C.c_int8 * 3)]


class ConvGroup(C.Structure):
    """Mirror of `t2v_conv_group` (include/t2v_hip.h)."""
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('mask', C.c_void_p), ('N', C.c_int32), ('D', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('ntaps', C.c_int32), ('dstride', C.c_int32), ('ydstride', C.c_int32), ('yoff', C.c_int32), ('Dy', C.c_int32),
                ('dz', C.c_int8 * MAX_TAPS), ('dy', C.c_int8 * MAX_TAPS), ('dx', C.c_int8 * MAX_TAPS), ('widx', C.c_int8 * MAX_TAPS)]


class AdamJob(C.Structure):
    """t2v_adam_job (include/t2v_hip.h)."""
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64)]


class PoolJob(C.Structure):
    """t2v_pool_job (include/t2v_hip.h)."""
    _fields_ = [('x', C.c_void_p), ('x2', C.c_void_p), ('y', C.c_void_p), ('add', C.c_void_p), ('NC', C.c_int32), ('D', C.c_int32), ('H', C.c_int32),
                ('W', C.c_int32), ('Do', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32), ('k', C.c_int32 * 3), ('s', C.c_int32 * 3),
                ('p', C.c_int32 * 3)]


class PoolBoxJob(C.Structure):
    """t2v_poolbox_job (include/t2v_hip.h)."""
    _fields_ = [('in_', C.c_void_p), ('mask', C.c_void_p), ('out', C.c_void_p), ('bias', C.c_void_p), ('NC', C.c_int32), ('D', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('tmode', C.c_int32), ('relu', C.c_int32), ('scale', C.c_float), ('C', C.c_int32)]


class MultiJob(C.Structure):
    """t2v_multi_job (include/t2v_hip.h)."""
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('c', C.c_void_p), ('out', C.c_void_p), ('out2', C.c_void_p), ('n', C.c_int64),
                ('d0', C.c_int32), ('d1', C.c_int32), ('d2', C.c_int32), ('f0', C.c_int32), ('f1', C.c_int32), ('reserved', C.c_int32)]


class PackJob(C.Structure):
    """Mirror of `t2v_pack_job` (include/t2v_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('Cout', C.c_int32), ('Cin', C.c_int32), ('T', C.c_int32),
                ('ntaps', C.c_int32), ('mode', C.c_int32), ('dst_rows', C.c_int32), ('dst_cols', C.c_int32),
                ('row_off', C.c_int32), ('col_off', C.c_int32), ('block_begin', C.c_int32), ('bx', C.c_int32), ('by', C.c_int32),
                ('taps', C.c_int8 * MAX_TAPS), ('pad_', C.c_int8 * 5)]


WGRAD_MAX_SRC = 6


class WgradSrc(C.Structure):
    """Mirror of `t2v_wgrad_src` (include/t2v_hip.h)."""
    _fields_ = [('slab', C.c_void_p), ('bias_slab', C.c_void_p), ('ntaps', C.c_int32), ('S', C.c_int32),
                ('map', C.c_int8 * MAX_TAPS), ('pad_', C.c_int8 * 5), ('tap_stride', C.c_int64), ('split_stride', C.c_int64)]


class WgradDest(C.Structure):
    """Mirror of `t2v_wgrad_dest` (include/t2v_hip.h)."""
    _fields_ = [('dw', C.c_void_p), ('dbias', C.c_void_p), ('CoCi', C.c_int64), ('T', C.c_int32), ('Cout', C.c_int32),
                ('nsrc', C.c_int32), ('accum', C.c_int32), ('accum_bias', C.c_int32), ('kind', C.c_int32),
                ('block_begin', C.c_int32), ('nblocks', C.c_int32), ('tap_major', C.c_int32), ('pad_', C.c_int32),
                ('src', WgradSrc * WGRAD_MAX_SRC)]


MAX_GROUPS = 8
_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_I3 = C.POINTER(C.c_int32)
_G = C.POINTER(ConvGeom)

# name -> argtypes (restype is int unless stated); must list EVERY symbol of include/t2v_hip.h
SIGNATURES = {
    't2v_pack_weight': [_P, _P, _I, _I, _I, _I3, _I, _I, _P],
    't2v_pack_job_bytes': [],
    't2v_pack_multi': [_P, _I, _I, _P],
    't2v_pack_weight_into': [_P, _P, _I, _I, _I, _I3, _I, _I, _I, _I, _I, _I, _P],
    't2v_conv_fwd': [_P, _P, _P, _P, _P, _G, _I, _P],
    't2v_conv_fwd_ws_floats': [_G],
    't2v_conv_wgrad_slab_floats': [_G, _I],
    't2v_conv_fwd_grouped_ws_floats': [_P, _I, _I, _I],
    't2v_conv_fwd_grouped': [_P, _I, _I, _I, _P, _P, _P, _I, _P],
    't2v_conv_wgrad_grouped_slab_floats': [_P, _I, _I, _I, _I, _I, _I],
    't2v_conv_fwd_plan': [_P, _I, _I, _I, _I, _I3],
    't2v_nonlocal_ok': [_I, _I],
    't2v_nonlocal_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    't2v_nonlocal_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    't2v_conv_wgrad_grouped_partial': [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    't2v_wgrad_dest_bytes': [],
    't2v_wgrad_reduce_multi': [_P, _I, _I, _P],
    't2v_conv_wgrad_plan': [_P, _I, _I, _I, _I, _I, _I, _I3],
    't2v_pack_weight_bf16': [_P, _P, _I, _I, _I, _P, _I, _I, _P],
    't2v_conv_fwd_grouped_bf16_ok': [_P, _I, _I, _I],
    't2v_conv_fwd_bf16_plan': [_P, _I, _I, _I, _I, _I3],
    't2v_conv_fwd_grouped_bf16': [_P, _I, _I, _I, _P, _P, _P, _I, _P],
    't2v_conv_wgrad_grouped': [_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    't2v_conv_wgrad_grouped_bias_slab_floats': [_P, _I, _I, _I, _I, _I, _I],
    't2v_conv_wgrad_grouped_bias': [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    't2v_conv_wgrad': [_P, _P, _P, _P, _G, _I3, _I, _I, _P],
    't2v_channel_sum_ws_floats': [_I, _I, _L],
    't2v_channel_sum': [_P, _P, _P, _I, _I, _L, _I, _P],
    't2v_channel_sum_grouped_ws_floats': [_P, _I, _I],
    't2v_channel_sum_grouped': [_P, _I, _I, _P, _P, _I, _P],
    't2v_relu': [_P, _P, _L, _P],
    't2v_relu_mask': [_P, _P, _P, _L, _P],
    't2v_add': [_P, _P, _P, _L, _P],
    't2v_axpby': [_F, _P, _F, _P, _P, _L, _P],
    't2v_scale_dev': [_P, _F, _P, _P, _L, _P],
    't2v_dot': [_P, _P, _P, _P, _L, _I, _P],
    't2v_fill': [_P, _F, _L, _P],
    't2v_tanh': [_P, _P, _L, _P],
    't2v_tanh_bwd': [_P, _P, _P, _L, _P],
    't2v_avgpool3d': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I3, _I3, _I3, _P],
    't2v_avgpool3d_bwd': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I3, _I3, _I3, _P],
    't2v_avgpool3d_multi': [_P, _I, _P],
    't2v_avgpool3d_bwd_multi': [_P, _I, _P],
    't2v_maxpool2x2': [_P, _P, _P, _L, _I, _I, _P],
    't2v_maxpool2x2_scatter': [_P, _P, _P, _L, _I, _I, _P],
    't2v_maxpool2x2_gather': [_P, _P, _P, _L, _I, _I, _P],
    't2v_rowsum': [_P, _P, _L, _L, _P],
    't2v_rowbcast': [_P, _P, _L, _L, _P],
    't2v_upsample2x': [_P, _P, _L, _I, _I, _P],
    't2v_upsample2x_bwd': [_P, _P, _L, _I, _I, _P],
    't2v_upsample2x_add': [_P, _P, _P, _L, _I, _I, _P],
    't2v_rsgan_mean_multi': [_P, _I, _P, _P],
    't2v_rsgan_mean_multi_bwd': [_P, _I, _P, _P],
    't2v_bn_ws_floats': [_I, _I, _L],
    't2v_bn_stats': [_P, _P, _P, _P, _P, _I, _I, _L, _F, _F, _P],
    't2v_bn_apply': [_P, _P, _P, _P, _P, _I, _I, _L, _I, _P],
    't2v_bn_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P],
    't2v_bn_train_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _F, _F, _I, _P, _P],
    't2v_bn_train_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P],
    't2v_bn_train_fwd_up': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P],
    't2v_bn_train_bwd_up': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    't2v_bn_train_bwd_add': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't2v_bn_eval': [_P, _P, _P, _P, _P, _P, _I, _I, _L, _F, _I, _P],
    't2v_lstm_gates': [_P, _P, _P, _P, _P, _I, _L, _P],
    't2v_lstm_gates_bwd': [_P, _P, _P, _P, _P, _P, _P, _I, _L, _P],
    't2v_skinny_gemm_splits': [_I, _I, _I],
    't2v_skinny_gemm_slab': [_P, _P, _P, _I, _I, _I, _P],
    't2v_lstm_gates_slab': [_P, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    't2v_lstm_gates_bwd_slab': [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    't2v_concat4': [_P, _P, _P, _P, _P, _I, _P],
    't2v_lstm_step_fused_ok': [_I, _I, _I],
    't2v_lstm_pack_major': [_P, _P, _I, _I, _P],
    't2v_lstm_step_fused': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    't2v_lstm_step_bwd_fused': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    't2v_lstm_pack_cols': [_P, _P, _I, _I, _P],
    't2v_bmm': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    't2v_softmax': [_P, _P, _L, _I, _P],
    't2v_softmax_bwd': [_P, _P, _P, _L, _I, _P],
    't2v_softmax_bwd_bwd_y': [_P, _P, _P, _P, _L, _I, _P],
    't2v_lstm_seq_step': [_P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _P],
    't2v_lstm_seq_step2': [_P, _P, _L, _L, _P, _I, _I, _P],
    't2v_lstm_train_step': [_P, _L, _P, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _P],
    't2v_lstm_train_step_bwd': [_P, _L, _P, _L, _I, _P, _P, _P, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    't2v_embedding_bwd': [_P, _P, _P, _L, _I, _I, _P],
    't2v_xent_fwd': [_P, _P, _P, _P, _L, _I, _P],
    't2v_xent_bwd': [_P, _P, _P, _P, _P, _L, _I, _P],
    't2v_argmax_rows': [_P, _P, _L, _I, _P],
    't2v_multi_ws_floats': [_I, _P, _I],
    't2v_multi': [_I, _P, _I, _P, _P, _P],
    't2v_rsgan': [_P, _P, _P, _I, _P],
    't2v_rsgan_bwd': [_P, _P, _P, _P, _P, _I, _P],
    't2v_gan_loss': [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    't2v_gan_loss_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    't2v_lerp_rows': [_P, _P, _P, _P, _I, _L, _P],
    't2v_row_sqnorm': [_P, _P, _I, _L, _P],
    't2v_row_scale': [_P, _F, _P, _P, _I, _L, _P],
    't2v_adam': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P],
    't2v_adam_multi': [_P, _I, _F, _F, _F, _F, _F, _F, _F, _P, _P],
    't2v_sgd_multi': [_P, _I, _F, _F, _F, _I, _P],
    't2v_adam_tick': [_P, _F, _F, _P],
    't2v_pyramid_gather': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    't2v_copy2d': [_P, _L, _P, _L, _L, _L, _P],
    't2v_cast_bf16': [_P, _P, _L, _I, _P],
    't2v_permute01': [_P, _P, _L, _L, _L, _P],
    't2v_permute12': [_P, _P, _L, _L, _L, _L, _P],
    't2v_subsample_frames': [_P, _P, _L, _L, _L, _L, _L, _I, _I, _P, _P],
    't2v_pyramid_scatter': [_P, _P, _I, _I, _I, _L, _I, _I, _I, _I, _I, _P],
    't2v_scalar_combine': [_P, C.POINTER(C.c_float), _I, _P, _P],
    't2v_gather_rows': [_P, _P, _P, _L, _L, _I, _P],
    't2v_pool_boxsum': [_P, _I, _P],
    't2v_pool_unbox': [_P, _I, _P],
    't2v_pool_conv_fwd_ws_floats': [_P, _I, _I, _I],
    't2v_pool_conv_fwd': [_P, _I, _I, _I, _P, _P, _P, _I, _P],
    't2v_pool_conv_dgrad': [_P, _I, _I, _I, _P, _P],
    't2v_pool_conv_dgrad_splits': [_P, _I, _I, _I],
    't2v_pool_conv_wgrad_slab_floats': [_P, _I, _I, _I, _I, _I],
    't2v_pool_conv_wgrad': [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    't2v_pool_conv_wgrad_partial': [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    't2v_wgrad_swap': [_P, _P, _I, _I, _I, _I, _P],
    't2v_pool_conv_plan': [_I, _P, _I, _I, _I, _I3],
    't2v_synth_clips': [_P, _I, _L, _I, _I, _I, _I3, _P, _P, _P, _P],
    't2v_prof_begin': [_I],
    't2v_prof_end': [C.POINTER(C.c_double), _I],
    't2v_version': [],
}
_RESTYPE = {'t2v_multi_ws_floats': C.c_int64, 't2v_pool_conv_fwd_ws_floats': C.c_int64, 't2v_pool_conv_wgrad_slab_floats': C.c_int64, 't2v_conv_wgrad_slab_floats': C.c_int64, 't2v_conv_wgrad_grouped_bias_slab_floats': C.c_int64, 't2v_conv_fwd_grouped_ws_floats': C.c_int64,
            't2v_conv_wgrad_grouped_slab_floats': C.c_int64, 't2v_conv_fwd_ws_floats': C.c_int64, 't2v_channel_sum_ws_floats': C.c_int64, 't2v_channel_sum_grouped_ws_floats': C.c_int64, 't2v_bn_ws_floats': C.c_int64, 't2v_version': C.c_char_p}

_lib = None


class T2VError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if libt2v_hip.so has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise T2VError('libt2v_hip.so is missing (%s): run `python -c "import __graft_entry__ as g; g.build()"` '
                           'or `make -C txt2vid_amd/csrc`. There is no CPU/PyTorch fallback.' % LIB_PATH)
        # torch first: it bundles its own libamdhip64.so.7 / libhsa-runtime64 (same SONAMEs as /opt/rocm).
        # Whichever is mapped first serves the whole process, and our kernels must launch on the SAME
        # HIP runtime that owns torch's device memory and streams.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        raise T2VError('%s failed with status %d' % (what, status))
