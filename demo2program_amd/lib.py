"""ctypes binding of libd2p_hip.so (C ABI: include/d2p.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``demo2program_amd.build.build_library()`` with
``hipcc --offload-arch=gfx950``.  There is NO fallback: if the library is missing
or a call fails, a RuntimeError is raised -- the product path never routes through
a CPU implementation.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libd2p_hip.so')

P = c_void_p      # device (or host, where noted in d2p.h) pointer
S = c_void_p      # hipStream_t

# name -> (restype, argtypes).  Mirrors include/d2p.h one to one; tests/test_abi.py checks
# that every prototype in the header is listed here and exported by the .so.
SIGNATURES = {
    'd2p_version': (c_int, []),
    'd2p_last_error': (c_char_p, []),
    'd2p_device_info': (c_int, [c_int, P, c_int, P, P, P]),
    'd2p_gemm_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_gemm_set_option': (c_int, [c_int]),
    'd2p_gemm_f32_rows': (c_int, [c_int, c_int, c_int, c_int, P, c_long, P, c_long, P, c_long, P, P, P, c_size_t, S]),
    'd2p_gemm_f32_tn_rows': (c_int, [c_int, c_int, c_int, P, c_long, P, P, c_long, P, P, c_long, c_int, P, c_size_t, S]),
    'd2p_gemm_f32_tn_rows_x2': (c_int, [c_int, c_int, c_int, P, c_long, P, c_long, P, P, c_long, P, c_long, P, c_long, P, P, c_int,
                                P, c_size_t, S]),
    'd2p_gemm_f32_tn_rows2': (c_int, [c_int, c_int, c_int, c_int, P, c_long, P, c_long, P, P, c_long, P, P, c_long, c_int, P,
                              c_size_t, S]),
    'd2p_gemm_f32_batched': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, c_long, c_long, c_long, P, c_long, c_long,
                                     c_long, P, c_long, c_long, c_long, P, c_long, c_long, c_int, c_int, S]),
    'd2p_gemm_force_plan': (c_int, [c_int, c_int]),
    'd2p_gemm_f32_nn': (c_int, [c_int, c_int, c_int, P, c_long, P, c_long, P, c_long, P, c_int, c_int, P, c_size_t, S]),
    'd2p_gemm_f32_nt': (c_int, [c_int, c_int, c_int, P, c_long, P, c_long, P, c_long, P, c_int, c_int, P, c_size_t, S]),
    'd2p_gemm_f32_tn': (c_int, [c_int, c_int, c_int, P, c_long, P, c_long, P, c_long, P, c_int, c_int, P, c_size_t, S]),
    'd2p_colsum_ws_bytes': (c_size_t, [c_int, c_int]),
    'd2p_colsum_f32': (c_int, [c_int, c_int, P, c_long, P, P, c_size_t, S]),
    'd2p_conv_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'd2p_conv_set_direct': (c_int, [c_int, c_int, c_int]),
    'd2p_conv_direct_tune': (c_int, [c_int, c_int, c_int]),
    'd2p_conv2d_nhwc_s2_same_fwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, P, S]),
    'd2p_conv2d_nhwc_s2_same_dgrad': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, S]),
    'd2p_conv2d_nhwc_s2_same_wgrad': (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, c_size_t, S]),
    'd2p_conv_bn_slices': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'd2p_conv_bn_affine_ok': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'd2p_conv2d_nhwc_s2_same_fwd_bn': (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, P, c_int, c_int, P, P,
                                               P, c_int, S]),
    'd2p_conv2d_nhwc_s2_same_wgrad_bn': (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, c_int, P, P, P,
                                                 c_size_t, S]),
    'd2p_bn_stats_from_partials': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, S]),
    'd2p_bn_apply_fwd': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, S]),
    'd2p_bn_group_bwd_coef': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_int, P, c_size_t, S]),
    'd2p_conv_dgrad_bn_slices': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'd2p_conv2d_nhwc_s2_same_dgrad_bn': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, c_int, P, c_int, S]),
    'd2p_conv_bnbwd_ok': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'd2p_conv2d_nhwc_s2_same_wgrad_bnbwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, c_int, c_int, P, P,
                                                    P, c_size_t, S]),
    'd2p_karel_encoder_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_karel_encoder_set_trace': (c_int, [P]),
    'd2p_karel_encoder_fwd': (c_int, [c_int, c_int, c_int, P, c_int, P, P, P, P, P, P, P, P, P, P, P, c_size_t, S]),
    'd2p_karel_encoder_bwd_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_rn_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_rn_fc1_fwd': (c_int, [c_int, c_int, c_int, P, P, P, P, P, c_long, P, P, P, P, P, P, P, c_float, P, c_size_t, S]),
    'd2p_rn_fc2_fwd': (c_int, [c_int, c_int, c_int, P, P, P, c_long, P, P, P, P, P, P, P, P, c_float, P, c_size_t, S]),
    'd2p_rn_fc2_bwd': (c_int, [c_int, c_int, c_int, P, P, P, P, c_long, P, P, P, P, P, P, c_size_t, S]),
    'd2p_rn_fc1_bwd': (c_int, [c_int, c_int, c_int, P, P, P, c_long, P, P, P, P, P, P, P, P, P, c_size_t, S]),
    'd2p_karel_encoder_bwd': (c_int, [c_int, c_int, c_int, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, S]),
    'd2p_bn_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_bn_batched_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'd2p_per_affine_rows': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, S]),
    'd2p_per_rows_nn': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, S]),
    'd2p_per_rows_tn_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'd2p_per_rows_tn': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, S]),
    'd2p_per_fc_bn_stats': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, S]),
    'd2p_per_fc_bn_bwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, S]),
    'd2p_bn_group_fwd_batched': (c_int, [c_int, c_long, c_long, c_long, c_long, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                                         P, P, P, c_float, P, c_size_t, S]),
    'd2p_bn_group_bwd_batched': (c_int, [c_int, c_long, c_long, c_long, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int,
                                         P, P, P, P, P, c_size_t, S]),
    'd2p_bn_group_fwd': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_float, P, c_size_t, S]),
    'd2p_bn_group_bwd': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, P, P, c_size_t, S]),
    'd2p_bn_group_bwd_sums': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, P, P, c_int, P, c_size_t, S]),
    'd2p_bn_inference_fwd': (c_int, [c_int, c_int, P, P, P, P, P, P, S]),
    'd2p_bn_set_fold': (c_int, [c_int]),
    'd2p_bn_update_moving': (c_int, [c_int, c_int, c_float, P, P, P, P, S]),
    'd2p_lstm_gate_fwd': (c_int, [c_int, c_int, P, c_long, P, P, P, c_int, P, P, P, S]),
    'd2p_lstm_gate_bwd': (c_int, [c_int, c_int, P, c_long, P, P, P, P, P, c_int, P, P, c_long, P, S]),
    'd2p_lstm_ws_bytes': (c_size_t, [c_int, c_int]),
    'd2p_lstm_set_fused': (c_int, [c_int]),
    'd2p_lstm_set_persistent': (c_int, [c_int]),
    'd2p_lstm_persist_error': (c_int, [c_int]),
    'd2p_lstm_persist_inject_error': (c_int, []),
    'd2p_lstm_persist_set_bwd_defer': (c_int, [c_int]),
    'd2p_lstm_persist_set_bwd_desc': (c_int, [c_int]),
    'd2p_lstm_flag_words': (c_size_t, []),
    'd2p_lstm_persist_set_sorted': (c_int, [c_int]),
    'd2p_lstm_persist_set_poll': (c_int, [c_int]),
    'd2p_lstm_persist_set_plan_cost': (c_int, [ctypes.c_double, ctypes.c_double]),
    'd2p_lstm_pack_weights': (c_int, [c_int, c_int, P, P, P, S]),
    'd2p_lstm_persist_set_direct': (c_int, [c_int]),
    'd2p_lstm_persist_set_trace': (c_int, [P, c_size_t, c_int]),
    'd2p_lstm_persist_set_wgs_per_cu': (c_int, [c_int, c_int]),
    'd2p_lstm_persist_set_cu_budget': (c_int, [c_int]),
    'd2p_lstm_persist_pair_launches': (c_int, []),
    'd2p_lstm_persist_set_fwd_wide': (c_int, [c_int, c_int, c_int, c_int]),
    'd2p_lstm_persist_set_fwd_plan_cost': (c_int, [ctypes.c_double, ctypes.c_double]),
    'd2p_lstm_persist_wide_launches': (c_int, [c_int]),
    'd2p_lstm_persist_wide_local_wgs': (c_int, [c_int]),
    'd2p_lstm_debug_flags': (c_int, [c_int]),
    'd2p_lstm_set_tiling': (c_int, [c_int, c_int, c_int]),
    'd2p_lstm_seq_fwd': (c_int, [c_int, c_int, c_int, P, c_long, c_long, P, P, P, P, P, P, P, P, P, c_size_t, S]),
    'd2p_lstm_seq_bwd': (c_int, [c_int, c_int, c_int, P, c_long, c_long, P, P, P, P, P, P, P, P, P, P, P, c_size_t, S]),
    'd2p_lstm_seq_fwd_multi': (c_int, [c_int, P, S]),
    'd2p_lstm_seq_bwd_multi': (c_int, [c_int, P, S]),
    'd2p_shift_tokens_tm': (c_int, [c_int, c_int, P, c_int, P, S]),
    'd2p_embedding_gather_oob0': (c_int, [c_int, c_int, c_int, P, P, P, S]),
    'd2p_embedding_scatter_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_embedding_scatter_add_oob0': (c_int, [c_int, c_int, c_int, P, P, P, P, c_size_t, S]),
    'd2p_greedy_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'd2p_greedy_decode': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, c_int, P, P, P, P, c_size_t, S]),
    'd2p_sched_sample': (c_int, [c_int, c_int, P, P, P, P, c_int, P, P, S]),
    'd2p_argmax_rows': (c_int, [c_int, c_int, P, c_long, P, S]),
    'd2p_xent_ws_bytes': (c_size_t, [c_int]),
    'd2p_softmax_xent_masked_fwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, c_long, c_long, c_long, P, P, P, P, c_size_t, S]),
    'd2p_softmax_xent_masked_bwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, c_long, c_long, c_long, P, P, c_float, P, S]),
    'd2p_sigmoid_xent_masked_fwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, c_long, c_long, c_long, P, P, P, P, c_size_t, S]),
    'd2p_sigmoid_xent_masked_bwd': (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, c_long, c_long, c_long, P, P, c_float, P, S]),
    'd2p_xent_bwd_dhout_multi': (c_int, [c_int, P, S]),
    'd2p_small_pair_products': (c_int, [c_int, P, S]),
    'd2p_loss_assemble': (c_int, [c_int, P, P, P, P, P, S]),
    'd2p_loss_from_partials': (c_int, [c_int, P, P, P, P, P, P, P, S]),
    'd2p_group_mean': (c_int, [c_int, c_int, c_int, P, P, P, S]),
    'd2p_group_mean_bwd': (c_int, [c_int, c_int, c_int, P, P, P, c_int, S]),
    'd2p_group_max': (c_int, [c_int, c_int, c_int, P, P, P, S]),
    'd2p_group_max_bwd': (c_int, [c_int, c_int, c_int, P, P, P, c_int, S]),
    'd2p_rn_pair_fwd': (c_int, [c_int, c_int, c_int, P, P, P, c_int, c_long, P, S]),
    'd2p_rn_pair_bwd': (c_int, [c_int, c_int, c_int, P, P, P, S]),
    'd2p_pair_mean_fwd': (c_int, [c_int, c_int, c_int, P, P, P, S]),
    'd2p_pair_mean_bwd': (c_int, [c_int, c_int, c_int, P, P, S]),
    'd2p_axpy': (c_int, [c_size_t, c_float, P, P, c_int, S]),
    'd2p_transpose_rt': (c_int, [c_int, c_int, c_int, P, P, S]),
    'd2p_pad_axis': (c_int, [c_long, c_int, c_int, c_int, P, P, c_int, c_int, S]),
    'd2p_probe_clock': (c_int, [c_int, c_int, c_int, c_int, P, P, P, S]),
    'd2p_zero_past_group_steps': (c_int, [c_int, c_int, c_int, c_int, P, P, S]),
    'd2p_l2norm_ws_bytes': (c_size_t, [c_size_t]),
    'd2p_l2norm_flat': (c_int, [c_size_t, P, c_float, P, P, c_size_t, S]),
    'd2p_prof_enable': (c_int, [c_int]),
    'd2p_prof_set_tag': (c_int, [c_int]),
    'd2p_prof_read': (c_int, [c_int, P, P, P]),
    'd2p_adam_clip_flat': (c_int, [c_size_t, P, P, P, P, P, c_float, c_float, c_float, P, c_float, c_float, c_float, S]),
    'd2p_step_status_publish': (c_int, [P, S]),
    'd2p_adam_clip_flat_guarded': (c_int, [c_size_t, P, P, P, P, P, c_float, c_float, c_float, P, c_float, c_float, c_float,
                                           P, P, P, S]),
}



class LstmFwdDesc(ctypes.Structure):
    """d2p_lstm_fwd_desc (include/d2p.h)."""
    _fields_ = [('M', c_int), ('U', c_int), ('n_steps', c_int),
                ('z', c_void_p), ('z_row_stride', c_long), ('z_t_stride', c_long),
                ('Wh', c_void_p), ('h0', c_void_p), ('c0', c_void_p), ('lens', c_void_p),
                ('hout', c_void_p), ('cs', c_void_p), ('h_final', c_void_p), ('c_final', c_void_p),
                ('ws', c_void_p), ('ws_bytes', c_size_t), ('flags', c_void_p), ('epoch', ctypes.c_uint),
                ('wpack', c_void_p), ('rowmap', c_void_p), ('slab_steps', c_void_p)]


class LstmBwdDesc(ctypes.Structure):
    """d2p_lstm_bwd_desc (include/d2p.h)."""
    _fields_ = [('M', c_int), ('U', c_int), ('n_steps', c_int),
                ('z', c_void_p), ('z_row_stride', c_long), ('z_t_stride', c_long),
                ('Wh', c_void_p), ('c0', c_void_p), ('lens', c_void_p), ('cs', c_void_p),
                ('dhout', c_void_p), ('dh_final', c_void_p), ('dc_final', c_void_p),
                ('dz', c_void_p), ('dh0', c_void_p), ('dc0', c_void_p),
                ('ws', c_void_p), ('ws_bytes', c_size_t), ('db', c_void_p), ('flags', c_void_p),
                ('epoch', ctypes.c_uint), ('wpack', c_void_p), ('rowmap', c_void_p), ('slab_steps', c_void_p)]


class PairProductsDesc(ctypes.Structure):
    """d2p_pair_products_desc (include/d2p.h)."""
    _fields_ = [('R', c_int), ('U', c_int), ('N4', c_int), ('S', c_void_p), ('A', c_void_p), ('Wx', c_void_p),
                ('G1', c_void_p), ('G2', c_void_p)]


class XentBwdDesc(ctypes.Structure):
    """d2p_xent_bwd_desc (include/d2p.h)."""
    _fields_ = [('sigmoid', c_int), ('R', c_int), ('V', c_int), ('G', c_int), ('n_steps', c_int), ('U', c_int),
                ('logits', c_void_p), ('labels', c_void_p), ('label_rs', c_long), ('label_ts', c_long),
                ('label_vs', c_long), ('lens', c_void_p), ('den', c_void_p), ('scale', c_float),
                ('dlogits', c_void_p), ('proj', c_void_p), ('dhout', c_void_p), ('hout', c_void_p),
                ('logits_out', c_void_p), ('loss_part', c_void_p)]


_lib = None


class D2PError(RuntimeError):
    pass


def load():
    """dlopen the library and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime (libamdhip64 of the ROCm it was built with); loading this
    # library before it would pull in the system runtime instead and leave the process with two --
    # kernels then fail with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    path = os.environ.get('D2P_LIB_PATH', LIB_PATH)       # (an alternative build of the same sources, for A/B runs)
    if not os.path.exists(path):
        raise D2PError(
            'libd2p_hip.so not found at %s -- run `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        if 'D2P_LIB_PATH' in os.environ and not hasattr(lib, name):
            continue                     # (an older build of the sources, loaded for an A/B run: newer entry points absent)
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(name, rc):
    if rc != 0:
        msg = load().d2p_last_error()
        raise D2PError('%s failed (code %d): %s' % (name, rc, msg.decode() if msg else ''))


class _Caller(object):
    """``call.d2p_xxx(...)`` -> invokes the C function, raises D2PError on a non-zero
    return code for int-returning entry points."""

    def __getattr__(self, name):
        lib = load()
        fn = getattr(lib, name)
        res = SIGNATURES[name][0]
        if res is c_int and name != 'd2p_version':
            def wrapped(*args):
                _check(name, fn(*args))
        else:
            wrapped = fn
        setattr(self, name, wrapped)
        return wrapped


call = _Caller()


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
