"""ctypes binding of libapamd.so (C ABI declared in include/animateportrait_amd.h).

The product path has NO fallback: if the shared library is missing or a call fails,
a RuntimeError is raised.  PyTorch is used only for device memory and streams; the
signatures below carry raw device pointers and sizes.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# APAMD_LIB: alternative build of the same library (A/B kernel experiments); default = the in-tree build
LIB_PATH = os.environ.get('APAMD_LIB') or os.path.join(_HERE, 'libapamd.so')

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
W_OIHW, W_IOHW = 0, 1

c_f32p = ctypes.c_void_p


class ApSrc(ctypes.Structure):
    _fields_ = [('data', c_f32p), ('mean', c_f32p), ('rstd', c_f32p),
                ('C', ctypes.c_int32), ('act', ctypes.c_int32)]


class ApConvDesc(ctypes.Structure):
    _fields_ = [('N', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
                ('Cout', ctypes.c_int32), ('KH', ctypes.c_int32), ('KW', ctypes.c_int32),
                ('stride', ctypes.c_int32), ('pad', ctypes.c_int32), ('pad_mode', ctypes.c_int32),
                ('transposed', ctypes.c_int32), ('output_padding', ctypes.c_int32),
                ('w_layout', ctypes.c_int32), ('w_flip', ctypes.c_int32), ('act', ctypes.c_int32),
                ('nsrc', ctypes.c_int32), ('precision', ctypes.c_int32),
                ('presplit', ctypes.c_int32), ('s2d_k', ctypes.c_int32),
                ('src', ApSrc * 3)]


class ApFusedNorm(ctypes.Structure):
    _fields_ = [('act', ctypes.c_int32), ('eps', ctypes.c_float), ('res_oct', ctypes.c_void_p), ('res_nchw', ctypes.c_void_p),
                ('y_oct', ctypes.c_void_p), ('xs', ctypes.c_void_p), ('mean', ctypes.c_void_p), ('rstd', ctypes.c_void_p),
                ('partials', ctypes.c_void_p), ('counters', ctypes.c_void_p)]


class ApWeightView(ctypes.Structure):
    _fields_ = [('w', c_f32p), ('s_co', ctypes.c_int64), ('s_ci', ctypes.c_int64), ('s_ky', ctypes.c_int64),
                ('s_kx', ctypes.c_int64), ('s2d_c', ctypes.c_int32), ('rows_c', ctypes.c_int32), ('ksrc', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


class ApOutView(ctypes.Structure):
    _fields_ = [('nstride', ctypes.c_int64), ('cstride', ctypes.c_int64), ('rstride', ctypes.c_int32),
                ('xstride', ctypes.c_int32), ('y_off', ctypes.c_int32), ('x_off', ctypes.c_int32),
                ('OH', ctypes.c_int32), ('OW', ctypes.c_int32)]


class ApWgradDesc(ctypes.Structure):
    _fields_ = [('N', ctypes.c_int32), ('M', ctypes.c_int32), ('GH', ctypes.c_int32), ('GW', ctypes.c_int32),
                ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('K', ctypes.c_int32), ('stride', ctypes.c_int32),
                ('pad', ctypes.c_int32), ('pad_mode', ctypes.c_int32), ('nsrc', ctypes.c_int32),
                ('precision', ctypes.c_int32), ('g', ApSrc), ('src', ApSrc * 3),
                ('src_xs', ctypes.c_void_p * 3), ('src_xs_s2d', ctypes.c_void_p), ('xs_parts', ctypes.c_int32)]


# name -> (restype, argtypes); every symbol include/animateportrait_amd.h declares
ABI_VERSION = 13     # AP_ABI_VERSION of include/animateportrait_amd.h this binding was written against

SIGNATURES = {
    'ap_abi_version': (ctypes.c_int32, []),
    'ap_version': (ctypes.c_char_p, []),
    'ap_last_error': (ctypes.c_char_p, []),
    'ap_conv2d_out_size': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(ctypes.c_int32)]),
    'ap_conv2d_packed_floats': (ctypes.c_int64, [ctypes.POINTER(ApConvDesc)]),
    'ap_conv2d_stat_tiles': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc)]),
    'ap_conv2d_wants_presplit': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc)]),
    'ap_split_prepass_bytes': (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'ap_split_prepass': (ctypes.c_int, [ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_void_p]),
    'ap_split_prepass_rows': (ctypes.c_int, [ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    'ap_split_prepass_s2d': (ctypes.c_int, [ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    'ap_norm_apply_split': (ctypes.c_int, [ctypes.POINTER(ApSrc), c_f32p, ctypes.c_int32, ctypes.c_float, c_f32p, c_f32p,
                                           ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           c_f32p, ctypes.c_void_p, ctypes.c_void_p]),
    'ap_norm_apply_split_ex': (ctypes.c_int, [ctypes.POINTER(ApSrc), c_f32p, ctypes.c_int32, ctypes.c_float, c_f32p, c_f32p,
                                              ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                              c_f32p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    'ap_conv2d_kernel_name': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), ctypes.c_char_p, ctypes.c_int32]),
    'ap_conv2d_pack_weights': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_pack_entry_bytes': (ctypes.c_int32, []),
    'ap_conv2d_pack_entries': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc), ctypes.POINTER(ApWeightView), c_f32p, ctypes.c_void_p,
                                              ctypes.c_int32]),
    'ap_conv2d_pack_run': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    'ap_conv2d_fwd': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_fwd_view': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), ctypes.POINTER(ApOutView), c_f32p, c_f32p, c_f32p,
                                          ctypes.c_void_p]),
    'ap_instnorm_finalize': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                            c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_instnorm_finalize_octet': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_float, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_octet_ok': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc)]),
    'ap_conv2d_fused_norm_ok': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc)]),
    'ap_conv2d_fused_norm_counters': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc)]),
    'ap_conv2d_fwd_norm': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), c_f32p, ctypes.POINTER(ApFusedNorm), ctypes.c_void_p]),
    'ap_conv2d_fwd_octet': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_bf16out_ok': (ctypes.c_int32, [ctypes.POINTER(ApConvDesc)]),
    'ap_conv2d_fwd_bf16out': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), c_f32p, c_f32p, ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_fwd_view_bf16out': (ctypes.c_int, [ctypes.POINTER(ApConvDesc), ctypes.POINTER(ApOutView), c_f32p, c_f32p, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
    'ap_instnorm_apply': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    'ap_warp_concat_fwd': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_float, ctypes.c_void_p]),
    'ap_warp_concat_fwd_split': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                                ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p]),
    'ap_warp_concat_fwd_ex': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_int32, ctypes.c_void_p]),
    'ap_motion_grid': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_resize_bilinear': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_grid_sample': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_conv_head_wgrad': (ctypes.c_int, [ctypes.POINTER(ApSrc), c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_conv_final_wgrad_workspace_floats': (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'ap_conv_final_wgrad': (ctypes.c_int, [ctypes.POINTER(ApSrc), c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_wgrad_k7_bf16_ok': (ctypes.c_int32, [ctypes.c_int32] * 6),
    'ap_wgrad_k7_bf16_workspace_floats': (ctypes.c_int64, [ctypes.c_int32] * 6),
    'ap_wgrad_k7_bf16': (ctypes.c_int, [ctypes.POINTER(ApSrc), ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv_final_dgrad_bf16_ok': (ctypes.c_int32, [ctypes.c_int32] * 4),
    'ap_conv_final_dgrad_bf16_workspace_floats': (ctypes.c_int64, [ctypes.c_int32] * 4),
    'ap_conv_final_dgrad_bf16': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv_d0_fwd_bf16_ok': (ctypes.c_int32, [ctypes.c_int32] * 5),
    'ap_conv_d0_fwd_bf16': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_wgrad_d0_bf16_ok': (ctypes.c_int32, [ctypes.c_int32] * 5),
    'ap_wgrad_d0_bf16_workspace_floats': (ctypes.c_int64, [ctypes.c_int32] * 5),
    'ap_wgrad_d0_bf16': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv_head_dgrad_bf16_ok': (ctypes.c_int32, [ctypes.c_int32] * 4),
    'ap_conv_head_dgrad_bf16': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p,
                                               ctypes.c_void_p]),
    'ap_act_bwd_bias_workspace_floats': (ctypes.c_int64, [ctypes.c_int32] * 4),
    'ap_act_bwd_bias': (ctypes.c_int, [c_f32p, ctypes.c_int32, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_wgrad_workspace_floats': (ctypes.c_int64, [ctypes.POINTER(ApWgradDesc)]),
    'ap_pad_materialize': (ctypes.c_int, [ctypes.POINTER(ApSrc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_wgrad': (ctypes.c_int, [ctypes.POINTER(ApWgradDesc), c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_instnorm_bwd': (ctypes.c_int, [c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p, c_f32p,
                                       ctypes.c_void_p]),
    'ap_instnorm_bwd_split_ok': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'ap_instnorm_bwd_split': (ctypes.c_int, [c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), c_f32p, c_f32p, ctypes.c_int32,
                                             ctypes.c_void_p]),
    'ap_conv2d_wgrad_gt_dims': (ctypes.c_int, [ctypes.POINTER(ApWgradDesc), ctypes.POINTER(ctypes.c_int32)]),
    'ap_conv2d_wgrad_pre': (ctypes.c_int, [ctypes.POINTER(ApWgradDesc), ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_conv2d_wgrad_xs_ok': (ctypes.c_int32, [ctypes.POINTER(ApWgradDesc)]),
    'ap_conv2d_wgrad_xs': (ctypes.c_int, [ctypes.POINTER(ApWgradDesc), ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_act_bwd': (ctypes.c_int, [c_f32p, ctypes.c_int32, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32,
                                  ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_bias_grad': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_bias_grad_workspace_floats': (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'ap_bias_grad_ws': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p, c_f32p,
                                       ctypes.c_void_p]),
    'ap_warp_concat_bwd': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                          ctypes.c_void_p]),
    'ap_tps_solve': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p,
                                    ctypes.c_void_p]),
    'ap_tps_warp': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_int32, ctypes.c_int32, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_adam_step': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_int32, ctypes.c_void_p]),
    'ap_reduce_workspace_floats': (ctypes.c_int64, []),
    'ap_reduce_mean': (ctypes.c_int, [ctypes.c_int32, c_f32p, c_f32p, ctypes.c_float, ctypes.c_int64, ctypes.c_float,
                                      c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_reduce_mean_bwd': (ctypes.c_int, [ctypes.c_int32, c_f32p, c_f32p, ctypes.c_float, ctypes.c_int64, ctypes.c_float,
                                          c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_mask_composite': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_mask_composite_bwd': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_void_p]),
    'ap_axpy': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]),
    'ap_crop_resize_fwd': (ctypes.c_int, [c_f32p, ctypes.c_void_p] + [ctypes.c_int32] * 9 +
                           [ctypes.c_float, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    'ap_crop_resize_bwd': (ctypes.c_int, [c_f32p, ctypes.c_void_p] + [ctypes.c_int32] * 9 +
                           [ctypes.c_float, c_f32p, ctypes.c_void_p]),
    'ap_kp_to_map': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    'ap_flow_post': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, ctypes.c_void_p]),
    'ap_landmark_discs': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_float, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    'ap_circle_rows': (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    'ap_lstm_workspace_bytes': (ctypes.c_int64, [ctypes.c_int32]),
    'ap_lstm_recurrence': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    'ap_lstm_timed_out': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    'ap_pixel_shuffle2': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p,
                                         ctypes.c_void_p]),
    'ap_lip_line_mask': (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f32p,
                                        ctypes.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load libapamd.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        'animateportrait_amd: %s not found. Build it with '
                        '`python -c "import __graft_entry__ as g; g.build()"` or '
                        '`make -C animateportrait_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
                l = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)
                    fn.restype = res
                    fn.argtypes = args
                if l.ap_abi_version() != ABI_VERSION:
                    raise RuntimeError('animateportrait_amd: %s speaks ABI %d, this binding ABI %d: rebuild the library '
                                       '(make -C animateportrait_amd/csrc)' % (LIB_PATH, l.ap_abi_version(), ABI_VERSION))
                _lib = l
    return _lib


def check(rc, what=''):
    if rc < 0:
        msg = lib().ap_last_error()
        raise RuntimeError('libapamd %s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))
    return rc
