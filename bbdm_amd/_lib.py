"""ctypes binding of libbbdm_hip.so (the C-ABI declared in include/bbdm_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the product path raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C bbdm_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# BBDM_HIP_LIB overrides the library path (A/B runs of kernel variants); the default is the in-tree build
LIB_PATH = os.environ.get("BBDM_HIP_LIB") or os.path.join(_HERE, "libbbdm_hip.so")
ABI_VERSION = 25

_P = c_void_p
# name -> (restype, argtypes); must list every symbol of include/bbdm_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "bbdm_version": (c_int, []),
    "bbdm_last_error": (c_char_p, []),
    "bbdm_device_cus": (c_int, []),
    "bbdm_nchw_to_nhwc_f32": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_nhwc_to_nchw_f32": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_conv_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_conv_pack_weight_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_conv_splitk_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbdm_conv2d_nhwc_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, c_int, _P, c_size_t, _P, _P, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_conv_stats_fusable": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbdm_conv2d_nhwc_stats_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, c_int, _P, c_size_t, _P, _P, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P]),
    "bbdm_winograd_output_stats_f32": (c_int, [c_int, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                               _P, c_int, c_int, _P, c_int, c_int, _P]),
    "bbdm_groupnorm_coeffs_f32": (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "bbdm_winograd_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_winograd_pack_weight_f32": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_pack_weight_bf3p_f32": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbdm_conv3x3_winograd_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, _P, c_int, c_int, _P,
                                           c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_tiles": (c_size_t, [c_int, c_int, c_int, c_int]),
    "bbdm_upsample_phase_weights_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "bbdm_winograd_input_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_gemm_f32": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_output_f32": (c_int, [c_int, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_conv_packed_dgrad_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "bbdm_conv_pack_weight_dgrad_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_conv_wgrad_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbdm_conv_wgrad_f32": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_wgrad_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbdm_conv3x3_winograd_wgrad_f32": (c_int, [c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_dy_transform_f32": (c_int, [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_gemm_tn_splits": (c_int, [c_int, c_int64, c_int, c_int]),
    "bbdm_gemm_tn_batched_f32": (c_int, [_P, c_int, c_size_t, _P, c_int, c_size_t, _P, c_int, c_int64, c_int, c_int, _P]),
    "bbdm_winograd_wgrad_finish_f32": (c_int, [c_int, _P, c_int, _P, c_int, c_int, _P]),
    "bbdm_winograd_wgrad_finish_bias_f32": (c_int, [c_int, _P, c_int, _P, c_int, c_int, _P, ctypes.c_longlong, _P, _P]),
    "bbdm_colsum_f32": (c_int, [_P, c_int, _P, _P, ctypes.c_longlong, c_int, _P]),
    "bbdm_colsum_batched_f32": (c_int, [_P, c_int, _P, _P, c_int, c_int, ctypes.c_longlong, c_int, _P]),
    "bbdm_groupnorm_stats_bytes": (c_size_t, [c_int, c_int]),
    "bbdm_groupnorm_stats_read_f64": (c_int, [_P, _P, c_int, c_int, _P]),
    "bbdm_groupnorm_stats_f32": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_groupnorm_apply_f32": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_float, c_int, c_int, _P]),
    "bbdm_attention_f32": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_attention_bwd_f32": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                       c_int, _P]),
    "bbdm_linear_bwd_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_linear_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_bb_loss_bwd_f32": (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
    "bbdm_groupnorm_bwd_workspace_doubles": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_groupnorm_bwd_f32": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, _P, _P,
                                       _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    "bbdm_timestep_embedding_f32": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "bbdm_linear_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_linear_packed_bytes": (c_size_t, [c_int, c_int]),
    "bbdm_linear_packed_supported": (c_int, [c_int, c_int, c_int]),
    "bbdm_linear_pack_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "bbdm_linear_packed_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_bb_q_sample_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "bbdm_bb_p_sample_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_int,
                                          _P, _P, _P, c_int, c_int, _P]),
    "bbdm_bb_predict_x0_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "bbdm_bb_loss_f32": (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
    "bbdm_gemm_packed_b_floats": (c_size_t, [c_int, c_int]),
    "bbdm_gemm_pack_b_f32": (c_int, [_P, c_int, c_size_t, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_gemm_batched_f32": (c_int, [_P, c_int, c_size_t, _P, _P, c_int, c_size_t, c_int, c_int, c_int, c_int, _P]),
    "bbdm_softmax_rows_f32": (c_int, [_P, c_int, ctypes.c_longlong, c_int, c_float, _P]),
    "bbdm_vq_nearest_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_attention_kv_planes_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "bbdm_attention_kv_planes_f32": (c_int, [_P, c_int, _P, ctypes.c_size_t, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_attention_planes_f32": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "bbdm_attention_kv_planes_h2_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "bbdm_attention_kv_planes_h2_f32": (c_int, [_P, c_int, _P, ctypes.c_size_t, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "bbdm_attention_planes_h2_f32": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "bbdm_h2_rowl1_f32": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "bbdm_h2_affine_bound_f32": (c_int, [_P, _P, _P, _P]),
    "bbdm_cross_attention_f32": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_cross_attention_bwd_f32": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P, c_int,
                                             c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_layernorm_bwd_f32": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "bbdm_geglu_bwd_f32": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int64, c_int, _P]),
    "bbdm_layernorm_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, ctypes.c_longlong, c_int, c_float, _P]),
    "bbdm_geglu_f32": (c_int, [_P, c_int, _P, c_int, ctypes.c_longlong, c_int, _P]),
    "bbdm_gemm_bf3_packed_halfs": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_gemm_bf3_pack_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "bbdm_gemm_bf3_supported": (c_int, [ctypes.c_longlong, c_int, c_int]),
    "bbdm_gemm_bf3_f32": (c_int, [_P, _P, _P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_conv1x1_bf3_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_winograd_gemm_bf3_f32": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_gemm_bf3p_a_bytes": (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    "bbdm_gemm_bf3p_b_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_gemm_bf3p_supported": (c_int, [ctypes.c_longlong, c_int, c_int]),
    "bbdm_gemm_bf3p_pack_b_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "bbdm_gemm_bf3p_split_rows_f32": (c_int, [_P, c_int, _P, c_int, ctypes.c_longlong, c_int, _P]),
    "bbdm_gemm_bf3p_f32": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    # the fp16-pair planes (ABI 24; csrc/h2_split.h)
    "bbdm_gemm_h2p_a_bytes": (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    "bbdm_gemm_h2p_b_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bbdm_absmax_f32": (c_int, [_P, ctypes.c_longlong, _P, _P]),
    "bbdm_absmax_rows_f32": (c_int, [_P, c_int, ctypes.c_longlong, c_int, _P, _P]),
    "bbdm_winograd_input_h2p_tr_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "bbdm_winograd_input_h2p_tr2_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "bbdm_winograd_dy_transform_h2p_f32": (c_int, [c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "bbdm_winograd_dy_gain": (c_float, [c_int]),
    "bbdm_gemm_h2p_tn_at_bytes": (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    "bbdm_gemm_h2p_tn_bt_bytes": (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    "bbdm_gemm_h2p_tn_f32": (c_int, [_P, _P, _P, c_int, ctypes.c_longlong, c_int, c_int, _P, c_float, _P, c_float, _P]),
    "bbdm_gemm_h2p_pack_b_f32": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, _P]),
    "bbdm_winograd_pack_weight_h2p_f32": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "bbdm_winograd_g_gain": (c_float, [c_int]),
    "bbdm_gemm_h2p_split_rows_f32": (c_int, [_P, c_int, _P, _P, c_int, ctypes.c_longlong, c_int, _P]),
    "bbdm_gemm_h2p_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_gemm_h2p_splitk_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong, c_int, c_int, c_int, _P]),
    "bbdm_h2_gn_bounds_f32": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    "bbdm_h2_stats_bound_f32": (c_int, [_P, c_int, c_int, _P, _P]),
    "bbdm_conv1x1_h2q_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, ctypes.c_longlong, c_int, c_int, _P, _P, _P]),
    "bbdm_conv1x1_h2s_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, ctypes.c_longlong, c_int, c_int, _P, _P, _P]),
    "bbdm_winograd_input_gain": (c_float, [c_int]),
    "bbdm_winograd_input_h2p_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "bbdm_winograd_input_h2p_gn_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               _P, _P, _P, c_int, c_int, c_int, ctypes.c_float, _P, _P]),
    "bbdm_winograd_gemm_h2p_f32": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "bbdm_winograd_gemm_h2p_splitk_f32": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "bbdm_conv1x1_bf3q_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_conv1x1_bf3s_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_winograd_input_bf3p_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_input_bf3p_gn_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                _P, _P, _P, c_int, c_int, c_int, ctypes.c_float, _P]),
    "bbdm_winograd_gemm_bf3p_f32": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_gemm_bf3p_fwd_splits": (c_int, [c_int, ctypes.c_longlong, c_int, c_int]),
    "bbdm_gemm_bf3p_splitk_f32": (c_int, [_P, _P, _P, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong, c_int, c_int, c_int, _P]),
    "bbdm_winograd_gemm_bf3p_splits": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbdm_winograd_gemm_bf3p_splitk_f32": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_winograd_output_splitk_stats_f32": (c_int, [c_int, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                                      _P, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "bbdm_winograd_input_bf3p_tr_f32": (c_int, [c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "bbdm_winograd_dy_transform_bf3p_f32": (c_int, [c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bbdm_gemm_bf3p_tn_at_bytes": (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    "bbdm_gemm_bf3p_tn_bt_bytes": (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    "bbdm_gemm_bf3p_tn_supported": (c_int, [ctypes.c_longlong, c_int, c_int]),
    "bbdm_gemm_bf3p_tn_splits": (c_int, [c_int, ctypes.c_longlong, c_int, c_int]),
    "bbdm_gemm_bf3p_tn_f32": (c_int, [_P, _P, _P, c_int, ctypes.c_longlong, c_int, c_int, _P]),
    "bbdm_images_to_u8_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bbdm_opt_chunk_elems": (c_int, []),
    "bbdm_adam_ema_step_f32": (c_int, [_P, c_int, c_int, c_double, c_double, c_double, c_double, c_double,
                                       ctypes.c_longlong, c_int, c_double, _P]),
    "bbdm_set_option": (c_int, [c_char_p, c_int]),              # header: "options" (tests / tools A-B runs)
    "bbdm_get_option": (c_int, [c_char_p, _P]),
}

_lib = None


class BBDMHipError(RuntimeError):
    pass


def bind(path: str):
    """dlopen ``path`` and bind every declared symbol (raises if one is missing or the ABI version differs)."""
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.bbdm_version()
    if v != ABI_VERSION:
        raise BBDMHipError(f"{path}: ABI version {v} != expected {ABI_VERSION}; rebuild it")
    return lib


def load():
    """Load the shared library (once) and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BBDMHipError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `make -C bbdm_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "bbdm_amd has no CPU/PyTorch fallback by design.")
    _lib = bind(LIB_PATH)
    return _lib


# ---- the places the package touches the HIP runtime through torch (device memory / streams are torch's) ---------
def require_gpu(*tensors):
    """No CPU fallback by design: every tensor handed to the kernels must live on a GPU."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise BBDMHipError("bbdm_amd runs on the GPU only (no CPU fallback by design); "
                               f"got a tensor on {t.device}")


def current_stream(device) -> int:
    """Raw ``hipStream_t`` of torch's current stream on ``device`` (the kernels are enqueued there)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def device_guard(device):
    """Context manager making ``device`` the current HIP device for the calls inside it.

    torch's default stream handle is 0 -- the NULL stream, which HIP binds to the *current* device, not to the device
    of the pointers passed: without this guard a model living on cuda:1 in a process that never called ``set_device``
    (the reference's ``main.py --gpu_ids 1``) would launch on device 0 with device-1 pointers."""
    import torch
    return torch.cuda.device(device)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().bbdm_last_error()
        raise BBDMHipError(f"{what or 'bbdm call'} failed (rc={rc}): {msg.decode() if msg else ''}")


class option:
    """``with _lib.option("bf3p_kernel", 4): ...`` -- set one of the library's integer options (include/bbdm_hip.h "options") for the
    block and restore the previous value.  Tests and tools/ A-B runs; the product path never changes an option."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        old = ctypes.c_int(0)
        call("bbdm_get_option", self.name, ctypes.byref(old))
        self.old = old.value
        call("bbdm_set_option", self.name, self.value)
        return self

    def __exit__(self, *exc):
        call("bbdm_set_option", self.name, self.old)
        return False


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    call("bbdm_get_option", name.encode(), ctypes.byref(v))
    return v.value


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero code."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)
