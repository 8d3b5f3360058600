"""ctypes binding of ``libmicronet_hip.so`` (C ABI declared in ``include/micronet_hip.h``).

The product path has NO fallback: if the gfx950 library is missing, or it is the CPU emulation build the unit
tests use, ``get_lib()`` raises.  Every entry point returns 0 or a negative errno-style code; ``check`` turns a
failure into a Python exception carrying ``mn_last_error()``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MN_LIB_PATH") or os.path.join(HERE, "lib", "libmicronet_hip.so")   # MN_LIB_PATH: ablation builds of the same library

MN_ACTQ_NONE, MN_ACTQ_DOREFA, MN_ACTQ_IAO, MN_ACTQ_SIGN8, MN_ACTQ_CODE8 = 0, 1, 2, 3, 4
MN_ALGO_AUTO, MN_ALGO_DIRECT, MN_ALGO_MFMA, MN_ALGO_QGEMM = 0, 1, 2, 3
MN_WQ_REAL, MN_WQ_TERNARY, MN_WQ_DOREFA, MN_WQ_IAO = 0, 1, 2, 3
MN_ACTQ_X_IS_CODE = 1
MN_ACTQ_CODES_GIVEN = 2
MN_ENOTSUP = -95


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "C", "H", "W", "O", "KH", "KW", "stride_h", "stride_w", "pad_h", "pad_w",
                                         "dil_h", "dil_w", "groups", "in_shuffle")]


class ActQ(C.Structure):
    _fields_ = [("mode", C.c_int32), ("bits", C.c_int32), ("q_type", C.c_int32), ("flags", C.c_int32),
                ("qp", C.c_void_p), ("codes", C.c_void_p), ("stats", C.c_void_p), ("dx_add", C.c_void_p), ("ste_mask", C.c_void_p), ("acc_mm", C.c_void_p)]


class WQ(C.Structure):
    """mn_wq: how the fake-quantised fp32 weights factor into integer codes x per-channel scale."""
    _fields_ = [("mode", C.c_int32), ("bits", C.c_int32), ("q_type", C.c_int32), ("per_channel", C.c_int32),
                ("scale", C.c_void_p), ("packed_fwd", C.c_void_p), ("packed_bwd", C.c_void_p)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("bytes", C.c_double), ("flops", C.c_double)]


class AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("weight_decay", C.c_float)]


_P, _I, _L, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
_G, _A, _W = C.POINTER(ConvGeom), C.POINTER(ActQ), C.POINTER(WQ)

# name -> (restype, argtypes); must list every symbol declared in include/micronet_hip.h
PROTOTYPES = {
    "mn_version": (_I, []),
    "mn_last_error": (C.c_char_p, []),
    "mn_last_kernel": (C.c_char_p, []),
    "mn_profile_next": (None, [_P, _P]),
    "mn_profile_enable": (None, [_I]),
    "mn_profile_collect": (_I, [C.POINTER(ProfEntry), _I]),
    "mn_is_emulation": (_I, []),
    "mn_dense_grad_terms": (_I, []),
    "mn_round_half_away": (_I, [_P, _P, _L, _P]),
    "mn_dorefa_act_fwd": (_I, [_P, _P, _L, _I, _P]),
    "mn_dorefa_act_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "mn_dorefa_w_ws_floats": (_L, [_L]),
    "mn_dorefa_w_fwd": (_I, [_P, _P, _L, _I, _P, _P]),
    "mn_dorefa_w_bwd": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "mn_binact_fwd": (_I, [_P, _P, _L, _P]),
    "mn_binact_bwd": (_I, [_P, _P, _P, _L, _P]),
    "mn_ternary_w_fwd": (_I, [_P, _P, _P, _L, _L, _P]),
    "mn_ternary_w_bwd": (_I, [_P, _P, _P, _P, _L, _L, _P]),
    "mn_ternary_w_fwd_multi": (_I, [_P, _P, _P, _P, _P, C.c_int32, _P]),
    "mn_ternary_w_bwd_multi": (_I, [_P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "mn_binary_w_fwd": (_I, [_P, _P, _P, _L, _L, _L, _P]),
    "mn_binary_w_bwd": (_I, [_P, _P, _P, _P, _L, _L, _P]),
    "mn_binary_w_fwd_multi": (_I, [_P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "mn_binary_w_bwd_multi": (_I, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "mn_iao_observe_ws_floats": (_L, [_L, _L]),
    "mn_iao_observe": (_I, [_P, _L, _L, _I, _I, _D, _P, _P, _P, _P]),
    "mn_iao_qparams": (_I, [_P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mn_iao_fq_fwd": (_I, [_P, _P, _L, _L, _P, _I, _I, _I, _P]),
    "mn_iao_fq_bwd": (_I, [_P, _P, _P, _L, _L, _P, _I, _I, _I, _P]),
    "mn_iao_union_range": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "mn_bn_stats_ws_floats": (_L, [_L, _L, _L]),
    "mn_bn_stats_fwd": (_I, [_P, _L, _L, _L, _P, _P, _P]),
    "mn_bn_stats_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _P]),
    "mn_bnsign_ws_floats": (_L, [_L]),
    "mn_bnsign_fwd": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P]),
    "mn_bnsign_fwd_i8": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P]),
    "mn_maxpool2x2_sign8_fwd": (_I, [_P, _L, _L, _L, _P, _P]),
    "mn_maxpool2x2_sign8_bwd": (_I, [_P, _P, _L, _L, _L, _P, _P]),
    "mn_bnsign_bwd_sums": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _P, _P, _P, _P, _P]),
    "mn_conv2d_bwd_weight_first_bn": (_I, [_G, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_weight_first_qa": (_I, [_G, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_first_gram_bnstats": (_I, [_G, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P]),
    "mn_bnsign_apply": (_I, [_P, _L, _L, _L, _P, _P, _P, _P, _I, _P]),
    "mn_qa_fwd_f32_mask": (_I, [_P, _P, _L, _L, _L, _L, _I, _P, _P, _P]),
    "mn_conv2d_first_bnact_fwd": (_I, [_G, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "mn_conv2d_bwd_first_mask_gram": (_I, [_G, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_first_xgram_ws_bytes": (_L, [_G]),
    "mn_conv2d_first_xgram": (_I, [_G, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_first_bn_gram": (_I, [_G, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_first_qa_gram": (_I, [_G, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_bnsign_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P]),
    "mn_maxpool2x2_f32_supported": (_I, [_L, _L]),
    "mn_maxpool2x2_f32_fwd": (_I, [_P, _L, _L, _L, _P, _P, _P]),
    "mn_maxpool2x2_f32_bwd": (_I, [_P, _P, _L, _L, _L, _P, _P]),
    "mn_bnrelu_fwd": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P]),
    "mn_bnrelu_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P]),
    "mn_avgpool_global_fwd": (_I, [_P, _L, _L, _P, _P]),
    "mn_avgpool_global_bwd": (_I, [_P, _L, _L, _P, _P]),
    "mn_bn2d_fwd": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P]),
    "mn_bn2d_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P]),
    "mn_qconv_bnsign_supported": (_I, [_G, _W]),
    "mn_qconv_bnsign_ws_bytes": (_L, [_G]),
    "mn_qconv_bnsign_stash_supported": (_I, [_G, _W]),
    "mn_qconv_bnsign_stash_ws_bytes": (_L, [_G]),
    "mn_qconv_bnsign_stash_chan_rows": (_I, [_G]),
    "mn_qconv_bnsign_fwd": (_I, [_G, _W, _P, _P, _P, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _L, _P]),
    "mn_qconv_bnsign_fwd_stash": (_I, [_G, _W, _P, _P, _P, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_qconv_bnsign_fwd_stash_pool_supported": (_I, [_G, _W]),
    "mn_qconv_bnsign_fwd_stash_pool": (_I, [_G, _W, _P, _P, _P, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_bnh_bwd_sums": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P]),
    "mn_bnh_bwd_apply": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _P, _P]),
    "mn_conv2d_bnh_supported": (_I, [_G, _W]),
    "mn_conv2d_bwd_data_bnh": (_I, [_G, _W, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_weight_bnh": (_I, [_G, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_bnh_supported": (_I, [_G, _W, _I]),
    "mn_conv2d_bwd_bnh_ws_bytes": (_L, [_G]),
    "mn_conv2d_bwd_bnh": (_I, [_G, _W, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_bnh_up_splits": (_I, [_G, _W, _I, _L]),
    "mn_conv2d_bwd_bnh_up": (_I, [_G, _W, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "mn_bnh_bwd_sums_final": (_I, [_P, _I, _L, _L, _L, _L, _P, _P, _P, _P]),
    "mn_conv2d_bwd_bnh_up9_splits": (_I, [_G, _W, _L]),
    "mn_conv2d_bwd_bnh_up9": (_I, [_G, _W, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "mn_conv2d_bwd_codes": (_I, [_G, _W, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_qa": (_I, [_G, _W, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_codes_up": (_I, [_G, _W, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P, _P, _I, _P, _P]),
    "mn_conv2d_bwd_qa_up": (_I, [_G, _W, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _L, _P, _P, _I, _P, _P]),
    "mn_qa_bwd_sums_final": (_I, [_P, _I, _L, _P, _P, _P, _P]),
    "mn_conv2d_bnh_pool_supported": (_I, [_G, _W]),
    "mn_conv2d_bwd_data_bnh_pool": (_I, [_G, _W, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P]),
    "mn_conv2d_bwd_weight_bnh_pool": (_I, [_G, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_qconv_bnsign_bwd": (_I, [_G, _W, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_qconv_bnsign_bwd_pooled": (_I, [_G, _W, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_signconv1x1_small_supported": (_I, [_L, _L, _L]),
    "mn_signconv1x1_small_fwd": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _P]),
    "mn_codeconv1x1_small_fwd": (_I, [_P, _I, _P, _P, _P, _L, _L, _L, _L, _P]),
    "mn_conv1x1_small_bwd_data": (_I, [_P, _P, _P, _L, _L, _L, _L, _P]),
    "mn_iao_fq_act_fwd": (_I, [_P, _P, _L, _P, _I, _I, _I, C.c_float, _P]),
    "mn_iao_fq_act_bwd": (_I, [_P, _P, _P, _L, _P, _I, _I, _I, C.c_float, _P]),
    "mn_iao_fq_avgpool_supported": (_I, [_L, _L, _L]),
    "mn_iao_fq_avgpool_fwd": (_I, [_P, _P, _L, _L, _L, _L, _P, _I, _I, _P]),
    "mn_iao_fq_avgpool_bwd": (_I, [_P, _P, _P, _L, _L, _L, _L, _P, _I, _I, _P]),
    "mn_kth_abs_ws_bytes": (_L, []),
    "mn_hist_observe": (_I, [_P, _L, _L, _I, C.c_double, _P, _P, _P, _P]),
    "mn_qconv_bnq_supported": (_I, [_G, _W, _I]),
    "mn_qconv_bnq_stash_bits": (_I, [_G, _W, _I]),
    "mn_qconv_bnq_ws_bytes": (_L, [_G]),
    "mn_qconv_bnq_fwd_stash": (_I, [_G, _W, _P, _I, _P, _P, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mn_bn_save_stats": (_I, [_P, _L, _L, _L, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P]),
    "mn_qa_supported": (_I, [_L, _L, _I]),
    "mn_qa_ws_floats": (_L, [_L]),
    "mn_qa_chan_from_save": (_I, [_P, _P, _P, _L, _P, _P]),
    "mn_qa_fwd": (_I, [_I, _P, _P, _L, _L, _L, _L, _I, _I, _P, _P, _P]),
    "mn_qa_bwd_sums": (_I, [_I, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _P, _P, _P, _P, _P]),
    "mn_qa_bwd_apply": (_I, [_I, _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P]),
    "mn_qa_bwd": (_I, [_I, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mn_conv2d_iao_codes_bytes": (_L, [_G, _A, _W]),
    "mn_conv2d_bwd_data_add_supported": (_I, [_G, _A, _W]),
    "mn_conv2d_iao_stats_rows": (_L, [_G, _A, _W]),
    "mn_bn_fwd_acc": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P, _I, _P, _L, _P, _P, _L, _P, _P]),
    "mn_bn_acc_prep": (_I, [_L, _L, _L, _P, _P, C.c_float, C.c_float, _P, _P, _P, _I, _P, _P, _L, _P, _P, _L, _P, _P, _P]),
    "mn_bnrelu_gap_supported": (_I, [_L, _L, _L]),
    "mn_bnrelu_gap_fwd": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    "mn_bnrelu_gap_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _P, _P, _P, _P]),
    "mn_cross_entropy_fwd": (_I, [_P, _P, _L, _L, _L, _P, _P, _P]),
    "mn_scale_by": (_I, [_P, _P, _P, _L, _P]),
    "mn_bn_apply_codes": (_I, [_P, _L, _L, _L, _P, _P, _P, _I, _P, _I, _P, _P, _P]),
    "mn_bn_apply": (_I, [_P, _L, _L, _L, _P, _P, _P, _I, _P, _P]),
    "mn_iao_qadd_bn_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "mn_iao_qadd_bn_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mn_iao_w_fwd_multi": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _I, _I, _P]),
    "mn_iao_w_bwd_multi": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mn_iao_qadd_ws_floats": (_L, []),
    "mn_iao_qadd_mm_count": (_L, [_L]),
    "mn_iao_qadd_fwd_mm": (_I, [_P, _P, _P, _L, _P, _I, _I, _I, _P, _P]),
    "mn_iao_observe_partials": (_I, [_P, _L, _I, _I, _D, _P, _P, _P]),
    "mn_iao_observe_partials_qparams": (_I, [_P, _L, _I, _I, _D, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mn_bnrelu_mm_count": (_L, [_L, _L, _L]),
    "mn_bnrelu_fwd_mm": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mn_iao_qadd_observe": (_I, [_P, _P, _L, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "mn_iao_qadd_observe_partials": (_I, [_P, _L, _P, _L, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mn_bn2d_fwd_mm": (_I, [_P, _L, _L, _L, _P, _P, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mn_iao_qadd_fwd": (_I, [_P, _P, _P, _L, _P, _I, _I, _I, _P]),
    "mn_iao_qadd_bwd": (_I, [_P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _P]),
    "mn_qd_packed_bytes": (_L, [_G]),
    "mn_qd_pack_multi": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "mn_qg_packed_bytes": (_L, [_G, _I]),
    "mn_qg_pack_multi": (_I, [_I, _P, _P, _P, _P, _P, _P]),
    "mn_qr_ws_floats": (_L, [_L]),
    "mn_qr_fwd": (_I, [_I, _P, _P, _I, _P, _P, _L, _L, _L, _L, _I, _P, _P, _P]),
    "mn_qr_bwd_sums": (_I, [_I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mn_qr_bwd_apply": (_I, [_I, _P, _P, _P, _I, _P, _P, _P, _P, _L, _L, _L, _L, _I, _P, _P, _P]),
    "mn_qr_bwd": (_I, [_I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mn_qlinear_supported": (_I, [_L, _L, _L]),
    "mn_qlinear_fwd": (_I, [_A, _P, _P, _P, _P, _L, _L, _L, _P]),
    "mn_qlinear_bwd_data": (_I, [_A, _P, _P, _P, _P, _L, _L, _L, _P]),
    "mn_qlinear_bwd_weight": (_I, [_A, _P, _P, _P, _P, _L, _L, _L, _P]),
    "mn_cifar_augment": (_I, [_P, _L, _P, _P, _P, _P, _L, _L, _L, _L, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "mn_iao_bnfold_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, C.c_float, _L, _L, _P, _P, _P]),
    "mn_iao_bnfold_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mn_tanh_f32": (_I, [_P, _P, _L, _P]),
    "mn_dorefa_w_fwd_multi": (_I, [_P, _P, _P, _P, C.c_int32, _I, _P]),
    "mn_dorefa_w_bwd_multi": (_I, [_P, _P, _P, _P, _P, C.c_int32, _I, _P]),
    "mn_dorefa_w_fwd_multi_cached": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "mn_dorefa_w_bwd_multi_cached": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "mn_adam_step": (_I, [C.POINTER(AdamTensor), _I, _I, C.c_float, C.c_float, C.c_float, _P]),
    "mn_adam_step_dev": (_I, [C.POINTER(AdamTensor), _I, _P, _P, C.c_float, C.c_float, C.c_float, _P]),
    "mn_conv2d_ws_bytes": (_L, [_G, _I, _I]),
    "mn_conv2d_mfma_supported": (_I, [_G, _I]),
    "mn_conv2d_first_supported": (_I, [_G, _I]),
    "mn_conv2d_qgemm_supported": (_I, [_G, _A, _W, _I]),
    "mn_conv2d_fwd": (_I, [_G, _A, _W, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mn_conv2d_bwd_data": (_I, [_G, _A, _W, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mn_conv2d_bwd_weight": (_I, [_G, _A, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mn_conv2d_fwd_act_mm_count": (_L, [_G, _A, _W]),
    "mn_conv2d_fwd_act": (_I, [_G, _A, _W, _P, _P, _P, _P, _I, _P, _P, _L, _P]),
    "mn_iao_fq_maxpool2x2_supported": (_I, [_L, _L]),
    "mn_iao_fq_maxpool2x2_mm_count": (_L, [_L, _L, _L]),
    "mn_iao_fq_maxpool2x2_fwd": (_I, [_P, _L, _L, _L, _P, _I, _I, _P, _P, _P, _P]),
    "mn_iao_fq_maxpool2x2_bwd": (_I, [_P, _P, _P, _L, _L, _L, _P, _I, _I, _I, _P, _P]),
    "mn_add_relu_mask": (_I, [_P, _P, _P, _P, _L, _P]),
    "mn_relu_mm_count": (_L, [_L]),
    "mn_relu_mm": (_I, [_P, _P, _L, _P, _P]),
    "mn_iaobf_gram_supported": (_I, [_G]),
    "mn_iaobf_gram_ws_bytes": (_L, [_G]),
    "mn_iaobf_gram": (_I, [_G, _P, _P, _P, _P, _L, _P]),
    "mn_iaobf_gram_stats": (_I, [_P, _P, _P, _P, _L, _L, _L, _D, _P, _P, _P]),
    "mn_iaobf_prep_fwd": (_I, [_P, _P, _P, _P, _L, _L, _P, C.c_float, C.c_float, _I, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mn_iaobf_prep_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P, _P, _D, C.c_float, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mn_iaobf_bwd_data_supported": (_I, [_G]),
    "mn_iaobf_bwd_data_ws_bytes": (_L, [_G]),
    "mn_iaobf_bwd_data": (_I, [_G, _A, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _P]),
    "mn_iaobf_g3_supported": (_I, [_G]),
    "mn_iaobf_g3_ws_bytes": (_L, [_G]),
    "mn_iaobf_g3_mm_count": (_L, [_G]),
    "mn_iaobf_g3_stats": (_I, [_G, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "mn_iaobf_g3_fwd": (_I, [_G, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P]),
    "mn_iaobf_g3_dyraw": (_I, [_G, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "mn_iaobf_g3_bwd_weight": (_I, [_G, _P, _P, _P, _P, _I, _I, _P, _P, _P, _L, _P]),
    "mn_iaobf_g3_bwd_data": (_I, [_G, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P]),
    "mn_iaobf_thin_supported": (_I, [_G]),
    "mn_iaobf_thin_mm_count": (_L, [_G]),
    "mn_iaobf_thin_pack": (_I, [_P, _L, _L, _P, _P]),
    "mn_iaobf_thin_fwd": (_I, [_G, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "mn_iaobf_thin_bwd_weight": (_I, [_G, _P, _P, _P, _I, _I, _P, _P, _P]),
    "mn_iaobf_thin_bwd_data": (_I, [_G, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P]),
}


class MicronetHipError(RuntimeError):
    pass


class Lib:
    def __init__(self, path):
        self.path = path
        self.cdll = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(self.cdll, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, rc, what=""):
        if rc != 0:
            msg = self.mn_last_error()
            raise MicronetHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
        return rc


_LIB = None


def load(path):
    return Lib(path)


def get_lib():
    """The gfx950 library, or an exception -- never a CPU fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MicronetHipError(
                "libmicronet_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m micronet_amd.build`; there is no CPU fallback." % LIB_PATH)
        lib = Lib(LIB_PATH)
        if lib.mn_is_emulation():
            raise MicronetHipError("%s is an emulation build; the product requires the gfx950 build" % LIB_PATH)
        _LIB = lib
    return _LIB
