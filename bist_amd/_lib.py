"""ctypes binding of libbist_hip.so (C ABI: include/bist_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises, and no
operator in bist_amd has a CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbist_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GATE = 0, 1, 2
# kernel families of bist_launch_count (include/bist_hip.h: BIST_K_*)
K_ST1_MFMA_FWD, K_ST1_MFMA_BWD, K_ST1_VALU, K_ST2_MFMA_FWD, K_ST2_MFMA_BWD, K_ST2_VALU, K_MHA_FWD, K_MHA_BWD_MFMA, K_MHA_BWD_VALU, K_ST1_FUSED, K_DECSTACK, K_ST1_FUSED_TRAIN, K_ST1_PBWD = range(13)


class BistGemm(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_rs", C.c_int64), ("a_ks", C.c_int64), ("b_rs", C.c_int64), ("b_ks", C.c_int64),
        ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("batch1", C.c_int32), ("batch2", C.c_int32),
        ("a_bs1", C.c_int64), ("a_bs2", C.c_int64), ("b_bs1", C.c_int64), ("b_bs2", C.c_int64),
        ("c_bs1", C.c_int64), ("c_bs2", C.c_int64), ("r_bs1", C.c_int64), ("r_bs2", C.c_int64),
        ("bias_bs2", C.c_int64),
        ("alpha", C.c_float), ("act", C.c_int32), ("res_outer", C.c_int32), ("res_inner", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_ctr", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("hint", C.c_int32), ("reserved", C.c_int32),
        ("ln_gain", C.c_void_p), ("ln_offset", C.c_void_p), ("ln_out", C.c_void_p), ("ln_ld", C.c_int64), ("ln_eps", C.c_float),
        ("ln_mode", C.c_int32),
        ("bias_bs1", C.c_int64),
    ]


class BistDrop(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint64), ("ctr", C.c_void_p)]


class BistDecLayer(C.Structure):
    _fields_ = [("ln_a", C.c_void_p * 5), ("ln_b", C.c_void_p * 5), ("Wqkv", C.c_void_p), ("bqkv", C.c_void_p),
                ("Wq", C.c_void_p * 3), ("bq", C.c_void_p * 3), ("Wo", C.c_void_p * 4), ("bo", C.c_void_p * 4),
                ("Kc", C.c_void_p * 3), ("VTc", C.c_void_p * 3), ("cmask", C.c_void_p * 3),
                ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("Lk", C.c_int32 * 3), ("LkP", C.c_int32 * 3)]


class BistLnGrad(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("a", C.c_void_p), ("da", C.c_void_p), ("db", C.c_void_p),
                ("rows", C.c_int64), ("lddy", C.c_int64), ("ldx", C.c_int64), ("eps", C.c_float)]


class BistLnSet(C.Structure):
    _fields_ = [("x", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p)]


class BistPtrDecSrc(C.Structure):
    _fields_ = [("M", C.c_void_p), ("c", C.c_void_p), ("mask", C.c_void_p), ("E", C.c_void_p), ("text", C.c_void_p), ("p_out", C.c_void_p),
                ("L", C.c_int32), ("pad_", C.c_int32)]


class BistKvFill(C.Structure):
    _fields_ = [("src", C.c_void_p), ("K", C.c_void_p), ("VT", C.c_void_p), ("Lk", C.c_int32), ("LkP", C.c_int32), ("ld", C.c_int64)]


class BistStageJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int64), ("src_row_bytes", C.c_int64), ("dst_row_bytes", C.c_int64),
                ("pad", C.c_uint64), ("pad_bytes", C.c_int32), ("reserved_", C.c_int32)]


class BistLnBwdSet(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("a", C.c_void_p), ("dx", C.c_void_p), ("da", C.c_void_p), ("db", C.c_void_p),
                ("dx_add", C.c_void_p), ("dz", C.c_void_p), ("drop_row0", C.c_uint64)]


class BistColSum(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int64), ("N", C.c_int32), ("ldx", C.c_int64)]


_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); exactly the declarations of include/bist_hip.h
SIGNATURES = {
    "bist_version": (C.c_int, []),
    "bist_last_error": (C.c_char_p, []),
    "bist_device_ok": (C.c_int, []),
    "bist_launch_count": (C.c_int64, [_I32]),
    "bist_launch_count_reset": (None, []),
    "bist_dev_set_stamps": (C.c_int, [_I32, _P]),
    "bist_dev_timestamp": (C.c_int, [_P, _P]),
    "bist_gemm": (C.c_int, [C.POINTER(BistGemm), _P]),
    "bist_gemm_pair": (C.c_int, [C.POINTER(BistGemm), C.POINTER(BistGemm), _P]),
    "bist_gemm_is_fast": (C.c_int, [C.POINTER(BistGemm)]),
    "bist_gemm_ln_ok": (C.c_int, [C.POINTER(BistGemm)]),
    "bist_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I64, _I64, _F, _I32, _P]),
    "bist_layernorm_fwd_multi": (C.c_int, [C.POINTER(BistLnSet), _I32, _I64, _I32, _I64, _I64, _F, _I32, _P]),
    "bist_layernorm_bwd_multi": (C.c_int, [C.POINTER(BistLnBwdSet), _I32, _I64, _I32, _I64, _I64, _I64, _F, _I64, C.POINTER(BistDrop), _I32, _P]),
    "bist_scaled_bias_fwd_z": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I64, _I32, _P]),
    "bist_scaled_bias_bwd_z": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I64, _I64, _I32, _P]),
    "bist_mha_core_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32,
                                    _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, C.POINTER(BistDrop), _I32, _P]),
    "bist_st_stage1_pv_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I32, C.POINTER(BistDrop), _I32, _I32, _P]),
    "bist_pack_frag_rows": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "bist_pack_frag_rows_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I32, _I32, _I32, _I32, _P]),
    "bist_st_stage1_fused_ok": (C.c_int, [_I32] * 7),
    "bist_st_stage1_fused_fwd": (C.c_int, [_P] * 9 + [_I32] * 8 + [_P]),
    "bist_st_stage1_fused_train_ok": (C.c_int, [_I32] * 7),
    "bist_st_stage1_fused_train_fwd": (C.c_int, [_P] * 12 + [C.POINTER(BistDrop), C.POINTER(BistDrop)] + [_I32] * 8 + [_P]),
    "bist_st_stage1_pv_bwd_p": (C.c_int, [_P, _I32, _P, _P, _P, _P, _I32, _P] + [_I32] * 6 + [_I64, _I64, _I32, C.POINTER(BistDrop), _I32, _P]),
    "bist_beam_step": (C.c_int, [_P] * 14 + [_I32] * 10 + [_F, _P]),
    "bist_decoder_stack_ok": (C.c_int, [_I32] * 5),
    "bist_decoder_layer_desc_bytes": (C.c_int64, []),
    "bist_decoder_stack_fwd": (C.c_int, [_P, _I32] + [_P] * 8 + [_I32, _I32, _I32, _I32, _P, _P, _I32, _P]),
    "bist_st_stage2_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, C.POINTER(BistDrop), _I32, _P]),
    "bist_scaled_bias_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "bist_scaled_bias_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "bist_embed_pe_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, C.POINTER(BistDrop), _I32, _P]),
    "bist_temporal_mask": (C.c_int, [_P, _P, _I64, _I64, _I32, _P]),
    "bist_fuse_modalities": (C.c_int, [_P, C.POINTER(C.c_void_p), _P, _I64, _I32, _I32, _I32, _P]),
    "bist_add_dropout_fwd": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, C.POINTER(BistDrop), _I32, _P]),
    "bist_add_bcast": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _P]),
    "bist_add_n": (C.c_int, [_P, _I32, _P, _I64, _I32, _P]),
    "bist_permute_ts": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "bist_pointer_mix_fwd": (C.c_int, [_P, _P, _I32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                       _P, _I64, _I32, _I32, _I32, _P]),
    "bist_log_softmax_fwd": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "bist_label_smoothing_fwd": (C.c_int, [_P, _P, _P, _I64, _I32, _F, _I32, _P]),
    "bist_sum_div": (C.c_int, [_P, _I64, _P, _P, _I32, _P]),
    "bist_epilogue_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, _I64, _I64, _I64, _I32, _F, C.c_uint64, _P, _I32, _P]),
    "bist_group_sum": (C.c_int, [_P, _P, _I64, _I32, _I64, _I32, _P]),
    "bist_group_sum_add": (C.c_int, [_P, _P, _P, _I64, _I32, _I64, _I32, _P]),
    "bist_group_sum_mask": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I64, C.POINTER(BistDrop), _I32, _P]),
    "bist_col_sum_acc": (C.c_int, [_P, _P, _I64, _I32, _I64, _I32, _P]),
    "bist_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _I64, _I64, _I64, _F, _P, _I64, _P, C.POINTER(BistDrop), _I32, _P]),
    "bist_col_sum_multi": (C.c_int, [C.POINTER(BistColSum), _I32, _I32, _P]),
    "bist_layernorm_param_grad_multi": (C.c_int, [C.POINTER(BistLnGrad), _I32, _I32, _I32, _P]),
    "bist_embed_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, C.POINTER(BistDrop), _I32, _P]),
    "bist_fuse_modalities_bwd": (C.c_int, [_P, C.POINTER(C.c_void_p), _P, _P, C.POINTER(C.c_void_p), _I64, _I32, _I32, _I32, _P]),
    "bist_mha_core_bwd": (C.c_int, [_P] * 9 + [_I32] * 5 + [_I64] * 16 + [_F, C.POINTER(BistDrop), _I32, _P]),
    "bist_st_stage1_pv_bwd": (C.c_int, [_P] * 5 + [_I32, _P] + [_I32] * 6 + [_I64, _I64, _I32, C.POINTER(BistDrop), _I32, _P]),
    "bist_st_stage2_bwd": (C.c_int, [_P] * 7 + [_I32] * 5 + [C.POINTER(BistDrop), _I32, _P]),
    "bist_pointer_mix_bwd": (C.c_int, [_P, _P, _I32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                       _P, _P, _P, _P, C.POINTER(C.c_void_p), _I64, _I32, _I32, _I32, _P]),
    "bist_log_softmax_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, _P]),
    "bist_label_smoothing_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _F, _I32, _P]),
    "bist_cast_from_f32": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "bist_add_f32_into": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "bist_adam_step": (C.c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _I32, _F, _I32, _I32, _P]),
    "bist_noam_hyper": (C.c_int, [_P, _P, _F, _F, _F, _F, _F, _F, _P]),
    "bist_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _F, _F, _F, _I32, _I32, _P]),
    "bist_text_vector_fwd": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P]),
    "bist_decoder_cache_fill": (C.c_int, [_P, _I32, _I32, _P]),
    "bist_stage_inputs": (C.c_int, [_P, _I32, _P]),
    "bist_xent_smooth_fwd": (C.c_int, [_P, _P, _I64, _I64, _I32, _F, _I32, _P, _P, _P]),
    "bist_xent_smooth_bwd": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _P, _P, _I32, _I32, _F, _I32, _P]),
    "bist_sum_div_groups": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _P]),
    "bist_stack_rows": (C.c_int, [_P, _I32, _P, _I64, _P]),
    "bist_switch_logits_fwd": (C.c_int, [_P, _I32, _P, _I64, _P, _P, _I32, _I64, _I32, _I32, _I32, _P]),
    "bist_switch_logits_bwd": (C.c_int, [_P, _I32, _P, _I64, _P, _I32, _P, _P, _P, _I64, _I32, _I32, _P, _I32, _I64, _I32, _I32, _I32, _P]),
    "bist_pointer_attn_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _F, _I32, _P]),
    "bist_pointer_attn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _F, _I32, _P]),
    "bist_pointer_decode_mix_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _P, _I64, _P, _F, _P, _I64, _I32, _I32, _I32, _P]),
    "bist_noam_hyper_pending": (C.c_int, [_P, _P, _F, _F, _F, _F, _F, _F, _P]),
    "bist_adam_apply_dev": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _F, _F, _F, _I32, _I32, _P]),
    "bist_adam_step_dev_bg": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _F, _F, _F, _I32, _I32, _I32, _P]),
    "bist_cast": (C.c_int, [_P, _P, _I64, _I32, _I32, _P]),
    "bist_graph_capture_tail": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "bist_graph_nodes": (C.c_int, [_P, C.POINTER(C.c_void_p), _I32, C.POINTER(C.c_int32)]),
    "bist_graph_split_plan": (C.c_int64, [_I32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I32, C.POINTER(C.c_int32), _I32, _I32, C.POINTER(C.c_int32), _I64]),
    "bist_graph_edges": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I32, C.POINTER(C.c_int32)]),
    "bist_graph_split_dump": (C.c_int64, [_P, C.POINTER(C.c_int32), _I64]),
    "bist_graph_split_create": (C.c_int, [_P, C.POINTER(C.c_int32), _I32, _I32, _I32, C.POINTER(C.c_void_p)]),
    "bist_graph_split_sync_words": (C.c_int64, [_P]),
    "bist_graph_split_sync_items": (C.c_int64, [_P, C.POINTER(C.c_int32), _I64]),
    "bist_graph_split_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bist_graph_split_build": (C.c_int, [_P, _P, _P, _I64]),
    "bist_graph_split_launch": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "bist_graph_split_launch_chain": (C.c_int, [_P, _I32, _P]),
    "bist_graph_split_destroy": (None, [_P]),
    "bist_graph_queues_distinct": (C.c_int, [_P, _P, _P, _I64]),
    "bist_dev_idle_wave": (C.c_int, [_P, _I64, _I32, _P]),
    "bist_flag_signal": (C.c_int, [_P, _P, _P]),
    "bist_graph_queue_pace": (C.c_int, [_P, _I32, _P, _I64, _P, C.POINTER(C.c_float)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"bist_amd: {LIB_PATH} is missing -- build it with `make -C bist_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class BistError(RuntimeError):
    pass


AFTER_LAUNCH = None      # bist_amd.graphsplit.Labels: called after every checked library call while a capture is being labelled


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise BistError(f"{what} failed (rc={rc}): {lib.bist_last_error().decode()}")
    if AFTER_LAUNCH is not None:
        AFTER_LAUNCH()
