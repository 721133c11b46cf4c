"""ctypes binding of libagile3d_hip.so (the C ABI declared in include/agile3d_hip.h).

The product path has no fallback: if the shared object is missing or a symbol cannot be
resolved, importing this module's ``load()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("A3D_LIB_PATH") or os.path.join(HERE, "libagile3d_hip.so")   # override: A/B of two builds

A3D_NUM_LEVELS = 5
A3D_MAX_QUERIES = 256
A3D_MAX_DEC_LAYERS = 8
OP_STEM, OP_CONV3, OP_DOWN, OP_UP, OP_LINEAR = 0, 1, 2, 3, 4
BUF_NONE, BUF_EXT_OUT = -1, -2
(TAB_XYZB, TAB_NBR27, TAB_GMASK27, TAB_CHILD8, TAB_GMASKDOWN, TAB_UP8, TAB_GMASKUP, TAB_UPROWS,
 TAB_ORIGROW, TAB_ORDER27, TAB_PRE27, TAB_PREDOWN, TAB_PREUP) = range(13)

c_float_p = C.c_void_p  # device pointers travel as integers


class BufDesc(C.Structure):
    _fields_ = [("level", C.c_int32), ("channels", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("level_in", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("in_buf", C.c_int32), ("in_coff", C.c_int32), ("out_buf", C.c_int32), ("out_coff", C.c_int32),
                ("res_buf", C.c_int32), ("res_coff", C.c_int32), ("relu", C.c_int32),
                ("kernel_volume", C.c_int32),
                ("w_dev", C.c_void_p), ("scale_dev", C.c_void_p), ("shift_dev", C.c_void_p),
                ("proj_buf", C.c_int32), ("proj_coff", C.c_int32), ("proj_cin", C.c_int32), ("reserved_", C.c_int32),
                ("head_w_dev", C.c_void_p), ("head_bias_dev", C.c_void_p), ("head_cout", C.c_int32), ("reserved2_", C.c_int32)]


_LAYER_FIELDS = ["c2s_in_w", "c2s_in_b", "c2s_out_w", "c2s_out_b", "c2s_norm_w", "c2s_norm_b",
                 "c2c_in_w", "c2c_in_b", "c2c_out_w", "c2c_out_b", "c2c_norm_w", "c2c_norm_b",
                 "ffn_w1", "ffn_b1", "ffn_w2", "ffn_b2", "ffn_norm_w", "ffn_norm_b",
                 "s2c_in_w", "s2c_in_b", "s2c_out_w", "s2c_out_b", "s2c_norm_w", "s2c_norm_b",
                 "c2s_wk_packed", "c2s_wv_packed", "s2c_wq_packed", "s2c_wo_packed", "query_pack"]


class DecoderLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class DecoderWeights(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_bg_queries", C.c_int32), ("dim_ff", C.c_int32),
                ("layers", DecoderLayer * A3D_MAX_DEC_LAYERS),
                ("decoder_norm_w", C.c_void_p), ("decoder_norm_b", C.c_void_p),
                ("mask_w0", C.c_void_p), ("mask_b0", C.c_void_p), ("mask_w2", C.c_void_p), ("mask_b2", C.c_void_p),
                ("bg_query_feat", C.c_void_p), ("bg_query_pos", C.c_void_p),
                ("gauss_B", C.c_void_p), ("time_table", C.c_void_p), ("mask_pack", C.c_void_p)]


class PackJob(C.Structure):
    """a3d_pack_job: one weight of a3d_pack_conv_weights_multi."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("K", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("src_cin", C.c_int32), ("src_cout", C.c_int32), ("transposed", C.c_int32), ("flip", C.c_int32),
                ("c0", C.c_int32), ("chunk0", C.c_int32), ("pad_", C.c_int32)]


class ProfEntry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("id", "bn", "kernel_volume", "cin", "cout", "n_out", "table", "level",
                                         "ksplit")] + [("ms", C.c_float)]


class DecoderSample(C.Structure):
    """a3d_decoder_sample: one batch sample of a3d_decoder_forward_batch."""
    _fields_ = [("feats128_dev", C.c_void_p), ("posenc_dev", C.c_void_p), ("n", C.c_int64),
                ("click_row", C.POINTER(C.c_int32)), ("click_obj", C.POINTER(C.c_int32)),
                ("click_time", C.POINTER(C.c_int32)), ("n_clicks", C.c_int32), ("n_objects", C.c_int32),
                ("logits_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("kv0_dev", C.c_void_p), ("kv0_state", C.c_int32), ("kv0_blocks", C.c_int32)]


class ClickSample(C.Structure):
    """a3d_click_sample: one sample of a3d_click_clusters_batch."""
    _fields_ = [("xyz_dev", C.c_void_p), ("pred_dev", C.c_void_p), ("labels_dev", C.c_void_p), ("n", C.c_int64),
                ("out_dev", C.c_void_p), ("n_out_dev", C.c_void_p), ("max_out", C.c_int32), ("workspace_dev", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("order_dev", C.c_void_p), ("inv_dev", C.c_void_p)]


class ArgmaxSample(C.Structure):
    """a3d_argmax_sample: one sample of a3d_argmax_labels_batch."""
    _fields_ = [("logits_dev", C.c_void_p), ("n", C.c_int64), ("n_classes", C.c_int32), ("n_clicks", C.c_int32),
                ("click_row", C.POINTER(C.c_int32)), ("click_obj", C.POINTER(C.c_int32)), ("pred_dev", C.c_void_p)]


class IouSample(C.Structure):
    """a3d_iou_sample: one sample of a3d_iou_counts_batch."""
    _fields_ = [("pred_dev", C.c_void_p), ("n_pred", C.c_int64), ("inverse_map_dev", C.c_void_p), ("labels_dev", C.c_void_p),
                ("n_full", C.c_int64)]


class ClickCluster(C.Structure):
    _fields_ = [("cluster_id", C.c_int32), ("row", C.c_int32), ("label", C.c_int32), ("pred", C.c_int32),
                ("error_size", C.c_float)]


A3D_MAX_CLICKS = 256
PROF_DENSE = 11
PROF_NAMES = ["spconv", "splitk_epilogue", "stem", "c2s_attn", "query_chain", "s2c_attn", "ln_mask", "posenc",
              "scene_sort_levels", "scene_tables", "click_simulator", "dense_gemm"]

# name -> (restype, argtypes): every symbol include/agile3d_hip.h declares
SYMBOLS = {
    "a3d_version": (C.c_int, []),
    "a3d_last_error": (C.c_char_p, []),
    "a3d_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_profile_enable": (C.c_int, [C.c_int]),
    "a3d_profile_read": (C.c_int, [C.POINTER(ProfEntry), C.c_int]),
    "a3d_scene_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "a3d_sort_pairs_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "a3d_sort_pairs_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "a3d_scene_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_void_p)]),
    "a3d_scene_destroy": (None, [C.c_void_p]),
    "a3d_scene_level_size": (C.c_int64, [C.c_void_p, C.c_int]),
    "a3d_scene_batch_ranges": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "a3d_scene_grid_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "a3d_scene_table": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "a3d_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "a3d_conv_weight_packed_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "a3d_pack_conv_weights_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "a3d_conv_deep_mode": (C.c_int, [C.c_int]),
    "a3d_program_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(BufDesc), C.c_int, C.POINTER(Op), C.c_int]),
    "a3d_program_buffer_offset": (C.c_size_t, [C.c_void_p, C.POINTER(BufDesc), C.c_int, C.c_int]),
    "a3d_program_run": (C.c_int, [C.c_void_p, C.POINTER(BufDesc), C.c_int, C.POINTER(Op), C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_posenc_fourier": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "a3d_posenc_batch_workspace_bytes": (C.c_size_t, [C.c_int]),
    "a3d_posenc_fourier_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_decoder_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "a3d_decoder_query_pack_floats": (C.c_size_t, [C.c_int32]),
    "a3d_decoder_mask_pack_floats": (C.c_size_t, []),
    "a3d_decoder_pack_query_weights": (C.c_int, [C.POINTER(DecoderWeights), C.c_int32, C.c_void_p, C.c_void_p]),
    "a3d_scene_wgrad_lists_bytes": (C.c_size_t, [C.c_void_p]),
    "a3d_scene_build_wgrad_lists": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_conv_wgrad_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "a3d_conv_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_conv_apply_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "a3d_conv_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_conv_state_bytes": (C.c_size_t, []),
    "a3d_conv_apply_acc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "a3d_conv_bn_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "a3d_conv_dgrad_bn": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_conv_bn_train_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int,
                                            C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_stem_wgrad_workspace_bytes": (C.c_size_t, [C.c_int]),
    "a3d_stem_wgrad_scene_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "a3d_stem_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "a3d_bn_local_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_bn_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "a3d_bn_backward_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_bn_backward_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "a3d_layernorm_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "a3d_layernorm_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "a3d_linear_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "a3d_linear_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_linear_wgrad_into_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "a3d_linear_wgrad_into": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_attn_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "a3d_softmax_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "a3d_softmax_rows_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "a3d_softmax_cols": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "a3d_softmax_cols_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "a3d_attn_apply_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "a3d_attn_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_group_max": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "a3d_next_layer_mask_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "a3d_next_layer_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_group_max_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "a3d_flash_c2s_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "a3d_flash_c2s_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_flash_c2s_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "a3d_flash_s2c_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "a3d_flash_s2c_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "a3d_flash_s2c_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_sum_squares_workspace_bytes": (C.c_size_t, []),
    "a3d_sum_squares": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_sum_squares_accumulate": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "a3d_mt_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "a3d_sum_squares_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_adamw_step_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p]),
    "a3d_bn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "a3d_bn_train_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_bn_train_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "a3d_column_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    "a3d_decoder_forward_batch": (C.c_int, [C.POINTER(DecoderWeights), C.POINTER(DecoderSample), C.c_int, C.c_void_p]),
    "a3d_decoder_forward": (C.c_int, [C.POINTER(DecoderWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "a3d_argmax_labels": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.c_int, C.c_void_p, C.c_void_p]),
    "a3d_iou_counts": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "a3d_argmax_labels_batch_workspace_bytes": (C.c_size_t, [C.c_int]),
    "a3d_argmax_labels_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_iou_counts_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "a3d_click_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "a3d_click_clusters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_click_clusters_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "a3d_click_spatial_order_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "a3d_click_spatial_order": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_mask_losses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_quantize_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "a3d_sparse_quantize": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_int64), C.c_void_p, C.c_size_t, C.c_void_p]),
    "a3d_click_loss_weights": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int, C.c_float,
                                         C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
}

_lib = None


class A3DError(RuntimeError):
    pass


ABI_VERSION = 3   # include/agile3d_hip.h: A3D_ABI_VERSION


def load():
    """dlopen the in-tree library and bind every symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise A3DError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.a3d_version()
    if got != ABI_VERSION:   # struct layouts / buffer contracts of another revision: refuse before the first real call
        raise A3DError(f"{LIB_PATH} speaks interface version {got}, this binding was written against {ABI_VERSION} "
                       "(include/agile3d_hip.h: A3D_ABI_VERSION); rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().a3d_last_error()
        raise A3DError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
