"""ctypes binding of libhtrvt_hip.so (the C ABI declared in include/htrvt.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is
missing or a symbol cannot be resolved, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- first: libhtrvt_hip.so must bind to the HIP runtime PyTorch-ROCm already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HTRVT_LIB") or os.path.join(_HERE, "lib", "libhtrvt_hip.so")   # HTRVT_LIB: A/B builds of the library

F32, BF16 = 0, 1
KMAJOR, MNMAJOR = 0, 1
GATHER_NONE, GATHER_CONV_FWD, GATHER_CONV_DGRAD, GATHER_CONV_WGRAD = 0, 1, 2, 3

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class RelayoutJob(C.Structure):
    """include/htrvt.h HtrvtRelayoutJob: one tensor of a table-driven re-layout launch (csrc/relayout.hip)"""
    _fields_ = [
        ("src", vp), ("dst0", vp), ("dst1", vp), ("kind", i32), ("d0", i32), ("d1", i32),
        ("taps", i32), ("cpad_in", i32), ("cpad_out", i32), ("row_taps", i32), ("tap0", i32),
        ("tile0", i32), ("tiles_x", i32), ("reserved_", i32 * 2),
    ]


RELAYOUT_PACK_CONV, RELAYOUT_CAST_TRANSPOSE, RELAYOUT_UNPACK_WGRAD, RELAYOUT_MAX_JOBS, RELAYOUT_ARG_JOBS = 0, 1, 2, 64, 48
HTRVT_ADAMW_SCALARS = 8


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("a_layout", i32), ("b_layout", i32), ("gather", i32),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("batch", i32), ("batch_inner", i32),
        ("sA_o", i64), ("sA_i", i64), ("sB_o", i64), ("sB_i", i64), ("sC_o", i64), ("sC_i", i64),
        ("split_k", i32), ("splitk_ws", vp),
        ("nB", i32), ("Hi", i32), ("Wi", i32), ("Ci", i32), ("Ho", i32), ("Wo", i32), ("Co", i32),
        ("kh", i32), ("kw", i32), ("sh", i32), ("sw", i32), ("ph", i32), ("pw", i32), ("Cpad", i32),
        ("cls_h", i32), ("cls_w", i32),
        ("A2", vp),
        ("alpha", f32), ("act", i32), ("c_f32", i32), ("accumulate", i32), ("tile", i32),
        ("bias", vp), ("colscale", vp), ("preact", vp), ("residual", vp), ("colstats", vp),
        ("relu_src", vp), ("bnb_x", vp * 2), ("bnb_mean", vp * 2), ("bnb_rstd", vp * 2), ("bnb_partial", vp * 2),
        ("bnb_tile0", i32), ("relu_scale", vp), ("relu_shift", vp), ("relu_bits", i32),
        ("A", vp), ("B", vp), ("C", vp),
    ]


# symbol -> (restype, argtypes); every symbol of include/htrvt.h must be listed here
PROTOTYPES = {
    "htrvt_version": (i32, []),
    "htrvt_last_error": (C.c_char_p, []),
    "htrvt_last_kernel": (C.c_char_p, []),
    "htrvt_gemm": (i32, [C.POINTER(GemmDesc), vp]),
    "htrvt_gemm_num_mtiles": (i32, [C.POINTER(GemmDesc)]),
    "htrvt_gemm_wgrad_tiling": (i32, [C.POINTER(GemmDesc), C.POINTER(i32), C.POINTER(i32)]),
    "htrvt_gemm_dgrad_merged_tiles": (i32, [C.POINTER(GemmDesc)]),
    "htrvt_img_stats": (i32, [vp, vp, i32, i32, f32, i32, vp]),
    "htrvt_conv1_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_stem_stats_rows": (i32, [i32, i32]),
    "htrvt_stem_stats": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "htrvt_stem_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_bn_finalize": (i32, [vp, i32, i32, f32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "htrvt_bn_eval_coeffs": (i32, [vp, vp, vp, vp, f32, vp, vp, vp, i32, vp]),
    "htrvt_bn_apply": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "htrvt_bn_apply_mask": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "htrvt_bn_relu_maxpool": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "htrvt_pool_tokens": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "htrvt_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]),
    "htrvt_softmax_rows": (i32, [vp, vp, i64, i32, i32, vp, i64, vp]),
    "htrvt_attn_supported": (i32, [i32, i32, i32]),
    "htrvt_attn_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "htrvt_attn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "htrvt_seq_whiten_fwd": (i32, [vp, vp, vp, i32, i32, f32, i32, vp]),
    "htrvt_seq_whiten_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "htrvt_layernorm_bwd_blocks": (i32, [i64]),
    "htrvt_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "htrvt_softmax_bwd_rows": (i32, [vp, vp, vp, i64, i32, f32, i32, vp]),
    "htrvt_colsum_workspace_floats": (C.c_size_t, [i64, i32]),
    "htrvt_colsum": (i32, [vp, i64, i32, i64, vp, vp, i32, i32, vp, vp]),
    "htrvt_rowsum_f32": (i32, [vp, i32, i32, vp, vp]),
    "htrvt_bn_bwd_blocks": (i32, [i64]),
    "htrvt_bn_bwd_reduce": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "htrvt_bn_bwd_finalize": (i32, [vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "htrvt_bn_bwd_apply": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "htrvt_bn_bwd_apply2": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "htrvt_maxpool_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "htrvt_pool_tokens_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "htrvt_conv1_bwd_rows": (i32, [i32, i32]),
    "htrvt_conv1_bwd_row_floats": (i32, [i32]),
    "htrvt_conv1_bwd": (i32, [vp] * 12 + [i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_conv1_wgrad_blocks": (i32, [i32, i32]),
    "htrvt_conv1_wgrad": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_pack_conv_weight": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_pack_conv_weight_slots": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_unpack_conv_wgrad": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "htrvt_cast_f32": (i32, [vp, vp, i64, i32, vp]),
    "htrvt_relayout_plan": (i32, [vp, i32]),
    "htrvt_relayout": (i32, [vp, i32, i32, i32, vp]),
    "htrvt_relayout_host": (i32, [vp, i32, i32, i32, vp]),
    "htrvt_relpos_bias_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_relpos_bias_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_cast_transpose_f32": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "htrvt_split_bf16": (i32, [vp, i64, i32, i64, vp, i32, i32, i32, vp, vp, vp]),
    "htrvt_elementwise_f32": (i32, [vp, vp, vp, i64, i32, vp]),
    "htrvt_class_scatter_f32": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "htrvt_sumsq_blocks": (i32, [i64]),
    "htrvt_sumsq": (i32, [vp, i64, vp, vp, vp]),
    "htrvt_sam_first_step": (i32, [vp, vp, vp, i64, f32, vp, vp]),
    "htrvt_sam_restore": (i32, [vp, vp, i64, vp]),
    "htrvt_ema_update": (i32, [vp, i32, i64, f64, vp]),
    "htrvt_ctc_greedy_decode": (i32, [vp, i32, i32, i32, i64, i32, vp, vp, vp]),
    "htrvt_adamw": (i32, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, i32, vp]),
    "htrvt_adamw_scalars": (i32, [f64, f64, f64, f64, f64, i32, vp]),
    "htrvt_adamw_dev": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "htrvt_line_max_scale": (i32, []),
    "htrvt_line_prepare": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "htrvt_ctc_workspace_floats": (C.c_size_t, [i32, i32, i32]),
    "htrvt_ctc_loss": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the gfx950 kernels first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C htr-vt_amd/csrc). "
            "There is no CPU / eager fallback for the HTR-VT hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"{what or 'htrvt'} failed ({rc}): {lib.htrvt_last_error().decode()}")
