"""ctypes binding of libcldrd_hip.so (C ABI: include/cldrd_hip.h).  No CPU fallback: if the library is missing
or a launch is rejected, the call raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLDRD_LIB selects another BUILD of the library (tools/build_dev.py: the development build with the experiments' knobs); the product
# library itself reads no environment variable
LIB_PATH = os.environ.get("CLDRD_LIB") or os.path.join(_HERE, "libcldrd_hip.so")

_lib = None

vp, ci, cf, cull, csz = C.c_void_p, C.c_int, C.c_float, C.c_ulonglong, C.c_size_t

SIGNATURES = {
    "cldrd_last_error": (C.c_char_p, []),
    "cldrd_version": (ci, []),
    "cldrd_device_ok": (ci, []),
    "cldrd_set_tuning": (ci, [C.c_char_p, ci]),
    "cldrd_gemm_nt16": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, cf, cf, cull, ci, ci, ci, vp]),
    "cldrd_gemm_nt16_ln": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, cf, cf, cull, ci, ci, ci, vp, vp, vp, vp, vp]),
    "cldrd_gemm_nt_splitk_workspace": (csz, [ci, ci, ci]),
    "cldrd_gemm_nt16_ws": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, cf, cf, cull, ci, ci, ci, vp, vp, vp, vp, vp, vp, csz, vp]),
    "cldrd_wgrad_splits": (ci, [ci, ci, ci]),
    "cldrd_wgrad16": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, csz, ci, vp]),
    "cldrd_wgrad_group_workspace": (csz, [vp, vp, vp, ci]),
    "cldrd_wgrad_group": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, csz, ci, vp]),
    "cldrd_attention_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp]),
    "cldrd_attention_bwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, vp]),
    "cldrd_attention_bits_words": (C.c_longlong, [ci, ci, ci, cf]),
    "cldrd_attention_fwd_bits": (ci, [vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp, vp, vp]),
    "cldrd_attention_bwd_bits": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, vp, vp]),
    "cldrd_attention_bwd_x": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, vp, ci, vp]),
    "cldrd_attention_cls_fwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp, vp]),
    "cldrd_attention_cls_bwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, vp]),
    "cldrd_attention_cls_bwd_x": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp]),
    "cldrd_attention_fwd_varlen": (ci, [vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp, vp, vp]),
    "cldrd_attention_bwd_varlen": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, vp, ci, vp]),
    "cldrd_attention_fwd_varlen_list": (ci, [vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp, vp, vp, ci, ci, vp]),
    "cldrd_attention_bwd_varlen_list": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, vp, ci, vp, ci, ci, vp]),
    "cldrd_attention_cls_fwd_varlen": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp, vp]),
    "cldrd_attention_cls_bwd_varlen": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cull, ci, vp]),
    "cldrd_add_rows_strided": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "cldrd_ln_partial_blocks": (ci, [ci]),
    "cldrd_embed_ln_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, cull, vp, ci, vp, vp, vp]),
    "cldrd_embed_ln_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cull, ci, vp, ci, vp, vp]),
    "cldrd_layernorm_fwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, cf, vp, ci, ci, vp, ci, vp, vp]),
    "cldrd_layernorm_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, cull, ci, ci, vp, vp]),
    "cldrd_ln_reduce_group": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "cldrd_colsum_bf16": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "cldrd_scatter_cls_grad": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "cldrd_score_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "cldrd_score_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "cldrd_loss_fwd_bwd": (ci, [ci, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf, ci, vp]),
    "cldrd_logit_norm_reg": (ci, [vp, ci, cf, vp, vp, vp, vp]),
    "cldrd_lambda_loss_fwd_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, cf, cf, ci, ci, ci, vp]),
    "cldrd_sqnorm_blocks": (ci, []),
    "cldrd_grad_clip_coef": (ci, [vp, csz, cf, vp, vp, vp]),
    "cldrd_sqnorm_partial": (ci, [vp, csz, vp, ci, vp]),
    "cldrd_clip_coef": (ci, [vp, ci, cf, vp, vp]),
    "cldrd_copy_segments": (ci, [vp, vp, vp, ci, vp]),
    "cldrd_zero_segments": (ci, [vp, vp, ci, vp]),
    "cldrd_adamw_step": (ci, [vp, vp, vp, vp, vp, vp, csz, cf, cf, cf, cf, cf, ci, vp, vp]),
    "cldrd_adamw_step_h16": (ci, [vp, vp, vp, vp, vp, vp, csz, cf, cf, cf, cf, cf, ci, vp, vp, csz, csz, vp]),
    "cldrd_cast_bf16": (ci, [vp, vp, csz, vp]),
    "cldrd_transpose_cast_batched": (ci, [vp, vp, vp, vp, ci, ci, vp]),
    "cldrd_transpose_bf16_batched": (ci, [vp, vp, vp, vp, ci, ci, vp]),
    "cldrd_topk_scan_filter": (ci, [vp, vp, ci, C.c_longlong, ci, vp, vp, vp, vp, ci, ci, vp]),
    "cldrd_topk_scan_filter_tiled": (ci, [vp, vp, ci, C.c_longlong, ci, vp, vp, vp, vp, ci, ci, vp]),
    "cldrd_cast_f16": (ci, [vp, vp, csz, vp, vp]),
    "cldrd_topk_prep_queries": (ci, [vp, vp, vp, vp, ci, ci, vp, vp]),
    "cldrd_topk_thresholds": (ci, [vp, vp, cf, ci, vp, vp, ci, vp]),
    "cldrd_topk_select": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp]),
    "cldrd_flatip_search": (ci, [vp, vp, vp, vp, vp, vp, C.c_longlong, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp]),
    "cldrd_topk_kth_largest": (ci, [vp, ci, ci, ci, ci, vp, vp]),
    "cldrd_topk_rescore": (ci, [vp, vp, ci, vp, vp, vp, ci, ci, vp]),
    "cldrd_topk_sort": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    "cldrd_row_sqnorm_max": (ci, [vp, csz, ci, vp, vp]),
    "cldrd_gather_cast_rows": (ci, [vp, vp, csz, csz, ci, vp]),
    "cldrd_index_col_mean_workspace": (csz, [csz, ci]),
    "cldrd_index_col_mean": (ci, [vp, csz, ci, vp, vp, csz, vp]),
    "cldrd_index_center_cast": (ci, [vp, vp, csz, ci, vp, vp, csz, csz, vp, vp, vp]),
    "cldrd_map_ids": (ci, [vp, vp, C.c_longlong, vp, csz, vp]),
    "cldrd_unpack_rows16": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "cldrd_gather_rows": (ci, [vp, vp, vp, ci, ci, vp]),
    "cldrd_gather_i64": (ci, [vp, vp, vp, ci, vp]),
    "cldrd_scatter_cls_grad_idx": (ci, [vp, vp, ci, ci, vp, ci, ci, vp]),
    "cldrd_add_rows_idx": (ci, [vp, vp, ci, ci, vp, ci, vp]),
    "cldrd_set_seed_base": (None, [vp]),
    "cldrd_set_optim_hyper": (None, [vp]),
    "cldrd_set_loss_scale": (None, [vp, ci]),
    "cldrd_set_norm_sink": (None, [vp, ci]),
    "cldrd_norm_sink_used": (ci, []),
    "cldrd_loss_scale_adapt": (ci, [vp, csz, vp, csz, vp, vp]),
    "cldrd_write_step_state": (ci, [vp, cull, cull, vp, cf, cf, cf, ci, vp, vp]),
    "cldrd_write_run_file": (C.c_longlong, [C.c_char_p, vp, vp, vp, C.c_longlong, ci, ci]),
    "cldrd_py_float_repr": (ci, [C.c_double, C.c_char_p]),
    "cldrd_merge_topk": (ci, [vp, vp, ci, C.c_longlong, ci, ci, vp, vp, ci]),
    "cldrd_merge_topk_device_workspace": (csz, [ci, C.c_longlong, ci, ci]),
    "cldrd_merge_topk_device": (ci, [vp, vp, ci, C.c_longlong, ci, ci, vp, vp, vp, csz, vp]),
}


class CldrdError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own HIP runtime (libamdhip64); loading ours before it would put a second runtime
    # in the process and the launches would see no device
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CldrdError(f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                         "cldrd_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name: str, *args):
    """Invoke an int-returning entry point; raise with cldrd_last_error() on rejection."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise CldrdError(f"{name}: {lib.cldrd_last_error().decode()}")
