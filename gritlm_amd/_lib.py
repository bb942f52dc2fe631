"""ctypes binding of libgritlm_hip.so (include/gritlm_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises.
The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRIT_HIP_LIB") or os.path.join(_HERE, "libgritlm_hip.so")      # GRIT_HIP_LIB: A/B builds (tools/ubench)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gritlm_hip.h")

ABI_VERSION = 5
GRIT_OK, GRIT_E_BADARG, GRIT_E_UNSUPPORTED, GRIT_E_LAUNCH, GRIT_E_RCCL = 0, -1, -2, -3, -4
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU, EPI_ROPE, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE, EPI_SWIGLU_BWD, EPI_RESIDUAL_F32 = 0, 1, 2, 3, 4, 5, 6, 7
POOL_MODES = {"mean": 0, "weightedmean": 1, "cls": 2, "lasttoken": 3}

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

_SIGNATURES = {
    "grit_version": (C.c_int, []),
    "grit_last_error_string": (C.c_char_p, []),
    "grit_embed_gather": (_i, [_p, _p, _p, _l, _i, _l, _p]),
    "grit_rmsnorm_fwd": (_i, [_p, _p, _p, _l, _i, _f, _p]),
    "grit_embed_gather_f32": (_i, [_p, _p, _p, _l, _i, _l, _p]),
    "grit_rmsnorm_fwd_f32in": (_i, [_p, _p, _p, _l, _i, _f, _p]),
    "grit_rmsnorm_fwd_f32in_f16": (_i, [_p, _p, _p, _l, _i, _f, _p]),
    "grit_rmsnorm_fwd_f16in": (_i, [_p, _p, _p, _i, _l, _i, _f, _p]),
    "grit_gemm_f16_nt": (_i, [_p, _p, _p, _l, _i, _i, _l, _l, _l, _i, _p, _l, _p]),
    "grit_gemm_f16_nt_rope": (_i, [_p, _p, _p, _l, _i, _i, _l, _l, _l, _p, _p, _p, _i, _i, _i, _p]),
    "grit_attn_bidir_f16_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_bidir_varlen_f16_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_causal_f16_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_attn_causal_varlen_f16_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_f16_overflow_flag": (_i, [C.POINTER(C.c_int), _i, _p]),
    "grit_rope_qk_inplace": (_i, [_p, _p, _p, _l, _i, _i, _i, _i, _l, _i, _p]),
    "grit_rope_qk_inplace_pos": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _l, _i, _p]),
    "grit_attn_bidir_varlen_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_causal_varlen_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_pool_norm_varlen_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "grit_gemm_bf16_nt": (_i, [_p, _p, _p, _l, _i, _i, _l, _l, _l, _i, _p, _l, _p]),
    "grit_gemm_bf16_nt_pair": (_i, [_p, _p, _p, _p, _l, _i, _l, _l, _l, _l, _p, _p, _p, _p, _l, _i, _l, _l, _l, _l, _i, _i, _p]),
    "grit_swiglu_block": (_i, []),
    "grit_mask_pack": (_i, [_p, _p, _i, _i, _p]),
    "grit_attn_bidir_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_causal_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_causal_window_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_attn_causal_window_varlen_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_attn_causal_window_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_attn_causal_window_varlen_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_pool_norm_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "grit_pool_norm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "grit_gemm_bf16_nt_rope": (_i, [_p, _p, _p, _l, _i, _i, _l, _l, _l, _p, _p, _p, _i, _i, _i, _p]),
    "grit_gemm_bf16_nt_grouped": (_i, [_p, _p, _p, _p, _p, _i, _l, _i, _i, _l, _l, _l, _l, _i, _p]),
    "grit_gemm_f16_nt_grouped": (_i, [_p, _p, _p, _p, _p, _i, _l, _i, _i, _l, _l, _l, _l, _i, _p]),
    "grit_moe_router_top2_f32": (_i, [_p, _p, _f, _p, _p, _p, _l, _i, _i, _p]),
    "grit_moe_combine_f32": (_i, [_p, _p, _p, _p, _p, _l, _i, _p]),
    "grit_gemv_bf16": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _l, _p]),
    "grit_rmsnorm_gemv_bf16": (_i, [_p, _p, _f, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p]),
    "grit_rmsnorm_gemv_bf16_deferred": (_i, [_p, _p, _f, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p]),
    "grit_rope_kv_append": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _p]),
    "grit_kv_append": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _p]),
    "grit_attn_decode_workspace_floats": (_l, [_i, _i, _i, _i]),
    "grit_attn_decode": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_decode_rope": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_argmax_advance": (_i, [_p, _l, _i, _p, _p, _p, _l, _p, _i, _p]),
    "grit_gemv_f16": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _l, _p, _l, _p]),
    "grit_rmsnorm_gemv_f16_deferred": (_i, [_p, _p, _f, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p]),
    "grit_attn_decode_rope_f16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_argmax_advance_f32": (_i, [_p, _l, _i, _p, _p, _p, _l, _p, _i, _p]),
    "grit_gemv_bf16_expert": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _l, _l, _l, _i, _p]),
    "grit_gemv_f16_expert": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _l, _l, _l, _i, _p]),
    "grit_moe_decode_combine_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "grit_rope_kv_append_rows": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _i, _p]),
    "grit_attn_decode_rows": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "grit_knn_workspace_bytes": (_l, [_i, _l, _i]),
    "grit_knn_topk": (_i, [_p, _p, _i, _l, _i, _l, _l, _i, _p, _p, _p, _p]),
    "grit_ce_fwd": (_i, [_p, _l, _p, _p, _p, _l, _i, _p]),
    "grit_ce_bwd": (_i, [_p, _l, _p, _p, _p, _f, _l, _i, _p]),
    "grit_moe_router_top2": (_i, [_p, _p, _p, _p, _l, _i, _i, _p]),
    "grit_moe_index_workspace_ints": (_l, [_l, _i]),
    "grit_moe_index": (_i, [_p, _l, _i, _p, _p, _p, _p, _p]),
    "grit_moe_combine": (_i, [_p, _p, _p, _p, _p, _l, _i, _p]),
    "grit_moe_combine_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _p]),
    "grit_moe_router_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _p]),
    "grit_moe_router_wgrad_workspace_floats": (_l, [_l, _i, _i]),
    "grit_moe_router_wgrad": (_i, [_p, _p, _p, _p, _l, _i, _i, _p]),
    "grit_gemm_bf16_nt_grouped_epi": (_i, [_p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _l, _l, _l, _l, _l, _i, _p]),
    "grit_pool_norm_varlen_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "grit_infonce_rows_fwd_bwd": (_i, [_p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "grit_transpose_bf16": (_i, [_p, _p, _l, _l, _l, _l, _p]),
    "grit_rmsnorm_bwd_workspace_rows": (_l, [_l]),
    "grit_rmsnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _f, _p]),
    "grit_swiglu_fwd": (_i, [_p, _p, _l, _i, _p]),
    "grit_swiglu_bwd": (_i, [_p, _p, _p, _l, _i, _p]),
    "grit_attn_bidir_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_causal_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_bidir_varlen_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _i, _i, _i, _l, _l, _f, _p]),
    "grit_attn_causal_varlen_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _i, _i, _i, _l, _l, _f, _p]),
    "grit_embed_scatter_add_sorted": (_i, [_p, _p, _p, _p, _l, _i, _l, _p]),
    "grit_accum_bf16_from_f32": (_i, [_p, _p, _l, _p]),
    "grit_comm_unique_id": (_i, [_p]),
    "grit_comm_init": (_i, [_p, _i, _i, C.POINTER(C.c_void_p)]),
    "grit_comm_allgather_packed": (_i, [_p, _p, _l, _p, _l, _i, _p, _p, _p]),
    "grit_comm_destroy": (_i, [_p]),
    "grit_stream_create_cu_mask": (_i, [_i, C.POINTER(C.c_void_p)]),
    "grit_stream_destroy": (_i, [_p]),
}
COMM_ID_BYTES = 128

_lib = None


class GritHipError(RuntimeError):
    pass


def header_symbols(path: str = HEADER_PATH) -> list[str]:
    """Every function declared in include/gritlm_hip.h (used by the ABI test)."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(grit_[a-z0-9_]+)\s*\(", src)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GritHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the native path.")
    # PyTorch ships its own libamdhip64; the library must bind to THAT runtime instance when both live in one process (streams and device
    # pointers cross the boundary).  Loaded first, it would resolve /opt/rocm's copy, torch would bring a second one, and the first launch
    # would fail with "no ROCm-capable device is detected" (found with build() followed by smoke() in one process): torch goes first.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is absent -> loud
        fn.restype = res
        fn.argtypes = args
    if lib.grit_version() != ABI_VERSION:
        raise GritHipError(f"ABI version mismatch: library {lib.grit_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().grit_last_error_string().decode()
        kind = {-1: "BADARG", -2: "UNSUPPORTED", -3: "LAUNCH", -4: "RCCL"}.get(rc, str(rc))
        raise GritHipError(f"{what} failed (GRIT_E_{kind}): {msg}")
