"""ctypes binding of libgamer_hip.so (the C ABI declared in include/gamer_hip.h).

There is deliberately no fallback: if the library is missing or an entry point fails, a
RuntimeError is raised.  The CPU oracle under ``oracle/`` is test infrastructure and is never
imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgamer_hip.so")
if os.environ.get("GAMER_LIB_PATH"):            # A/B runs of another build of the same library (tools/)
    LIB_PATH = os.path.abspath(os.environ["GAMER_LIB_PATH"])
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gamer_hip.h")

_lib: Optional[C.CDLL] = None

c_void_p, c_int, c_float, c_int64, c_uint64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64


class GemmDesc(C.Structure):
    """Mirror of gamer_gemm_desc."""
    _fields_ = [
        ("A", c_void_p), ("a_rs", c_int64), ("a_ks", c_int64),
        ("B", c_void_p), ("b_rs", c_int64), ("b_ks", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("alpha", c_float),
        ("accumulate", c_int),
        ("groups", c_int),
        ("group_mode", c_int),
        ("group_offsets", c_void_p),
        ("strideB", c_int64), ("strideC", c_int64),
        ("kchunk", c_int),
        ("resid", c_void_p),
        ("row_map", c_void_p),
        ("p_drop", c_float),
        ("seed", c_uint64),
        ("rowdot_other", c_void_p),
        ("rowdot_out", c_void_p),
        ("rowdot_S", c_int),
        ("qk_wq", c_void_p), ("qk_wk", c_void_p),
        ("qk_eps", c_float),
        ("qk_cos", c_void_p), ("qk_sin", c_void_p),
        ("qk_bias_q", c_void_p), ("qk_bias_k", c_void_p), ("qk_bias_v", c_void_p),
        ("qk_act_idx", c_void_p),
        ("qk_pos_ids", c_void_p),
        ("qk_q_rot", c_void_p), ("qk_k_rot", c_void_p),
        ("qk_S", c_int), ("qk_nq", c_int), ("qk_nkv", c_int),
        ("b_planes", c_void_p), ("b_plane_stride", c_int64),
        ("amax_a", c_void_p), ("amax_b", c_void_p),
        ("amax_c", c_void_p), ("amax_c_col0", c_int),
        ("wgrad_ws", c_void_p), ("wgrad_ws_floats", c_int64),
        ("sw_gu", c_void_p), ("sw_ld", c_int64),
        ("group_div", c_int), ("sw_tbl", c_void_p), ("b_planes_t", c_void_p),
        ("sw_hm", c_void_p), ("sw_row_group", c_void_p),
    ]


class GemmBf16Desc(C.Structure):
    """Mirror of gamer_gemm_bf16_desc."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64),
        ("B", c_void_p), ("ldb", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("accumulate", c_int),
        ("groups", c_int),
        ("group_mode", c_int),
        ("group_offsets", c_void_p),
        ("strideB", c_int64), ("strideC", c_int64),
        ("kchunk", c_int),
        ("resid", c_void_p),
        ("row_map", c_void_p),
        ("p_drop", c_float),
        ("seed", c_uint64),
        ("rowdot_other", c_void_p),
        ("rowdot_out", c_void_p),
        ("rowdot_S", c_int),
        ("qk_wq", c_void_p), ("qk_wk", c_void_p), ("qk_eps", c_float),
        ("qk_cos", c_void_p), ("qk_sin", c_void_p),
        ("qk_bias_q", c_void_p), ("qk_bias_k", c_void_p), ("qk_bias_v", c_void_p),
        ("qk_act_idx", c_void_p), ("qk_pos_ids", c_void_p),
        ("qk_q_rot", c_void_p), ("qk_k_rot", c_void_p),
        ("qk_S", c_int), ("qk_nq", c_int), ("qk_nkv", c_int),
        ("sw_gu", c_void_p), ("sw_ld", c_int64),
        ("wgrad_ws", c_void_p), ("wgrad_ws_floats", c_int64),
    ]


P, I, F, L, U = c_void_p, c_int, c_float, c_int64, c_uint64

# name -> argtypes (everything returns int except the two noted below)
_SIGNATURES = {
    "gamer_router_fwd": [P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "gamer_expert_lists": [P, I, I, I, P, P, P, P, P],
    "gamer_embedding_fwd": [P, P, I, I, I, P, P],
    "gamer_embedding_bwd": [P, P, I, I, I, I, P, P],
    "gamer_embedding_bwd_ordered": [P, P, I, I, I, I, P, P, L, P],
    "gamer_embedding_bwd_ordered_ws_bytes": [I, I, I],
    "gamer_rmsnorm_fwd": [P, P, I, I, F, P, P, I, P],
    "gamer_rmsnorm_bwd": [P, P, P, I, P, I, I, F, I, P, P, I, P, P, F, U, P],
    "gamer_colsum_reduce": [P, I, I, I, P, P],
    "gamer_colsum_reduce_batched": [P, L, I, I, I, P, I, P],
    "gamer_rowtable_fwd": [P, P, P, I, I, P, I, I, P],
    "gamer_rowtable_bwd": [P, I, I, P, P, I, I, I, P, P, L, P],
    "gamer_gemm_f32": [C.POINTER(GemmDesc), P],
    "gamer_gemm_f32_split": [C.POINTER(GemmDesc), c_int, P],
    "gamer_split3_guard": [I],
    "gamer_gemm_bf16": [C.POINTER(GemmBf16Desc), P],
    "gamer_cast_params_bf16": [P, P, P, P, I, I, P],
    "gamer_attn_fwd_bf16": [P, I, P, I, P, I, P, P, I, I, I, I, F, F, U, P, P, P, P, P, P, P],
    "gamer_attn_bwd_bf16": [P, I, P, I, P, I, P, P, P, P, P, I, I, I, I, F, F, U, P, P, I, P, I, P, I, P, I, P, P, P, P],
    "gamer_session_spans": [P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "gamer_qknorm_rope_fwd": [P, I, I, I, I, P, P, F, P, P, P, P, P, P, P, P, P, P],
    "gamer_qknorm_rope_bwd": [P, P, P, I, I, I, I, P, P, F, P, P, P, P, P, I, P, P, P, P, P, P, P, P, L, P],
    "gamer_attn_row_order": [P, I, I, P, P, P, P],
    "gamer_attn_fwd": [P, I, P, I, P, I, P, P, P, P, I, I, I, I, F, F, U, P, P, P, P, P, I, P, P],
    "gamer_attn_bwd": [P, I, P, I, P, I, P, P, P, P, P, P, P, I, I, I, I, F, F, U, P, P, I, P, I, P, I, P, P, P, P, P, I, P],
    "gamer_split3_planes": [P, P, L, L, P],
    "gamer_absmax_f32": [P, I, L, I, I, L, P, P],
    "gamer_absmax_multi_f32": [P, P, I, P, P],
    "gamer_split2h_planes_multi": [P, P, I, P, P, P],
    "gamer_amax_sink": [P, P],
    "gamer_amax_sink3": [P, P, P],
    "gamer_attn_split_amax": [P, P, P, P],
    "gamer_attn_fwd_split": [P, I, P, I, P, I, P, P, P, I, I, I, I, F, F, U, P, P, P, P, P, I, P, P],
    "gamer_attn_bwd_split": [P, I, P, I, P, I, P, P, P, P, P, P, P, I, I, I, I, F, F, U, P, P, I, P, I, P, I, P, P, P, I, P, P, P],
    "gamer_residual_dropout_fwd": [P, P, P, I, I, F, U, P, P],
    "gamer_residual_dropout_bwd": [P, P, I, I, F, U, P, P],
    "gamer_swiglu_fwd": [P, P, L, F, U, P, P],
    "gamer_swiglu_bwd": [P, P, P, L, F, U, P],
    "gamer_swiglu_fwd_ld": [P, L, I, I, F, U, P, P],
    "gamer_swiglu_bwd_ld": [P, L, I, I, P, F, U, P],
    "gamer_swiglu_fwd_ld_tbl": [P, L, I, I, F, U, P, P, P, P],
    "gamer_swiglu_bwd_ld_tbl": [P, L, I, I, P, F, U, P, P, P],
    "gamer_inject_table_fwd": [P, P, L, I, I, I, I, I, P, P],
    "gamer_segment_colsum_ws_floats": [I, I, I],
    "gamer_segment_colsum": [P, L, I, I, P, I, P, L, P, P],
    "gamer_inject_table_bwd": [P, P, P, L, I, I, I, I, I, P, P, P, P],
    "gamer_silu_gate_fwd": [P, P, L, P, P, F, U, P],
    "gamer_silu_gate_bwd": [P, P, P, L, P, P, F, U, P],
    "gamer_check_labels": [P, L, I, I, P, P],
    "gamer_ce_fwd": [P, I, P, I, I, I, F, I, P, P, P, P, P],
    "gamer_ce_bwd": [P, I, P, I, I, I, F, I, P, P, F, F, P, P],
    "gamer_sumsq": [P, L, P, I, P],
    "gamer_adamw": [P, P, P, P, L, L, F, F, F, F, F, I, F, F, P, I, P, P],
    "gamer_fill_f32": [P, L, F, P],
    "gamer_bias_act_fwd": [P, P, I, I, I, P, P],
    "gamer_bias_act_bwd": [P, P, I, I, I, P, P, I, P],
    "gamer_layernorm_fwd": [P, P, P, P, I, I, F, P, P, P, P, P],
    "gamer_layernorm_bwd": [P, P, P, P, P, I, I, P, P, P, I, P],
    "gamer_attn_dense_fwd": [P, I, P, I, P, I, P, P, I, I, I, I, F, F, U, P, I, P, P],
    "gamer_attn_dense_bwd": [P, I, P, I, P, I, P, P, I, I, I, I, F, F, U, P, P, I, P, P, I, P, I, P, I, P],
    "gamer_reload_env": [],
    "gamer_split2h_transpose_multi": [P, P, I, P, P],
    "gamer_trie_logprobs": [P, L, P, P, P, P, P, I, I, P, P],
    "gamer_trie_advance": [P, P, P, P, P, I, P, P],
    "gamer_kv_append": [P, I, P, I, P, P, I, I, I, I, I, P],
    "gamer_attn_decode": [P, I, P, I, P, I, P, P, P, I, I, I, I, P, I, I, I, I, I, F, P, P],
    "gamer_attn_decode_split": [P, I, P, I, P, I, P, P, P, I, I, I, I, P, I, I, I, I, I, F, P, P, P, P],
}


# bf16 twins: same argument kinds as the fp32 entry point (activation pointers are gamer_bf16* instead of float*)
for _n in ("rmsnorm_fwd", "rmsnorm_bwd", "rowtable_fwd", "rowtable_bwd", "qknorm_rope_fwd", "qknorm_rope_bwd", "swiglu_fwd",
           "swiglu_bwd", "swiglu_fwd_ld", "swiglu_bwd_ld", "silu_gate_fwd", "silu_gate_bwd", "ce_fwd", "ce_bwd"):
    _SIGNATURES[f"gamer_{_n}_bf16"] = _SIGNATURES[f"gamer_{_n}"]


def header_symbols():
    """Names of every entry point declared in include/gamer_hip.h."""
    txt = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"\b(gamer_[a-z0-9_]+)\s*\(", txt)) - {"gamer_gemm_desc"})


def load(build_if_missing: bool = False) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build as _build
            _build.build()
        else:
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m gamer_amd.build` "
                "(gamer_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.gamer_abi_version.restype = c_int
    lib.gamer_abi_version.argtypes = []
    lib.gamer_last_error.restype = C.c_char_p
    lib.gamer_last_error.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = args
    lib.gamer_embedding_bwd_ordered_ws_bytes.restype = c_int64      # (a size, not an error code: call it on the library object)
    lib.gamer_segment_colsum_ws_floats.restype = c_int64
    _lib = lib
    return lib


def check(rc: int, name: str):
    if rc != 0:
        msg = load().gamer_last_error().decode(errors="replace")
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (so kernels order with torch ops on that stream)."""
    return torch.cuda.current_stream().cuda_stream


# Called (with the entry point's name) when an entry point returns non-zero, before the RuntimeError is raised: host-side state
# that was prepared for "the next launch" (ops: maxima slots handed to a producer that never ran) is dropped here.
FAILURE_HOOKS = []


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        for hook in FAILURE_HOOKS:
            hook(name)
    check(rc, name)
