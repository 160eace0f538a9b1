"""torch.ops.gamer.*: the hot-path kernels registered with the PyTorch dispatcher (TORCH_LIBRARY in csrc/torch_ops.cpp
over the C ABI of libgamer_hip.so) - the registration SURVEY.md section 8(b) names for the drop-in boundary.

``load()`` makes ``torch.ops.gamer.rmsnorm_fwd`` / ``rmsnorm_bwd`` / ``linear`` / ``qkv_rope_fwd`` /
``mb_attention_fwd`` / ``mb_attention_bwd`` / ``swiglu_fwd`` / ``swiglu_bwd`` / ``routed_swiglu_fwd`` / ``routed_swiglu_bwd`` /
``lmhead_ce_fwd`` / ``lmhead_ce_bwd`` / ``fused_adamw_clip`` / ``allreduce_bucket`` available (fp32 or bf16 activations by tensor dtype).  There is no CPU implementation: calling an
op with CPU tensors raises the dispatcher's NotImplementedError.
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgamer_torch.so")
OPS = ("rmsnorm_fwd", "rmsnorm_bwd", "linear", "qkv_rope_fwd", "mb_attention_fwd", "mb_attention_bwd", "swiglu_fwd",
       "swiglu_bwd", "routed_swiglu_fwd", "routed_swiglu_bwd", "lmhead_ce_fwd", "lmhead_ce_bwd", "fused_adamw_clip",
       "allreduce_bucket")
_fragment = None
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m gamer_amd.build` (no CPU fallback)")
        from . import _lib
        _lib.load()                          # libgamer_hip.so first (same directory; the op library links against it)
        torch.ops.load_library(LIB_PATH)
        _register_allreduce_bucket()
        _loaded = True
    return torch.ops.gamer


def _register_allreduce_bucket():
    """gamer::allreduce_bucket(flat, start, end): in-place SUM of flat[start:end] over the default process group - one
    per-layer gradient bucket of gamer_amd.dp.GradAllReducer as a dispatcher op.  The collective is
    torch.distributed's (backend "nccl" = RCCL over xGMI), so the op is defined from Python on the C++ library's namespace;
    without an initialised group it is the identity (one rank)."""
    global _fragment
    import torch.distributed as dist
    _fragment = torch.library.Library("gamer", "FRAGMENT")
    _fragment.define("allreduce_bucket(Tensor(a!) flat, int start, int end) -> ()")

    def impl(flat, start, end):
        if flat.dim() != 1 or not (0 <= start <= end <= flat.numel()):
            raise RuntimeError(f"allreduce_bucket: flat must be 1-d and 0 <= start <= end <= numel, got {tuple(flat.shape)} [{start}, {end})")
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(flat[start:end], op=dist.ReduceOp.SUM)

    _fragment.impl("allreduce_bucket", impl, "CompositeExplicitAutograd")
