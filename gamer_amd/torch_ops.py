"""torch.ops.gamer.*: the hot-path kernels registered with the PyTorch dispatcher (TORCH_LIBRARY in csrc/torch_ops.cpp
over the C ABI of libgamer_hip.so) - the registration SURVEY.md section 8(b) names for the drop-in boundary.

``load()`` makes ``torch.ops.gamer.rmsnorm_fwd`` / ``rmsnorm_bwd`` / ``linear`` / ``qkv_rope_fwd`` /
``mb_attention_fwd`` / ``mb_attention_bwd`` / ``swiglu_fwd`` / ``swiglu_bwd`` / ``lmhead_ce_fwd`` / ``lmhead_ce_bwd`` /
``fused_adamw_clip`` available (fp32 or bf16 activations by tensor dtype).  There is no CPU implementation: calling an
op with CPU tensors raises the dispatcher's NotImplementedError.
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgamer_torch.so")
OPS = ("rmsnorm_fwd", "rmsnorm_bwd", "linear", "qkv_rope_fwd", "mb_attention_fwd", "mb_attention_bwd", "swiglu_fwd",
       "swiglu_bwd", "lmhead_ce_fwd", "lmhead_ce_bwd", "fused_adamw_clip")
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m gamer_amd.build` (no CPU fallback)")
        from . import _lib
        _lib.load()                          # libgamer_hip.so first (same directory; the op library links against it)
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.gamer
