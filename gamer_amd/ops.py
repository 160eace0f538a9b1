"""Thin tensor-level wrappers over the C ABI (one function per entry point).

All tensors must live on the HIP device, fp32 / int32 / int64 as documented in
include/gamer_hip.h.  Kernels are enqueued on torch's current stream.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import torch

from . import _lib
from ._lib import GemmBf16Desc, GemmDesc, call, ptr, stream_ptr


def _sfx(t: torch.Tensor) -> str:
    """Entry-point suffix for the activation dtype of ``t``: fp32 -> "", bf16 -> "_bf16" (the AMP variant)."""
    if t.dtype == torch.float32:
        return ""
    if t.dtype == torch.bfloat16:
        return "_bf16"
    raise RuntimeError(f"activations must be float32 or bfloat16, got {t.dtype}")


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on the HIP device (gamer_amd has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")


# ----------------------------------------------------------------------------------------------
def reload_env():
    """The library caches its GAMER_* kernel switches (one read per process): call this after changing os.environ in-process."""
    call("gamer_reload_env")


class env_switches:
    """`with env_switches(GAMER_GEMM_AS=0): ...` - set GAMER_* kernel switches for a block inside a running process (tests, A/B
    tools) and have the library read them again on entry and exit."""

    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)
        reload_env()
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_env()


def router_fwd(ids, attn_mask, actions, behavior_lut, num_positions, pad_id, eos_id, out: dict):
    B, S = ids.shape
    _chk(ids, torch.int64, "input_ids")
    call("gamer_router_fwd", ptr(ids), ptr(attn_mask), ptr(actions), ptr(behavior_lut), behavior_lut.numel(),
         B, S, num_positions, pad_id, eos_id,
         ptr(out["expert"]), ptr(out["beh_idx"]), ptr(out["act_idx"]),
         ptr(out["kl_self"]), ptr(out["kl_cross"]), ptr(out["ql_cross"]),
         ptr(out["empty_self"]), ptr(out["empty_cross"]),
         ptr(out["tile_empty_self"]), ptr(out["tile_empty_cross"]), ptr(out["bad_token"]), stream_ptr())


def alloc_router_outputs(B, S, device):
    n_tiles = (S + 31) // 32
    i32 = dict(dtype=torch.int32, device=device)
    out = {k: torch.empty(B, S, **i32) for k in
           ("expert", "beh_idx", "act_idx", "kl_self", "kl_cross", "ql_cross", "empty_self", "empty_cross")}
    out["tile_empty_self"] = torch.empty(B, n_tiles, **i32)
    out["tile_empty_cross"] = torch.empty(B, n_tiles, **i32)
    out["bad_token"] = torch.zeros(1, **i32)
    return out


def alloc_session_outputs(B, S, device):
    i32 = dict(dtype=torch.int32, device=device)
    return {"span_self": torch.empty(B, S, 4, **i32), "span_cross": torch.empty(B, S, 4, **i32),
            "pos_ids": torch.empty(B, S, **i32), "violations": torch.zeros(1, **i32)}


def session_spans(session_ids, extended_session_ids, attn_mask, num_positions, n_rope_positions, router: dict,
                  out: dict):
    """Qwen3SessionMulti masks as per-query key spans (gamer_session_spans).  Run after router_fwd: overwrites the
    router's empty_* / tile_empty_* with the session masks' empty rows."""
    B, S = session_ids.shape
    _chk(session_ids, torch.int64, "session_ids")
    if extended_session_ids is not None:
        _chk(extended_session_ids, torch.int64, "extended_session_ids")
    call("gamer_session_spans", ptr(session_ids), ptr(extended_session_ids), ptr(attn_mask), ptr(router["kl_cross"]),
         ptr(router["ql_cross"]), B, S, num_positions, n_rope_positions, ptr(out["span_self"]),
         ptr(out["span_cross"]), ptr(out["pos_ids"]), ptr(router["empty_self"]), ptr(router["empty_cross"]),
         ptr(router["tile_empty_self"]), ptr(router["tile_empty_cross"]), ptr(out["violations"]), stream_ptr())


def expert_lists(expert, num_experts, perm, slot, offsets, work):
    B, S = expert.shape
    call("gamer_expert_lists", ptr(expert), B, S, num_experts, ptr(perm), ptr(slot), ptr(offsets), ptr(work),
         stream_ptr())


def embedding_fwd(ids, W, x):
    T = ids.numel()
    V, H = W.shape
    call("gamer_embedding_fwd", ptr(ids), ptr(W), V, T, H, ptr(x), stream_ptr())


def embedding_bwd(ids, dx, pad_id, dW):
    T = ids.numel()
    V, H = dW.shape
    call("gamer_embedding_bwd", ptr(ids), ptr(dx), V, T, H, pad_id, ptr(dW), stream_ptr())


_EMB_WS = {}


def embedding_bwd_ordered(ids, dx, pad_id, dW):
    """dW[id] += dx rows, without float atomics: the same bits on every run (gamer_embedding_bwd_ordered)."""
    T = ids.numel()
    V, H = dW.shape
    need = int(_lib.load().gamer_embedding_bwd_ordered_ws_bytes(V, T, H))
    ws = _EMB_WS.get(dx.device)
    if ws is None or ws.numel() < need:
        _EMB_WS[dx.device] = None
        ws = _EMB_WS[dx.device] = torch.empty(need, dtype=torch.uint8, device=dx.device)
    call("gamer_embedding_bwd_ordered", ptr(ids), ptr(dx), V, T, H, pad_id, ptr(dW), ptr(ws), ws.numel(), stream_ptr())


def rmsnorm_fwd(x, w, eps, y, ldy=None, dst_rows=None):
    T, H = x.shape
    ld = ldy if ldy else y.stride(0)
    _arm_sink((y, (1, 0, T, ld, ld), False))       # (the consumer GEMM reads rows of ld columns: rowtable_fwd adds the rest)
    call("gamer_rmsnorm_fwd" + _sfx(y), ptr(x), ptr(w), T, H, eps, ptr(dst_rows), ptr(y), ld, stream_ptr())


def rmsnorm_bwd(x, w, dy, lddy, eps, dx, dw_partial, accumulate_dx=True, dy_rows=None, mask_out=None, mask_rows=None,
                p=0.0, seed=0):
    """mask_out: also write dropout_mask(seed) * dx (the input gradient of the next residual branch), rows scattered
    through mask_rows when given - the second pass gamer_residual_dropout_bwd would make over dx."""
    T, H = x.shape
    if mask_out is not None:
        _arm_sink((mask_out, (1, 0, T, H, H), False))
    call("gamer_rmsnorm_bwd" + _sfx(dy), ptr(x), ptr(w), ptr(dy), lddy, ptr(dy_rows), T, H, eps, 1 if accumulate_dx else 0,
         ptr(dx), ptr(dw_partial), dw_partial.shape[0], ptr(mask_out), ptr(mask_rows), p, seed, stream_ptr())


def colsum_reduce(partial, out, accumulate=False):
    rows, cols = partial.shape
    call("gamer_colsum_reduce", ptr(partial), rows, cols, 1 if accumulate else 0, ptr(out), stream_ptr())


def colsum_reduce_batched(partials, n, outs_table, accumulate=False):
    """partials [>= n, rows, cols] (contiguous); outs_table: int64 device tensor of n output addresses."""
    _, rows, cols = partials.shape
    call("gamer_colsum_reduce_batched", ptr(partials), rows * cols, rows, cols, n, ptr(outs_table), 1 if accumulate else 0,
         stream_ptr())


def rowtable_fwd(table, idx, y, ldy, col0, dst_rows=None):
    T = idx.numel()
    E = table.shape[1]
    _arm_sink((y, (1, 0, T, ldy, ldy), "only"))    # joins the slot the RMSNorm that wrote the other columns opened, if any
    call("gamer_rowtable_fwd" + _sfx(y), ptr(table), ptr(idx), ptr(dst_rows), T, E, ptr(y), ldy, col0, stream_ptr())


def rowtable_bwd(dy, lddy, col0, idx, dtable, dy_rows=None, partial=None):
    """partial: fp32 scratch (>= rows * E floats): the ordered form (no float atomics, same bits on every run)."""
    T = idx.numel()
    rows, E = dtable.shape
    call("gamer_rowtable_bwd" + _sfx(dy), ptr(dy), lddy, col0, ptr(idx), ptr(dy_rows), T, E, rows, ptr(dtable), ptr(partial),
         partial.numel() if partial is not None else 0, stream_ptr())


# fp32 matmul form in effect: 0 = v_mfma_f32_32x32x2_f32, 6 / 9 = gamer_gemm_f32_split (exact three-way bf16 cut of both
# operands, 6 or 9 piece products on the bf16 pipe), 3 = its two-way fp16 form (per-tensor power-of-two scales from
# gamer_absmax_f32, 3 piece products).  Default 0; an Engine / DecodeSession with another form scopes its
# own calls with `with ops.f32_matmul(form):`, which restores the previous value on exit - two engines with different
# forms, tools and tests in one process do not leak their setting into each other.
F32_MATMUL_TERMS = 0
MATMUL_MODES = {"f32": 0, "split3": 3, "split6": 6, "split9": 9}
# Pre-cut weights of the split forms: (address of the fp32 flat parameter buffer, its bytes, address of the three bf16 planes,
# plane stride in elements) or None.  Set - scoped, like the matmul form - by the engine that owns the buffers; gemm() hands a
# B operand that lies inside the flat buffer to gamer_gemm_f32_split together with its planes.
WEIGHT_PLANES = None


class f32_matmul:
    """Context manager: fp32 GEMMs issued inside the block use `mode` ("f32" | "split6" | "split9" or 0 / 6 / 9)."""

    def __init__(self, mode, planes=None):
        self.mode, self.planes = mode, planes

    def __enter__(self):
        global WEIGHT_PLANES
        self.prev = set_f32_matmul(self.mode)
        self.prev_planes = WEIGHT_PLANES
        WEIGHT_PLANES = self.planes
        return self

    def __exit__(self, *exc):
        global WEIGHT_PLANES
        set_f32_matmul(self.prev)
        WEIGHT_PLANES = self.prev_planes
        return False


def scoped_f32_matmul(get_mode, get_planes=None):
    """Decorator form of f32_matmul: `get_mode(*args)` names the form for the duration of the call, `get_planes(*args)`
    (optional) the pre-cut weight planes (WEIGHT_PLANES)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            with f32_matmul(get_mode(*args), get_planes(*args) if get_planes is not None else None):
                return fn(*args, **kwargs)
        return wrapper
    return deco


# split3: one device word per operand tensor for the bits of its largest magnitude.  Slots come from a ring that is zeroed as a
# whole when it wraps (stream-ordered: every GEMM that read an old slot was enqueued before the memset).
_AMAX_RING = {}
_AMAX_SLOTS = 1024
AMAX_WORDS = 256            # GAMER_AMAX_WORDS: one maximum = 16 words 64 bytes apart (producers spread their atomics over them)
# Optional reuse of a maximum for a tensor that is read by several GEMMs while it does not change (x by forward and weight
# gradient, dy by weight and input gradient, a weight by every GEMM of a step): inside `with cache:` (an amax_reuse) a slot is
# keyed by (address, extent) and measured once for tensors the OWNER declared unchanging - `stable(...)` tensors / address
# ranges until the next `reset()`, `hold(...)` tensors for the duration of a with-block.  Everything else is measured per GEMM.
_AMAX_REUSE = None


class amax_reuse:
    def __init__(self, everything=False):
        self.slots, self.pending, self.pools, self.used = {}, {}, [], 0
        self.stable_ptrs, self.stable_ranges, self.held = set(), [], {}
        # dense tensors inside the FIRST stable range (the parameter buffer) that GEMMs asked for: measured together by one
        # launch at the start of the next pass (gamer_absmax_multi_f32) instead of one launch each
        self._wkeys, self._wtable, self._wtable_n = [], None, 0
        # (a buffer of the parameter buffer's size, or None): the fp16 pieces of those tensors packed at their values' offsets,
        # rebuilt by the same pass-start launch pair; a GEMM whose B operand is exactly one of them reads them instead of cutting
        self.planes, self._plane_keys = None, set()
        # (a second buffer of the same size, or None): the packed pieces of the TRANSPOSES of registered matrices at their values'
        # offsets (register_transposed: the layers with 256 output features and a long contraction - their forward runs on the
        # output-stationary kernel, which wants W^T [K][256]); rebuilt from `planes` right after them
        self.planes_t, self._tkeys, self._ttable, self._tplane_keys = None, [], None, set()
        self.everything = everything          # tools: every tensor counts as unchanging (kernel timing)

    def __enter__(self):
        global _AMAX_REUSE
        self.prev, _AMAX_REUSE = _AMAX_REUSE, self
        return self

    def __exit__(self, *exc):
        global _AMAX_REUSE
        _AMAX_REUSE = self.prev
        return False

    def stable(self, *tensors):
        self.stable_ptrs.update(t.data_ptr() for t in tensors if t is not None)

    def stable_range(self, base, nbytes):
        self.stable_ranges.append((int(base), int(nbytes)))

    def reset(self):
        """Start of a forward pass: every cached maximum is dropped, the slot words are zeroed (one memset per pool)."""
        self.slots.clear()
        self.pending.clear()
        self.held.clear()
        self.stable_ptrs.clear()        # (the owner declares its unchanging tensors again: addresses of a freed workspace may be reused)
        for pool in self.pools:
            pool.zero_()
        self.used = 0
        if self._wkeys and self.stable_ranges and len(self._wkeys) <= 1024:
            base = self.stable_ranges[0][0]
            dev = self._wdev
            if self._wtable is None or self._wtable_n != len(self._wkeys):
                tab = [v for (p_, n_) in self._wkeys for v in ((p_ - base) // 4, n_)]
                self._wtable = torch.tensor(tab, dtype=torch.int64, device=dev)
                self._wtable_n = len(self._wkeys)
            first = self._new_slot(dev)
            for _ in range(len(self._wkeys) - 1):
                self._new_slot(dev)                 # (consecutive: a fresh pool holds 1024 slots)
            call("gamer_absmax_multi_f32", base, self._wtable.data_ptr(), len(self._wkeys), first, stream_ptr())
            for e, (p_, n_) in enumerate(self._wkeys):
                self.slots[(p_, "dense", n_)] = first + 4 * AMAX_WORDS * e
            self._plane_keys = set()
            if self.planes is not None:
                call("gamer_split2h_planes_multi", base, self._wtable.data_ptr(), len(self._wkeys), first, self.planes.data_ptr(),
                     stream_ptr())
                self._plane_keys = {(p_, "dense", n_) for (p_, n_) in self._wkeys}
                self._tplane_keys = set()
                if self.planes_t is not None and self._tkeys:
                    # every registered matrix whose tensor has pieces in this pass: one launch transposes them all
                    live = [(p_, b_, r_, c_) for (p_, b_, r_, c_) in self._tkeys if (p_, "dense", b_ * r_ * c_) in self._plane_keys]
                    if live:
                        if self._ttable is None or self._ttable[0] != live:
                            tab = [v for (p_, b_, r_, c_) in live for i in range(b_) for v in ((p_ - base) // 4 + i * r_ * c_, r_, c_)]
                            self._ttable = (live, torch.tensor(tab, dtype=torch.int64, device=dev), len(tab) // 3)
                        call("gamer_split2h_transpose_multi", self.planes.data_ptr(), self._ttable[1].data_ptr(), self._ttable[2],
                             self.planes_t.data_ptr(), stream_ptr())
                        self._tplane_keys = {(p_, "dense", b_ * r_ * c_) for (p_, b_, r_, c_) in live}

    def register(self, *tensors):
        """Dense parameter tensors (inside the first stable range) that GEMMs only ever read VIEWS of (the injecting layers'
        gate|up weight: its 256 hidden columns): measured and cut with the others from the next pass on (plane_parent)."""
        if not self.stable_ranges:
            return
        b0, n0 = self.stable_ranges[0]
        for t in tensors:
            p, n = t.data_ptr(), t.numel()
            if (b0 <= p and p + 4 * n <= b0 + n0 and n % 4 == 0 and t.is_contiguous() and
                    not any(p_ < p + 4 * n and p < p_ + 4 * n_ for (p_, n_) in self._wkeys)):
                self._wkeys.append((p, n))
                self._wdev = t.device

    def reserve(self, n):
        """Make sure `n` more slots exist without another pool allocation (a pool is zero-filled when it is allocated: a
        hipGraph capture must not contain that fill - replayed, it would wipe the maxima of everything measured before)."""
        while len(self.pools) * 1024 - self.used < n and self.pools:
            self.pools.append(torch.zeros(1024 * AMAX_WORDS, dtype=torch.int32, device=self.pools[0].device))

    def register_transposed(self, t, batch, rows, cols):
        """The dense parameter tensor `t` = `batch` row-major [rows][cols] matrices: from the next pass on the pieces of their
        transposes are kept beside the pieces of `t` (planes_t; see __init__)."""
        if self.planes_t is None or not t.is_contiguous() or t.numel() != batch * rows * cols or rows % 4 or cols % 4:
            return
        key = (t.data_ptr(), int(batch), int(rows), int(cols))
        if key not in self._tkeys:
            self._tkeys.append(key)
            self.register(t)

    def plane_parent(self, p, geom):
        """Slot of the maximum of the dense parameter tensor with piece planes that contains the operand view (p, geom) - or None
        (also when the view IS such a tensor: the exact-key path handles that)."""
        key = self._key(p, geom)
        if key in self._plane_keys or len(key) == 3:
            return None
        batch, stride, rows, cols, ld = geom
        end = p + 4 * ((batch - 1) * stride + (rows - 1) * ld + cols)
        for (p_, kind, n_) in self._plane_keys:
            if p_ <= p and end <= p_ + 4 * n_:
                return self.slots.get((p_, kind, n_))
        return None

    def hold(self, *tensors):
        cache = self

        class _Hold:
            def __enter__(self_h):
                for t in tensors:
                    cache.held[t.data_ptr()] = cache.held.get(t.data_ptr(), 0) + 1

            def __exit__(self_h, *exc):
                for t in tensors:
                    p = t.data_ptr()
                    cache.held[p] -= 1
                    if cache.held[p] == 0:
                        del cache.held[p]
                        for k in [k for k in cache.slots if k[0] == p]:
                            del cache.slots[k]
                return False
        return _Hold()

    def _cached(self, p):
        if self.everything or p in self.stable_ptrs or p in self.held:
            return True
        return any(b <= p < b + n for b, n in self.stable_ranges)

    @staticmethod
    def _key(p, geom):
        batch, stride, rows, cols, ld = geom
        if ld == cols and (batch == 1 or stride == rows * ld):
            return (p, "dense", batch * rows * cols)
        return (p,) + tuple(geom)

    def _new_slot(self, device):
        i, j = divmod(self.used, 1024)
        if i == len(self.pools):
            self.pools.append(torch.zeros(1024 * AMAX_WORDS, dtype=torch.int32, device=device))
        self.used += 1
        return self.pools[i].data_ptr() + 4 * AMAX_WORDS * j

    def preset(self, x, geom, accumulate=False):
        """A slot the PRODUCER of x is about to fill (gamer_amax_sink): the next GEMM that reads x with this extent takes
        it instead of measuring x.  accumulate: a second producer writing other columns of the same tensor."""
        key = self._key(x.data_ptr(), geom)
        if accumulate and key in self.pending:
            return self.pending[key]
        if accumulate == "only":                    # a later producer of a tensor whose first producer did not open a slot
            return None
        self.slots.pop(key, None)
        ptr_ = self.pending[key] = self._new_slot(x.device)
        return ptr_

    def peek(self, x, geom):
        """The slot `slot(x, geom)` would return without measuring - its producer's, or a cached one - or None; nothing is consumed
        (a generation's prompt pass notes the maxima of the keys / values it caches before the layer's attention takes them)."""
        key = self._key(x.data_ptr(), geom)
        return self.pending.get(key) or self.slots.get(key)

    def slot(self, x, geom):
        p = x.data_ptr()
        key = self._key(p, geom)
        keep = self._cached(p)
        if key in self.pending:                     # written by its producer together with the tensor
            ptr_ = self.pending.pop(key)
            if keep:
                self.slots[key] = ptr_
            return ptr_
        if keep and key in self.slots:
            return self.slots[key]
        ptr_ = self._new_slot(x.device)
        call("gamer_absmax_f32", ptr(x), *geom, ptr_, stream_ptr())
        if keep:
            self.slots[key] = ptr_
            if (len(key) == 3 and self.stable_ranges and key[2] % 4 == 0 and
                    self.stable_ranges[0][0] <= p < self.stable_ranges[0][0] + self.stable_ranges[0][1] and
                    not any(p_ < p + 4 * key[2] and p < p_ + 4 * n_ for (p_, n_) in self._wkeys)):
                # a parameter tensor: part of the one-launch measurement from the next pass on.  Tensors that OVERLAP a registered
                # one (one expert's slice of a stacked weight in the decode step) stay per-pass measurements: their packed pieces
                # would land on the registered tensor's, cut with another scale
                self._wkeys.append((p, key[2]))
                self._wdev = x.device
        return ptr_


def _drop_pending_on_failure(name):
    """A rejected entry point never wrote the maxima slots its producers were armed with (the C side disarms the sink itself,
    set_error): forget them, or a later GEMM keyed by the same (address, extent) would scale by a slot that still holds 0."""
    if _AMAX_REUSE is not None:
        _AMAX_REUSE.pending.clear()


_lib.FAILURE_HOOKS.append(_drop_pending_on_failure)

_AMAX_ATTN = os.environ.get("GAMER_AMAX_ATTN", "1") != "0"      # the attention kernels as producers (o, dv) - A/B switch


def _arm_sink(*outs):
    """outs: (tensor, (batch, stride, rows, cols, ld), accumulate) for the first / second output of the kernel launched next.
    With a maxima cache in effect and the split3 form selected, its slots are handed to that kernel (gamer_amax_sink)."""
    if _AMAX_REUSE is None or F32_MATMUL_TERMS != 3 or outs[0][0].dtype != torch.float32:
        return
    slots = [_AMAX_REUSE.preset(t, g, acc) for t, g, acc in outs]
    if slots[0] is None:
        return
    if len(slots) > 2:
        call("gamer_amax_sink3", slots[0], slots[1], slots[2])
    else:
        call("gamer_amax_sink", slots[0], slots[1] if len(slots) > 1 else None)


def scoped_amax(get_cache):
    """Decorator: `get_cache(*args)` (an amax_reuse or None) is the maxima cache in effect for the duration of the call."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            cache = get_cache(*args)
            if cache is None:
                return fn(*args, **kwargs)
            with cache:
                return fn(*args, **kwargs)
        return wrapper
    return deco


def absmax_slot(x, batch, stride, rows, cols, ld):
    """Device pointer of a word holding the bits of max |x| over `batch` matrices [rows, cols] (leading dimension ld)."""
    if _AMAX_REUSE is not None:
        return _AMAX_REUSE.slot(x, (batch, stride, rows, cols, ld))
    key = (x.device.index, stream_ptr())
    ring = _AMAX_RING.get(key)
    if ring is None:
        ring = _AMAX_RING[key] = [torch.zeros(_AMAX_SLOTS * AMAX_WORDS, dtype=torch.int32, device=x.device), 0]
    if ring[1] == _AMAX_SLOTS:
        ring[0].zero_()
        ring[1] = 0
    slot = ring[0].data_ptr() + 4 * AMAX_WORDS * ring[1]
    ring[1] += 1
    call("gamer_absmax_f32", ptr(x), batch, stride, rows, cols, ld, slot, stream_ptr())
    return slot


def split3_planes(x, planes):
    """planes [3, n] bf16 <- the three exact bf16 pieces of x [n] fp32 (gamer_split3_planes)."""
    call("gamer_split3_planes", ptr(x), ptr(planes), x.numel(), planes.stride(0), stream_ptr())


# Ordered weight gradients (fp32 forms): the split-K chunks of gamer_gemm_f32 / _split store their partial tiles in a workspace
# and a second kernel adds them in chunk order (gamer_gemm_desc.wgrad_ws) instead of combining them with fp32 atomics, whose
# order varies from run to run.  ON by default since round 4: measured equal or faster than the atomics (247.4 vs 247.0 ms per
# step at per-GPU batch 1024, 33.10 vs 33.20 at 128 - 25 GB of float atomics per step become plain stores).
# GAMER_WGRAD_TWO_PASS=0 or `with ops.deterministic(False):` give the atomics form.
DETERMINISTIC_WGRAD = os.environ.get("GAMER_WGRAD_TWO_PASS", "1") != "0"
# The bf16 weight-gradient GEMM has the same ordered form (gamer_gemm_bf16_desc.wgrad_ws, same second pass).  There it measured
# 0.6 ms per step SLOWER than the atomics (104.2 against 103.6 ms at per-GPU batch 1024, equal at 128; the bf16 chunks are long and
# few, so there are few atomics to save), so it is opt-in: Engine(dtype="bf16", deterministic=True), GAMER_WGRAD_TWO_PASS_BF16=1.
DETERMINISTIC_WGRAD_BF16 = os.environ.get("GAMER_WGRAD_TWO_PASS_BF16", "0") == "1"
_WGRAD_WS = {}


class deterministic:
    """Scope: ordered (two-pass) weight gradients on / off for the fp32 forms; ``bf16`` (None = leave alone) likewise for the
    bf16 weight-gradient GEMM."""

    def __init__(self, on: bool = True, bf16=None):
        self.on, self.bf16 = bool(on), bf16

    def __enter__(self):
        global DETERMINISTIC_WGRAD, DETERMINISTIC_WGRAD_BF16
        self.prev, DETERMINISTIC_WGRAD = DETERMINISTIC_WGRAD, self.on
        self.prev16 = DETERMINISTIC_WGRAD_BF16
        if self.bf16 is not None:
            DETERMINISTIC_WGRAD_BF16 = bool(self.bf16)
        return self

    def __exit__(self, *exc):
        global DETERMINISTIC_WGRAD, DETERMINISTIC_WGRAD_BF16
        DETERMINISTIC_WGRAD, DETERMINISTIC_WGRAD_BF16 = self.prev, self.prev16
        return False


def _wgrad_workspace(device, floats: int) -> torch.Tensor:
    """grow-only scratch per device for the chunk partial tiles of the deterministic weight gradient"""
    t = _WGRAD_WS.get(device)
    if t is None or t.numel() < floats:
        _WGRAD_WS[device] = None
        t = _WGRAD_WS[device] = torch.empty(int(floats), dtype=torch.float32, device=device)
    return t


class split3_guard:
    """Context manager: the row-range guard of the split3 GEMMs on / off for the block (gamer_split3_guard in include/gamer_hip.h;
    on by default).  Tests turn it off to show what it guards against."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _lib.load().gamer_split3_guard(1 if self.on else 0)
        return self

    def __exit__(self, *exc):
        _lib.load().gamer_split3_guard(self.prev)
        return False


def set_f32_matmul(mode) -> int:
    """mode: "f32" | "split6" | "split9" (or 0 / 6 / 9); returns the previous setting."""
    global F32_MATMUL_TERMS
    prev = F32_MATMUL_TERMS
    terms = MATMUL_MODES[mode] if isinstance(mode, str) else int(mode)
    if terms not in (0, 3, 6, 9):
        raise ValueError(f"fp32 matmul mode {mode!r}: one of {sorted(MATMUL_MODES)}")
    F32_MATMUL_TERMS = terms
    return prev


def gemm(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, alpha=1.0, accumulate=False, groups=1, group_mode=0,
         group_offsets=None, strideB=0, strideC=0, kchunk=0, resid=None, row_map=None, p_drop=0.0, seed=0,
         rowdot=None, qknorm=None, c_amax=None, swiglu_bwd=None, group_div=0, sw_tbl=None, swiglu_fwd=None):
    """C[m][n] (=|+=) alpha * sum_k A(m,k) B(n,k); see gamer_gemm_desc in include/gamer_hip.h.  bf16 operands go to
    gamer_gemm_bf16 (k-contiguous x k-contiguous, or the token-major wgrad form; see gamer_gemm_bf16_desc)."""
    if A.dtype == torch.bfloat16:
        if group_div > 1 or sw_tbl is not None or swiglu_fwd is not None:
            raise RuntimeError("group_div / sw_tbl / swiglu_fwd are options of the fp32 GEMM")
        return _gemm_bf16(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, alpha, accumulate, groups, group_mode,
                          group_offsets, strideB, strideC, kchunk, resid, row_map, p_drop, seed, rowdot, qknorm, swiglu_bwd)
    d = GemmDesc()
    d.A = ptr(A); d.a_rs = a_rs; d.a_ks = a_ks
    d.B = ptr(Bm); d.b_rs = b_rs; d.b_ks = b_ks
    d.C = ptr(Cm); d.ldc = ldc
    d.M = M; d.N = N; d.K = K
    d.alpha = alpha
    d.accumulate = 1 if accumulate else 0
    d.groups = groups
    d.group_mode = group_mode
    d.group_offsets = ptr(group_offsets)
    d.strideB = strideB; d.strideC = strideC
    d.kchunk = kchunk
    d.resid = ptr(resid)
    d.row_map = ptr(row_map)
    d.p_drop = p_drop
    d.seed = seed
    if rowdot is not None:                 # (other [M, ldc], out [M / S, N / 64, S], S): see gamer_gemm_desc
        d.rowdot_other, d.rowdot_out, d.rowdot_S = ptr(rowdot[0]), ptr(rowdot[1]), int(rowdot[2])
    if qknorm is not None:                 # the q|k|v epilogue: dict with the arguments of qknorm_rope_fwd
        q = qknorm
        d.qk_wq, d.qk_wk, d.qk_eps = ptr(q["wq"]), ptr(q["wk"]), float(q["eps"])
        d.qk_cos, d.qk_sin = ptr(q["cos"]), ptr(q["sin"])
        d.qk_bias_q, d.qk_bias_k, d.qk_bias_v = ptr(q.get("bias_q")), ptr(q.get("bias_k")), ptr(q.get("bias_v"))
        d.qk_act_idx, d.qk_pos_ids = ptr(q.get("act_idx")), ptr(q.get("pos_ids"))
        d.qk_q_rot, d.qk_k_rot = ptr(q["q_rot"]), ptr(q["k_rot"])
        d.qk_S, d.qk_nq, d.qk_nkv = int(q["S"]), int(q["nq"]), int(q["nkv"])
    if (c_amax is not None and _AMAX_REUSE is not None and F32_MATMUL_TERMS == 3 and group_mode == 0 and not accumulate
            and resid is None and qknorm is None and c_amax[1] % 64 == 0):
        # c_amax = (view of C from column col0 on, col0): the maximum of those columns comes out of this GEMM's epilogue into the
        # slot the next matrix product that reads the view (with this extent) will take - no gamer_absmax_f32 pass over it
        view, col0 = c_amax
        slot_c = _AMAX_REUSE.preset(view, (1, 0, M, N - col0, ldc))
        d.amax_c, d.amax_c_col0 = slot_c, int(col0)
    if swiglu_bwd is not None:
        # (gu, ld): C = d(hm) is consumed by the SwiGLU backward in the epilogue - gu <- d gate | d up - and never stored
        gu, ld_gu = swiglu_bwd
        d.sw_gu, d.sw_ld = ptr(gu), int(ld_gu)
        d.sw_tbl = ptr(sw_tbl)
        if _AMAX_REUSE is not None and F32_MATMUL_TERMS == 3 and ld_gu == 2 * N:
            d.amax_c, d.amax_c_col0 = _AMAX_REUSE.preset(gu, (1, 0, 1, M * ld_gu, M * ld_gu)), 0
    d.group_div = int(group_div)
    if swiglu_fwd is not None:
        # (hm, tbl or None, row_group or None): C = gate | up AND hm = dropout(silu(gate + tg) * (up + tu)) from one call
        # (gamer_gemm_desc.sw_hm); the maximum of hm goes to the slot its consumer GEMM will take
        hm, tbl, row_group = swiglu_fwd
        d.sw_hm, d.sw_tbl, d.sw_row_group = ptr(hm), ptr(tbl), ptr(row_group)
        if _AMAX_REUSE is not None and F32_MATMUL_TERMS == 3:
            d.amax_c, d.amax_c_col0 = _AMAX_REUSE.preset(hm, (1, 0, 1, M * (N // 2), M * (N // 2))), 0
    if group_mode == 1 and DETERMINISTIC_WGRAD:
        n_chunks = (K + kchunk - 1) // kchunk + (groups if group_offsets is not None else 0)
        need = n_chunks * ((M + 127) // 128) * ((N + 127) // 128) * 16384
        ws = _wgrad_workspace(A.device, need)
        d.wgrad_ws, d.wgrad_ws_floats = ws.data_ptr(), ws.numel()
    if F32_MATMUL_TERMS == 3:
        # operand extents: A(m, k) at A + m a_rs + k a_ks, B(n, k) at B + n b_rs + k b_ks (one of each stride pair is 1)
        ga = (M, K, a_rs) if a_ks == 1 else (K, M, a_ks)
        gb = (N, K, b_rs) if b_ks == 1 else (K, N, b_ks)
        nb = (groups // max(1, int(group_div))) if (group_mode == 0 and groups > 1) else 1
        d.amax_a = absmax_slot(A, 1, 0, *ga)
        geom_b = (nb, strideB if nb > 1 else 0) + gb
        c = _AMAX_REUSE
        parent = c.plane_parent(Bm.data_ptr(), geom_b) if (c is not None and c.planes is not None and group_mode == 0) else None
        if parent is not None:
            # B is a VIEW (some columns of every row) of a parameter tensor with piece planes: the parent's maximum scales it (>= the
            # view's own; the planes were cut with it) and its pieces lie at the same offsets as the values
            d.amax_b = parent
            d.b_planes = c.planes.data_ptr() + (Bm.data_ptr() - c.stable_ranges[0][0])
            call("gamer_gemm_f32_split", C.byref(d), 3, stream_ptr())
            return
        d.amax_b = absmax_slot(Bm, *geom_b)
        if c is not None and c.planes is not None and group_mode == 0 and c._key(Bm.data_ptr(), geom_b) in c._plane_keys:
            # B is a parameter tensor whose fp16 piece planes were built at the start of this pass (same scale as amax_b gives)
            d.b_planes = c.planes.data_ptr() + (Bm.data_ptr() - c.stable_ranges[0][0])      # packed pieces at B's offsets
            if a_ks == 1 and b_ks == 1 and c._key(Bm.data_ptr(), geom_b) in c._tplane_keys:
                d.b_planes_t = c.planes_t.data_ptr() + (Bm.data_ptr() - c.stable_ranges[0][0])    # ... and of B^T
        call("gamer_gemm_f32_split", C.byref(d), 3, stream_ptr())
        return
    if F32_MATMUL_TERMS:
        if WEIGHT_PLANES is not None and group_mode == 0:
            base, nbytes, pl, stride = WEIGHT_PLANES
            off = Bm.data_ptr() - base
            if 0 <= off < nbytes:                       # B is a view of the engine's flat parameter buffer: hand over its planes
                d.b_planes, d.b_plane_stride = pl + off // 2, stride
        call("gamer_gemm_f32_split", C.byref(d), F32_MATMUL_TERMS, stream_ptr())
    else:
        call("gamer_gemm_f32", C.byref(d), stream_ptr())


def _gemm_bf16(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, alpha, accumulate, groups, group_mode, group_offsets,
               strideB, strideC, kchunk, resid, row_map, p_drop, seed, rowdot, qknorm=None, swiglu_bwd=None):
    if Bm.dtype != torch.bfloat16 or alpha != 1.0:
        raise RuntimeError("gamer_gemm_bf16 takes two bf16 operands and alpha = 1")
    d = GemmBf16Desc()
    if group_mode == 0:
        if a_ks != 1 or b_ks != 1:
            raise RuntimeError("bf16 GEMM: both operands must be k-contiguous (dgrad runs on the transposed weight copy)")
        d.lda, d.ldb = a_rs, b_rs
        want = torch.float32 if resid is not None else torch.bfloat16
    else:
        if a_rs != 1 or b_rs != 1:
            raise RuntimeError("bf16 wgrad: both operands must be token-major")
        d.lda, d.ldb = a_ks, b_ks
        want = torch.float32
    if Cm.dtype != want:
        raise RuntimeError(f"bf16 GEMM: C must be {want}, got {Cm.dtype}")
    d.A, d.B, d.C, d.ldc = ptr(A), ptr(Bm), ptr(Cm), ldc
    d.M, d.N, d.K = M, N, K
    d.accumulate = 1 if accumulate else 0
    d.groups, d.group_mode, d.group_offsets = groups, group_mode, ptr(group_offsets)
    d.strideB, d.strideC, d.kchunk = strideB, strideC, kchunk
    d.resid, d.row_map, d.p_drop, d.seed = ptr(resid), ptr(row_map), p_drop, seed
    if rowdot is not None:
        d.rowdot_other, d.rowdot_out, d.rowdot_S = ptr(rowdot[0]), ptr(rowdot[1]), int(rowdot[2])
    if qknorm is not None:                 # the q|k|v epilogue (AMP arithmetic of qknorm_rope_fwd on bf16 activations)
        q = qknorm
        d.qk_wq, d.qk_wk, d.qk_eps = ptr(q["wq"]), ptr(q["wk"]), float(q["eps"])
        d.qk_cos, d.qk_sin = ptr(q["cos"]), ptr(q["sin"])
        d.qk_bias_q, d.qk_bias_k, d.qk_bias_v = ptr(q.get("bias_q")), ptr(q.get("bias_k")), ptr(q.get("bias_v"))
        d.qk_act_idx, d.qk_pos_ids = ptr(q.get("act_idx")), ptr(q.get("pos_ids"))
        d.qk_q_rot, d.qk_k_rot = ptr(q["q_rot"]), ptr(q["k_rot"])
        d.qk_S, d.qk_nq, d.qk_nkv = int(q["S"]), int(q["nq"]), int(q["nkv"])
    if swiglu_bwd is not None:             # (gu, ld): see gamer_gemm_bf16_desc.sw_gu
        d.sw_gu, d.sw_ld = ptr(swiglu_bwd[0]), int(swiglu_bwd[1])
    if group_mode == 1 and DETERMINISTIC_WGRAD_BF16:
        n_chunks = (K + kchunk - 1) // kchunk + (groups if group_offsets is not None else 0)
        need = n_chunks * ((M + 127) // 128) * ((N + 127) // 128) * 16384
        ws = _wgrad_workspace(A.device, need)
        d.wgrad_ws, d.wgrad_ws_floats = ws.data_ptr(), ws.numel()
    call("gamer_gemm_bf16", C.byref(d), stream_ptr())


def linear_dgrad_t(dy, lddy, WT, ldwt, dx, lddx, M, K_out, N_in, accumulate=False, **grp):
    """dx[M,N_in] = dy[M,K_out] @ W[K_out,N_in] given WT = W^T [N_in, ldwt >= K_out] (k-contiguous on both sides: the
    form the bf16 path uses, against the transposed weight copy; K_out may include zero padding)."""
    gemm(dy, lddy, 1, WT, ldwt, 1, dx, lddx, M, N_in, K_out, accumulate=accumulate, **grp)


def qkv_fused_ok(x, M: int, N: int) -> bool:
    """The q|k|v GEMM can carry the per-head RMSNorm + RoPE epilogue: whole 128-row / 128-column tiles (fp32 or bf16)."""
    return x.dtype in (torch.float32, torch.bfloat16) and M % 128 == 0 and N % 128 == 0


def linear_fwd(x, ldx, W, ldw, y, ldy, M, N, K, accumulate=False, **grp):
    """y[M,N] = x[M,K] @ W[N,K]^T"""
    gemm(x, ldx, 1, W, ldw, 1, y, ldy, M, N, K, accumulate=accumulate, **grp)


def linear_dgrad(dy, lddy, W, ldw, dx, lddx, M, N_out, K_in, accumulate=False, **grp):
    """dx[M,K_in] = dy[M,N_out] @ W[N_out,K_in]"""
    gemm(dy, lddy, 1, W, 1, ldw, dx, lddx, M, K_in, N_out, accumulate=accumulate, **grp)


def pick_kchunk(rows: int, grouped: bool) -> int:
    """Token chunk per workgroup of the fp32 wgrad split.  Measured on MI355X at T = 517k (tools/wgrad_probe.py):
    1024 rows for the grouped (expert) form and 2048 for the plain one are within 3 % of the best for every
    shape on this path; 4k+ chunks lose 15-25 % to the tail (too few, too long workgroups)."""
    chunk = 1024 if grouped else 2048
    while chunk > 256 and rows < 64 * chunk:          # small problems: keep >= ~64 chunks
        chunk //= 2
    return chunk


# Token chunk of the split-K wgrad, measured once per (shape, dtype, matmul form) on first use: the best chunk depends on
# how tiles x chunks lands on the chip's 512 workgroup slots (tools/wgrad_probe.py: at 64,640 tokens the fixed rule loses
# 10-17 % on the o_proj, expert and head shapes; at 517k tokens 0-3 %).  GAMER_WGRAD_TUNE=0 keeps the fixed rules.
_WGRAD_TUNED = {}
_WGRAD_TUNE = os.environ.get("GAMER_WGRAD_TUNE", "1") != "0"


def _tune_allowed() -> bool:
    """On-line measurement only in single-process runs: under data parallelism every rank must pick the SAME chunk for
    a shape (the chunking is the fp32 summation partition of the weight gradient, and a sweep on one rank would skew
    step 0), so ranks use the shipped / file table (gamer_amd/wgrad_chunks.json, GAMER_WGRAD_TUNE_FILE) and the fixed
    rule - all deterministic functions of the shape."""
    if not _WGRAD_TUNE:
        return False
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
    except Exception:                               # noqa: BLE001
        pass
    return int(os.environ.get("WORLD_SIZE", "1")) <= 1
# GAMER_WGRAD_TUNE_FILE=<json>: chunks measured by an earlier process are read from / added to this file (profiling runs:
# the sweep's launches would otherwise sit in the kernel statistics of the profiled step)
_WGRAD_TUNE_FILE = os.environ.get("GAMER_WGRAD_TUNE_FILE", "")


def _tune_key_str(key) -> str:
    return "|".join(str(k) for k in key)


_WGRAD_SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wgrad_chunks.json")


def _load_tune_file():
    """Shipped table (chunks measured on MI355X for the bench shapes, tools/wgrad_table.py) overlaid by the user's file."""
    import json
    out = {}
    for path in (_WGRAD_SHIPPED, _WGRAD_TUNE_FILE):
        if path == _WGRAD_SHIPPED and os.environ.get("GAMER_WGRAD_IGNORE_SHIPPED", "0") not in ("", "0"):      # (tools/wgrad_table.py re-measures)
            continue
        if path and os.path.exists(path):
            try:
                with open(path) as f:
                    out.update({k: int(v) for k, v in json.load(f).items()})
            except (OSError, ValueError):
                pass
    return out


def _save_tune_file():
    """Atomic (temp file + rename): concurrent writers cannot leave a torn JSON behind."""
    import json
    import tempfile
    try:
        d = os.path.dirname(os.path.abspath(_WGRAD_TUNE_FILE))
        fd, tmp = tempfile.mkstemp(prefix=".wgrad_tune.", dir=d)
        with os.fdopen(fd, "w") as f:
            json.dump(_WGRAD_FILE_CACHE, f, indent=0)
        os.replace(tmp, _WGRAD_TUNE_FILE)
    except OSError:
        pass


_WGRAD_FILE_CACHE = _load_tune_file()


def _rule_kchunk(dy, rows, N_out, K_in, groups):
    if dy.dtype == torch.bfloat16:
        # bf16 tiles take 16x less matrix time than fp32 ones: chunks as long as possible (fewer fp32 atomics: every chunk
        # adds the whole 128 x 128 tile) while the grid still fills the chip's 512 workgroup slots once
        tiles = ((N_out + 127) // 128) * ((K_in + 127) // 128)
        per_group = max(1, rows // max(groups, 1))
        chunks = max(1, 512 // (tiles * max(groups, 1)))
        kchunk = min(16384, max(256, -(-per_group // chunks)))
        return (kchunk + 63) // 64 * 64
    return pick_kchunk(rows, groups > 1)


def _tune_kchunk(dy, lddy, x, ldx, dW, lddw, rows, N_out, K_in, groups, group_offsets, strideC):
    rule = _rule_kchunk(dy, rows, N_out, K_in, groups)
    if dy.dtype == torch.bfloat16:
        cands = {rule, 512, 1024, 2048, 4096, 8192, 16384, 1088, 2112, 3136, 4160, 6208}      # (multiples of 64; not only powers of two: see below)
    else:
        cands = {rule, 384, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096}
        # ... and the chunks that make the grid a whole number of rounds of the chip's 512 workgroup slots (at per-GPU batch 128 a
        # power-of-two chunk leaves a quarter of the slots empty or starts a second, mostly empty round)
        tiles = ((N_out + 127) // 128) * ((K_in + 127) // 128)
        per_group = max(1, rows // max(groups, 1))
        for rounds in (1, 2, 3, 4):
            n_chunks = max(1, (512 * rounds) // (tiles * max(groups, 1)))
            c = (-(-per_group // n_chunks) + 31) // 32 * 32
            if 256 <= c <= 8192:
                cands.add(c)
        if F32_MATMUL_TERMS == 3 and N_out % 256 == 0 and K_in % 256 == 0:
            # the 256 x 256-tile kernel (csrc/gemm_wg.hip): one workgroup per CU, 256 slots; and chunks that are NOT a power of two -
            # with one, every workgroup's stream starts a multiple of 2 MB from the others' and they walk the memory channels in
            # step (measured: 2080 against 2048 tokens 5-15 % faster on the same shape, either kernel)
            tiles2 = (N_out // 256) * (K_in // 256)
            for rounds in (1, 2, 3, 4):
                n_chunks = max(1, (256 * rounds) // (tiles2 * max(groups, 1)))
                c = (-(-per_group // n_chunks) + 31) // 32 * 32
                if 256 <= c <= 8192:
                    cands.add(c)
            cands.update({1056, 2080, 3104, 4128})
    cands = sorted(c for c in cands if c <= max(256, rows))
    scratch = torch.zeros_like(dW)                 # the sweep must not touch the real gradient
    best, best_t = rule, float("inf")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for c in cands:
        ts = []
        for it in range(4):
            s.record()
            gemm(dy, 1, lddy, x, 1, ldx, scratch, lddw, N_out, K_in, rows, groups=groups, group_mode=1,
                 group_offsets=group_offsets, strideC=strideC, kchunk=c)
            e.record()
            e.synchronize()
            if it:
                ts.append(s.elapsed_time(e))
        t = min(ts)
        if t < best_t * (0.98 if c != rule else 1.0):      # ties go to the smaller chunk / the rule
            best, best_t = c, t
    return best


def linear_wgrad(dy, lddy, x, ldx, dW, lddw, rows, N_out, K_in, groups=1, group_offsets=None, strideC=0, kchunk=None):
    """dW[N_out,K_in] += dy[rows,N_out]^T @ x[rows,K_in]   (dW must be initialised).  The token chunks of the split-K form are
    combined in chunk order through a workspace (two passes, the default in fp32; opt-in for bf16: ``deterministic``) or with fp32
    atomics (GAMER_WGRAD_TWO_PASS=0)."""
    if kchunk is None:
        key = (rows, N_out, K_in, groups, dy.dtype, F32_MATMUL_TERMS)
        kchunk = _WGRAD_TUNED.get(key)
        if kchunk is None and _tune_key_str(key) in _WGRAD_FILE_CACHE:
            kchunk = _WGRAD_TUNED[key] = _WGRAD_FILE_CACHE[_tune_key_str(key)]
        if kchunk is None:
            capturing = torch.cuda.is_current_stream_capturing()
            if rows >= 4096 and not capturing and _tune_allowed():
                kchunk = _tune_kchunk(dy, lddy, x, ldx, dW, lddw, rows, N_out, K_in, groups, group_offsets, strideC)
                if _WGRAD_TUNE_FILE:
                    _WGRAD_FILE_CACHE[_tune_key_str(key)] = int(kchunk)
                    _save_tune_file()
            else:
                kchunk = _rule_kchunk(dy, rows, N_out, K_in, groups)
            if not capturing:
                _WGRAD_TUNED[key] = kchunk
    gemm(dy, 1, lddy, x, 1, ldx, dW, lddw, N_out, K_in, rows, groups=groups, group_mode=1,
         group_offsets=group_offsets, strideC=strideC, kchunk=kchunk)


def qknorm_rope_fwd(qkv, S, nq, nkv, wq, wk, eps, cos_t, sin_t, q_rot, k_rot, bias_q=None, bias_k=None, bias_v=None,
                    act_idx=None, pos_ids=None):
    """pos_ids: int32 [T] RoPE table row per token (session model); None = position in the sequence."""
    T = qkv.shape[0]
    outs = [(q_rot, (1, 0, T, nq * 64, nq * 64), False), (k_rot, (1, 0, T, nkv * 64, nkv * 64), False)]
    if bias_v is not None:
        # the cross block: v + bias_v is written back into the v columns of q|k|v - their maximum is the attention's third operand
        outs.append((qkv[:, (nq + nkv) * 64:], (1, 0, T, nkv * 64, qkv.stride(0)), False))
    _arm_sink(*outs)
    call("gamer_qknorm_rope_fwd" + _sfx(qkv), ptr(qkv), T, S, nq, nkv, ptr(wq), ptr(wk), eps, ptr(cos_t), ptr(sin_t),
         ptr(bias_q), ptr(bias_k), ptr(bias_v), ptr(act_idx), ptr(q_rot), ptr(k_rot), ptr(pos_ids), stream_ptr())


def qknorm_partial_numel(nb1: int = 0) -> int:
    """fp32 scratch of qknorm_rope_bwd that never limits its wave count (see gamer_hip.h)."""
    return 8192 * (1 + nb1) * 64


def qknorm_rope_bwd(qkv, dq_rot, dk_rot, S, nq, nkv, wq, wk, eps, cos_t, sin_t, dqkv, dwq, dwk, bias_q=None,
                    bias_k=None, act_idx=None, nb1=0, dbias_q=None, dbias_k=None, dbias_v=None, pos_ids=None,
                    partial=None):
    """partial: fp32 scratch for the per-wave sums of the weight / bias gradients (allocated here when None)."""
    T = qkv.shape[0]
    if partial is None:
        partial = torch.empty(qknorm_partial_numel(nb1), dtype=torch.float32, device=qkv.device)
    # the q and k columns of d(q|k|v) are written here, the v columns by the attention backward before: its slot, if it opened one
    _arm_sink((dqkv, (1, 0, T, dqkv.shape[1], dqkv.stride(0)), "only"))
    call("gamer_qknorm_rope_bwd" + _sfx(qkv), ptr(qkv), ptr(dq_rot), ptr(dk_rot), T, S, nq, nkv, ptr(wq), ptr(wk), eps,
         ptr(cos_t), ptr(sin_t), ptr(bias_q), ptr(bias_k), ptr(act_idx), nb1, ptr(dqkv), ptr(dwq), ptr(dwk),
         ptr(dbias_q), ptr(dbias_k), ptr(dbias_v), ptr(pos_ids), ptr(partial), partial.numel(), stream_ptr())


def attn_row_order(row_empty, perm, tile_kind, tile_maxpos):
    B, S = row_empty.shape
    call("gamer_attn_row_order", ptr(row_empty), B, S, ptr(perm), ptr(tile_kind), ptr(tile_maxpos), stream_ptr())


def attn_fwd(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse,
             order=None, uniform_len=0, q_span=None):
    """order = (perm, tile_kind, tile_maxpos) from attn_row_order, or None for the natural row order.
    uniform_len: see include/gamer_hip.h (0 = training semantics).
    q_span: int32 [B,S,4] per-query key spans from session_spans (session model), None = plain causal."""
    pm, tk, tm = order if order is not None else (None, None, None)
    call("gamer_attn_fwd", ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(kl), ptr(ql), ptr(row_empty), ptr(tile_empty),
         B, S, nq, nkv, scale, p_drop, seed, ptr(o), ptr(lse), ptr(pm), ptr(tk), ptr(tm), uniform_len, ptr(q_span),
         stream_ptr())


def attn_bwd(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed,
             delta, dq, lddq, dk, lddk, dv, lddv, order=None, ds_work=None, q_span=None, delta_ready=False):
    """ds_work: optional fp32 scratch of attn_ds_work_numel(B, S, nq) elements (dS spill, see gamer_hip.h)."""
    pm, tk, tm = order if order is not None else (None, None, None)
    call("gamer_attn_bwd", ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(o), ptr(d_o), ptr(lse), ptr(kl), ptr(ql),
         ptr(row_empty), ptr(tile_empty), B, S, nq, nkv, scale, p_drop, seed, ptr(delta), ptr(dq), lddq, ptr(dk),
         lddk, ptr(dv), lddv, ptr(pm), ptr(tk), ptr(tm), ptr(ds_work), ptr(q_span),
         1 if (delta_ready and ds_work is not None) else 0, stream_ptr())


def attn_operand_maxima(q, ldq, k, ldk, v, ldv, T, nq, nkv, d_o=None):
    """Slots with the maxima of the attention operands (the three-product fp16 form): measured here unless their producers
    left them in the cache in effect (ops.amax_reuse)."""
    sl = [absmax_slot(q, 1, 0, T, nq * 64, ldq), absmax_slot(k, 1, 0, T, nkv * 64, ldk), absmax_slot(v, 1, 0, T, nkv * 64, ldv)]
    sl.append(absmax_slot(d_o, 1, 0, T, nq * 64, nq * 64) if d_o is not None else None)
    return sl


def attn_fwd_split(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, nkv, scale, p_drop, seed, o, lse, order=None,
                   h2=False, uniform_len=0, q_span=None):
    """gamer_attn_fwd with its products on the 16-bit matrix pipe; training semantics.  h2 = False: exact three-way bf16
    cuts, six piece products; h2 = True: two-way fp16 cuts scaled per tensor, three piece products (gamer_attn_split_amax)."""
    pm, tk, tm = order if order is not None else (None, None, None)
    if h2:
        call("gamer_attn_split_amax", *attn_operand_maxima(q, ldq, k, ldk, v, ldv, B * S, nq, nkv))
    if _AMAX_ATTN:
        _arm_sink((o, (1, 0, 1, B * S * nq * 64, B * S * nq * 64), False))
    call("gamer_attn_fwd_split", ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(kl), ptr(ql), ptr(row_empty), B, S, nq, nkv,
         scale, p_drop, seed, ptr(o), ptr(lse), ptr(pm), ptr(tk), ptr(tm), int(uniform_len), ptr(q_span), stream_ptr())


def attn_bwd_split(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, row_empty, tile_empty, B, S, nq, nkv, scale, p_drop, seed,
                   delta, dq, lddq, dk, lddk, dv, lddv, order=None, delta_ready=False, ds_work=None, dv_of=None, h2=False,
                   q_span=None):
    """gamer_attn_bwd with its products on the bf16 pipe; delta_ready: `delta` already holds dO.O; ds_work: the dS spill
    scratch of attn_bwd (None = recompute form); dv_of: the d(q|k|v) tensor whose v columns `dv` is (its maximum is then
    collected by the kernels that write it: this one and qknorm_rope_bwd)."""
    pm, tk, tm = order if order is not None else (None, None, None)
    if h2:           # the three-product fp16 form (see attn_fwd_split)
        call("gamer_attn_split_amax", *attn_operand_maxima(q, ldq, k, ldk, v, ldv, B * S, nq, nkv, d_o=d_o))
    if dv_of is not None and _AMAX_ATTN:
        _arm_sink((dv_of, (1, 0, dv_of.shape[0], dv_of.shape[1], dv_of.stride(0)), False))
    call("gamer_attn_bwd_split", ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(o), ptr(d_o), ptr(lse), ptr(kl), ptr(ql),
         ptr(row_empty), ptr(tile_empty), B, S, nq, nkv, scale, p_drop, seed, ptr(delta), ptr(dq), lddq, ptr(dk), lddk, ptr(dv),
         lddv, ptr(pm), ptr(tk), ptr(tm), 1 if delta_ready else 0, ptr(ds_work), ptr(q_span), stream_ptr())


def attn_fwd_bf16(q, ldq, k, ldk, v, ldv, kl, ql, B, S, nq, nkv, scale, p_drop, seed, o, lse, q_span=None, order=None):
    """bf16 attention (empty rows -> 0, see gamer_attn_fwd_bf16 in include/gamer_hip.h).
    order = (perm, tile_maxpos, row_empty): visit the query rows through the row order of attn_row_order."""
    perm, tmax, rempty = order if order is not None else (None, None, None)
    call("gamer_attn_fwd_bf16", ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(kl), ptr(ql), B, S, nq, nkv, scale, p_drop,
         seed, ptr(o), ptr(lse), ptr(q_span), ptr(perm), ptr(tmax), ptr(rempty), stream_ptr())


def attn_bwd_bf16(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, B, S, nq, nkv, scale, p_drop, seed, delta, dq, lddq, dk,
                  lddk, dv, lddv, q_span=None, delta_ready=False, order=None):
    perm, tmax, rempty = order if order is not None else (None, None, None)
    call("gamer_attn_bwd_bf16", ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(o), ptr(d_o), ptr(lse), ptr(kl), ptr(ql),
         B, S, nq, nkv, scale, p_drop, seed, ptr(delta), ptr(dq), lddq, ptr(dk), lddk, ptr(dv), lddv, ptr(q_span),
         1 if delta_ready else 0, ptr(perm), ptr(tmax), ptr(rempty), stream_ptr())


def cast_params_bf16(flat, out, out_t, table, n_entries, n_tiles):
    """bf16 (and transposed bf16) operand copies of the fp32 master parameters; table from engine.Bf16Shadow."""
    call("gamer_cast_params_bf16", ptr(flat), ptr(out), ptr(out_t), ptr(table), n_entries, n_tiles, stream_ptr())


def attn_ds_work_numel(B, S, nq):
    n = (S + 31) // 32
    return B * nq * n * n * 1024


def residual_dropout_fwd(x, delta, p, seed, src_rows=None, out=None):
    """out = x + drop(delta[src]); out defaults to x (in place)."""
    T, H = x.shape
    call("gamer_residual_dropout_fwd", ptr(x), ptr(delta), ptr(src_rows), T, H, p, seed,
         ptr(out if out is not None else x), stream_ptr())


def residual_dropout_bwd(dx, p, seed, ddelta, src_rows=None):
    T, H = dx.shape
    call("gamer_residual_dropout_bwd", ptr(dx), ptr(src_rows), T, H, p, seed, ptr(ddelta), stream_ptr())


def swiglu_fwd(g, u, n, p, seed, hm):
    _arm_sink((hm, (1, 0, 1, n, n), False))
    call("gamer_swiglu_fwd" + _sfx(g), ptr(g), ptr(u), n, p, seed, ptr(hm), stream_ptr())


def swiglu_bwd(g, u, dhm, n, p, seed):
    _arm_sink((g, (1, 0, 1, n, n), False), (u, (1, 0, 1, n, n), False))
    call("gamer_swiglu_bwd" + _sfx(g), ptr(g), ptr(u), ptr(dhm), n, p, seed, stream_ptr())


def swiglu_fwd_ld(gu, ld, T, I, p, seed, hm):
    """hm[T, I] = drop(silu(gu[:, :I]) * gu[:, I:2I]) on the output of the fused gate|up projection (row stride ld)."""
    _arm_sink((hm, (1, 0, 1, T * I, T * I), False))
    call("gamer_swiglu_fwd_ld" + _sfx(gu), ptr(gu), ld, T, I, p, seed, ptr(hm), stream_ptr())


def swiglu_bwd_ld(gu, ld, T, I, dhm, p, seed):
    """In place: gu[:, :I] <- d gate, gu[:, I:2I] <- d up.  The maximum of the whole [T, 2I] gradient goes to ONE slot (it is
    one GEMM operand from here on) when the buffer is contiguous (ld == 2 I)."""
    if ld == 2 * I:
        geom = (1, 0, 1, T * ld, T * ld)
        _arm_sink((gu, geom, False), (gu, geom, True))
    call("gamer_swiglu_bwd_ld" + _sfx(gu), ptr(gu), ld, T, I, ptr(dhm), p, seed, stream_ptr())


def swiglu_fwd_ld_tbl(gu, ld, T, I, p, seed, hm, tbl, row_group):
    """hm = drop(silu(g) * u) with g | u = gu[t] + tbl[row_group[t]] (fp32; see gamer_inject_table_fwd)."""
    _arm_sink((hm, (1, 0, 1, T * I, T * I), False))
    call("gamer_swiglu_fwd_ld_tbl", ptr(gu), ld, T, I, p, seed, ptr(hm), ptr(tbl), ptr(row_group), stream_ptr())


def swiglu_bwd_ld_tbl(gu, ld, T, I, dhm, p, seed, tbl, row_group):
    if ld == 2 * I:
        _arm_sink((gu, (1, 0, 1, T * ld, T * ld), False), (gu, (1, 0, 1, T * ld, T * ld), True))
    call("gamer_swiglu_bwd_ld_tbl", ptr(gu), ld, T, I, ptr(dhm), p, seed, ptr(tbl), ptr(row_group), stream_ptr())


def inject_table_fwd(Eb, W, ldw, col0, E, twoI, tbl):
    """tbl [E * NB1, 2I] <- the behaviour-embedding share of the gate|up projection per (expert, behaviour) (gamer_inject_table_fwd)."""
    NB1, EB = Eb.shape
    call("gamer_inject_table_fwd", ptr(Eb), ptr(W), ldw, col0, E, twoI, NB1, EB, ptr(tbl), stream_ptr())


def segment_colsum_ws_floats(rows, cols, nseg) -> int:
    return int(_lib.load().gamer_segment_colsum_ws_floats(rows, cols, nseg))


def segment_colsum(x, ld, rows, cols, offsets, nseg, ws, out):
    call("gamer_segment_colsum", ptr(x), ld, rows, cols, ptr(offsets), nseg, ptr(ws), ws.numel(), ptr(out), stream_ptr())


def inject_table_bwd(seg, Eb, W, ldw, col0, E, twoI, dW, dEb, scratch):
    NB1, EB = Eb.shape
    call("gamer_inject_table_bwd", ptr(seg), ptr(Eb), ptr(W), ldw, col0, E, twoI, NB1, EB, ptr(dW), ptr(dEb), ptr(scratch), stream_ptr())


def silu_gate_fwd(a, gate, out, resid=None, p=0.0, seed=0):
    """out = a * silu(gate), or resid + dropout(a * silu(gate)) when resid is given (fused residual add)."""
    call("gamer_silu_gate_fwd" + _sfx(a), ptr(a), ptr(gate), a.numel(), ptr(out), ptr(resid), p, seed, stream_ptr())


def silu_gate_bwd(a, gate, dout, da, dgate, p=0.0, seed=0):
    """p > 0: dout is the residual-stream gradient and the forward's dropout mask (seed) is applied to it first."""
    _arm_sink((da, (1, 0, 1, a.numel(), a.numel()), False), (dgate, (1, 0, 1, a.numel(), a.numel()), False))
    call("gamer_silu_gate_bwd" + _sfx(a), ptr(a), ptr(gate), ptr(dout), a.numel(), ptr(da), ptr(dgate), p, seed, stream_ptr())


def check_labels(labels, V, ignore_index, bad_label):
    call("gamer_check_labels", ptr(labels), labels.numel(), V, ignore_index, ptr(bad_label), stream_ptr())


def ce_fwd(logits, ldl, labels, V, temperature, ignore_index, lse, row_loss, loss_sum, count):
    B, S = labels.shape
    call("gamer_ce_fwd" + _sfx(logits), ptr(logits), ldl, ptr(labels), B, S, V, temperature, ignore_index, ptr(lse), ptr(row_loss),
         ptr(loss_sum), ptr(count), stream_ptr())


def ce_bwd(logits, ldl, labels, V, temperature, ignore_index, lse, count_dev, denom_host, dloss, dloss_dev=None):
    """dloss_dev: optional fp32 device scalar multiplied into dloss (autograd's incoming gradient, no host read)."""
    B, S = labels.shape
    _arm_sink((logits, (1, 0, B * S, V, ldl), False))
    call("gamer_ce_bwd" + _sfx(logits), ptr(logits), ldl, ptr(labels), B, S, V, temperature, ignore_index, ptr(lse), ptr(count_dev),
         float(denom_host), float(dloss), ptr(dloss_dev), stream_ptr())


def sumsq(g, partial):
    call("gamer_sumsq", ptr(g), g.numel(), ptr(partial), partial.numel(), stream_ptr())


def adamw(p, g, m, v, n_decay, lr, beta1, beta2, eps, weight_decay, step, max_norm, grad_scale, partial, norm_out):
    call("gamer_adamw", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), n_decay, lr, beta1, beta2, eps, weight_decay,
         step, max_norm, grad_scale, ptr(partial), partial.numel(), ptr(norm_out), stream_ptr())


def fill(t, value):
    call("gamer_fill_f32", ptr(t), t.numel(), float(value), stream_ptr())


# ---- evaluation path: trie-constrained beam-search scoring ---------------------------------------
def trie_logprobs(logits2d, row_index, beam_score, node, child_start, child_tok, V, scores):
    """scores[n] = log_softmax(logits2d[row_index[n], :V]) + beam_score[n] on the child tokens of trie node[n],
    -inf elsewhere."""
    N = row_index.numel()
    call("gamer_trie_logprobs", ptr(logits2d), logits2d.stride(0), ptr(row_index), ptr(beam_score), ptr(node),
         ptr(child_start), ptr(child_tok), N, V, ptr(scores), stream_ptr())


def trie_advance(node, token, child_start, child_tok, child_node, out):
    call("gamer_trie_advance", ptr(node), ptr(token), ptr(child_start), ptr(child_tok), ptr(child_node),
         node.numel(), ptr(out), stream_ptr())


def kv_append(k, v, kg, vg, g):
    """kg / vg [N, tmax, C] <- k [N, C], v [N, C] (row-strided views) at generated position g."""
    N, tmax, C = kg.shape
    call("gamer_kv_append", ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(kg), ptr(vg), kg.stride(1), tmax, g, N, C, stream_ptr())


def attn_decode(q, kp, vp, key_ok, kg, vg, t, gen_ok, uniform, B, nb, L0, nq, nkv, scale, o, amax=None):
    """One new token per beam against the prompt cache (per sample) + generated cache (per beam).
    amax = (slot of max |kp|, slot of max |vp|): the three-piece fp16 form (gamer_attn_decode_split)."""
    tmax = kg.shape[1]
    if amax is not None:
        n = o.numel()
        _arm_sink((o, (1, 0, 1, n, n), False))
        call("gamer_attn_decode_split", ptr(q), q.stride(0), ptr(kp), kp.stride(0), ptr(vp), vp.stride(0), ptr(key_ok), ptr(kg),
             ptr(vg), kg.stride(1), tmax, t, 1 if gen_ok else 0, ptr(uniform), B, nb, L0, nq, nkv, scale, ptr(o), amax[0], amax[1],
             stream_ptr())
        return
    call("gamer_attn_decode", ptr(q), q.stride(0), ptr(kp), kp.stride(0), ptr(vp), vp.stride(0), ptr(key_ok), ptr(kg),
         ptr(vg), kg.stride(1), tmax, t, 1 if gen_ok else 0, ptr(uniform), B, nb, L0, nq, nkv, scale, ptr(o),
         stream_ptr())


# ---- post-LN encoder of the discriminative baselines (SeqRec/modules/layers/transformer.py) --------------------
ACTIVATIONS = {"none": 0, "relu": 1, "gelu": 2, "swish": 3, "tanh": 4, "sigmoid": 5, "elu": 6}


def bias_act_fwd(x, bias, act: int, y=None):
    """x[T,N] <- x + bias in place; y = act(x + bias) when given."""
    T, N = x.shape
    call("gamer_bias_act_fwd", ptr(x), ptr(bias), T, N, act, ptr(y), stream_ptr())


def bias_act_bwd(pre, dy, act: int, dx, db_partial):
    T, N = dy.shape
    call("gamer_bias_act_bwd", ptr(pre), ptr(dy), T, N, act, ptr(dx), ptr(db_partial), db_partial.shape[0], stream_ptr())


def layernorm_fwd(x, res, w, b, eps, v_out, y, mean, rstd):
    T, H = x.shape
    call("gamer_layernorm_fwd", ptr(x), ptr(res), ptr(w), ptr(b), T, H, eps, ptr(v_out), ptr(y), ptr(mean), ptr(rstd),
         stream_ptr())


def layernorm_bwd(v, w, mean, rstd, dy, dx, dw_partial, db_partial):
    T, H = v.shape
    call("gamer_layernorm_bwd", ptr(v), ptr(w), ptr(mean), ptr(rstd), ptr(dy), T, H, ptr(dx), ptr(dw_partial),
         ptr(db_partial), dw_partial.shape[0], stream_ptr())


def _mask_strides(mask, B, H, S):
    """element strides of an additive mask broadcastable to [B,H,S,S] (size-1 dims get stride 0)"""
    if mask is None:
        return None, None
    if mask.dim() != 4 or mask.dtype != torch.float32:
        raise RuntimeError("attention mask must be a 4-d float32 tensor broadcastable to [B, heads, S, S]")
    want = (B, H, S, S)
    st = []
    for d in range(4):
        if mask.shape[d] == want[d] and mask.shape[d] != 1:
            st.append(mask.stride(d))
        elif mask.shape[d] == 1:
            st.append(0)
        else:
            raise RuntimeError(f"attention mask shape {tuple(mask.shape)} does not broadcast to {want}")
    return mask, (C.c_int64 * 4)(*st)


def attn_dense_fwd(q, k, v, mask, B, S, H, dh, scale, p_drop, seed, o, lse):
    m, st = _mask_strides(mask, B, H, S)
    call("gamer_attn_dense_fwd", ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(m), st, B, S, H, dh,
         scale, p_drop, seed, ptr(o), o.stride(0), ptr(lse), stream_ptr())


def attn_dense_bwd(q, k, v, mask, B, S, H, dh, scale, p_drop, seed, o, d_o, lse, dq, dk, dv):
    m, st = _mask_strides(mask, B, H, S)
    call("gamer_attn_dense_bwd", ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(m), st, B, S, H, dh,
         scale, p_drop, seed, ptr(o), ptr(d_o), o.stride(0), ptr(lse), ptr(dq), dq.stride(0), ptr(dk), dk.stride(0),
         ptr(dv), dv.stride(0), stream_ptr())
