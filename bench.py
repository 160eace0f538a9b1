#!/usr/bin/env python3
"""Headline benchmark: train-step sequences/s of the Qwen3Multi SMB decoder (max_his_len=100).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 it is launched with
torch.distributed.run, one rank per GPU (RCCL) - and when it is started WITHOUT a launcher (no WORLD_SIZE in the
environment) it starts the N ranks itself, as ref:scripts/train_SMB_decoder.sh:123-153 does, and relays rank 0's
line.  W untimed steps, then exactly K timed steps between barrier + synchronize, MAX over ranks, rank 0 prints
ONE JSON line.

Workload (BASELINE.json configs[1]): shipped Qwen3Multi architecture (8 layers, hidden 256, 6/3 heads,
6 position-routed experts, V=1041), fp32, per-GPU micro-batch 1024 sequences x 505 tokens
(101 items x 5 tokens), synthetic ids, seeded random weights, dropout 0.2 ON, one step =
forward + backward + (gradient all-reduce) + clip_grad_norm_(1.0) + AdamW.  Default = the north star's strong
scaling: global batch 1024, per-GPU batch 1024 / N (1024 / 512 / 256 / 128 at 1 / 2 / 4 / 8 GPUs); ``--weak`` keeps
1024 sequences per GPU (BASELINE configs[2]: 8 x 1024 with ``--dtype bf16``).

Extra objects on the JSON line:
  roofline     the dominant kernel family = the one with the most time per step among ALL families with a FLOP model
               (GEMM families and attention forward / backward, self / cross): algorithmic FLOPs of its launches
                / their summed duration, measured with HIP events on the launch stream inside the timed
               region; `roofline.gemm` = the same object for the largest GEMM family
  cpu_baseline the CPU oracle (a port of the reference algorithm, oracle/) timed on the host cores,
               rank 0, N=1 only, on a bounded sample (micro-batch 32, 1 warm-up + 3 timed
               steps, same sequence shape)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = 64 FLOP/clk/SIMD
BF16_MATRIX_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD)
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(cfg, batch, session: bool = False) -> dict:
    """per non-pad token 2*MACs of every projection / expert / head GEMM, plus
    1536 FLOP per allowed (query, key) pair per attention layer; train step = 3 x forward."""
    ids, keep, a = batch["input_ids"], batch["attention_mask"].bool(), batch["actions"]
    H, dh = cfg.hidden_size, cfg.head_dim
    nq, nkv, I, Eb = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.behavior_embedding_dim
    NQ, NKV = nq * dh, nkv * dh
    L = cfg.num_hidden_layers
    n_cross = len(cfg.cross_attention_decoder)
    n_inj = len(cfg.behavior_injection_decoder)
    attn_proj = 2 * H * (NQ + 2 * NKV) + 2 * NQ * H
    f_tok = L * attn_proj + n_cross * (attn_proj + 2 * H * H)
    f_tok += n_inj * (3 * 2 * (H + Eb) * I - 2 * Eb * I) + (L - n_inj) * 3 * 2 * H * I     # gate+up on Din, down on I
    f_tok += 2 * H * cfg.vocab_size
    t_nonpad = int(keep.sum())
    if session:
        # Qwen3SessionMulti masks: count the allowed pairs per sequence from the oracle-independent definition
        # (own item up to the query + strictly earlier sessions; cross additionally needs a lower level)
        p_self = p_cross = 0
        sess = batch["session_ids"]
        S = ids.shape[1]
        i_idx, j_idx = torch.arange(S)[:, None], torch.arange(S)[None, :]
        in_item = (j_idx // cfg.num_positions == i_idx // cfg.num_positions) & (j_idx <= i_idx)
        for b0 in range(0, ids.shape[0], 64):
            sl = slice(b0, b0 + 64)
            earlier = sess[sl, None, :] < sess[sl, :, None]
            kp = keep[sl, None, :]
            p_self += int(((in_item[None] | earlier) & kp).sum())
            p_cross += int((earlier & (a[sl, None, :] < a[sl, :, None]) & kp).sum())
    else:
        csum = keep.long().cumsum(1)
        p_self = int(csum.sum())                                       # sum_i #{j<=i, keep_j} (all query rows)
        p_cross = 0
        for lv in torch.unique(a).tolist():
            less = ((a < lv) & keep).long().cumsum(1)
            p_cross += int(less[a == lv].sum())
    pair = 4 * dh * nq
    f_fwd = t_nonpad * f_tok + pair * (L * p_self + n_cross * p_cross)
    return dict(per_token=f_tok, fwd=f_fwd, step=3 * f_fwd, tokens=t_nonpad, p_self=p_self, p_cross=p_cross)


# Attention FLOPs per allowed (query, key) pair and query head (1536 per pair forward at nq = 6, F_step = 3 F_fwd)
ATTN_FWD_FLOP_PER_PAIR_HEAD = 4.0 * 64
ATTN_BWD_FLOP_PER_PAIR_HEAD = 8.0 * 64


class KernelTimer:
    """HIP-event timing of every C-ABI launch on the launch stream (torch's current stream)."""

    def __init__(self):
        self.records = []
        self.enabled = False
        self._next = None
        self.only = None                         # when set: record only these labels (the timed region: dominant family)
        self._dgrad_t = False
        self._bytes = 0.0
        self.pairs = {"self": 0, "cross": 0}     # allowed (query, key) pairs of the current batch

    def reset(self):
        self.records, self.enabled, self._next, self.only = [], False, None, None
        self._dgrad_t, self._bytes = False, 0.0

    def install(self):
        from gamer_amd import ops
        orig_call, orig_gemm = ops.call, ops.gemm
        timer = self

        def timed_call(name, *args):
            if not timer.enabled or name in ("gamer_amax_sink", "gamer_attn_split_amax"):       # (not launches)
                return orig_call(name, *args)
            if name == "gamer_absmax_f32":              # split3: the operand maxima in front of a GEMM keep their own row
                label, flops, nbytes = "absmax", 0.0, 0.0
            else:
                label, flops = timer._next if timer._next else (name.replace("gamer_", "").replace("_bf16", ""), 0.0)
                nbytes = timer._bytes if timer._next else 0.0
                timer._next = None
                timer._bytes = 0.0
            if timer.only is not None and label not in timer.only:
                return orig_call(name, *args)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            orig_call(name, *args)
            e.record()
            timer.records.append((label, s, e, flops, nbytes))

        orig_dgrad_t = ops.linear_dgrad_t

        def timed_dgrad_t(*a, **kw):
            timer._dgrad_t = True                       # bf16 dgrad runs in the forward layout on the transposed weights
            try:
                return orig_dgrad_t(*a, **kw)
            finally:
                timer._dgrad_t = False

        def timed_gemm(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, **kw):
            kind = "gemm_fwd" if (a_ks == 1 and b_ks == 1) else ("gemm_dgrad" if a_ks == 1 else "gemm_wgrad")
            if timer._dgrad_t:
                kind = "gemm_dgrad"
            esz = A.element_size()
            timer._bytes = (M * K + N * K) * esz + M * N * Cm.element_size() * (2 if (kw.get("accumulate") or kw.get("resid") is not None) else 1)
            if kw.get("resid") is not None:
                kind = "gemm_fwd_resid"                 # fused residual + dropout epilogue (a different instantiation)
            if kw.get("rowdot") is not None:
                kind = "gemm_dgrad_delta"               # o_proj dgrad that also emits the attention backward's delta
            if kw.get("swiglu_fwd") is not None:
                kind = "gemm_fwd_swiglu"                # gate|up projection with the SwiGLU forward in its epilogue: C and hm are stored
                timer._bytes = (M * K + N * K) * esz + (M * N + M * N // 2) * esz
            if kw.get("swiglu_bwd") is not None:
                kind = "gemm_dgrad_swiglu"              # down-projection dgrad with the SwiGLU backward in its epilogue: C is not
                timer._bytes = (M * K + N * K) * esz + 4 * M * N * esz     # stored; gate | up are read and overwritten (4 M N)
            timer._next = (kind, 2.0 * M * N * K)
            return orig_gemm(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, **kw)

        orig_af16, orig_ab16 = ops.attn_fwd_bf16, ops.attn_bwd_bf16

        def timed_attn_fwd16(q, ldq, k, ldk, v, ldv, kl, ql, B, S, nq, *rest, **kw):
            pairs = timer.pairs["self" if ql is None else "cross"]
            timer._next = ("attn_fwd_self" if ql is None else "attn_fwd_cross", 4.0 * 64 * nq * pairs)
            return orig_af16(q, ldq, k, ldk, v, ldv, kl, ql, B, S, nq, *rest, **kw)

        def timed_attn_bwd16(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, B, S, nq, *rest, **kw):
            pairs = timer.pairs["self" if ql is None else "cross"]
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            was = timer.enabled and (timer.only is None or ("attn_bwd_self" if ql is None else "attn_bwd_cross") in timer.only)
            if not was:
                en = timer.enabled
                timer.enabled = False
                try:
                    return orig_ab16(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, B, S, nq, *rest, **kw)
                finally:
                    timer.enabled = en
            timer.enabled = False
            s.record()
            r = orig_ab16(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, B, S, nq, *rest, **kw)
            e.record()
            timer.enabled = True
            # algorithmic = SURVEY 8(d): F_step = 3 F_fwd, i.e. the backward is the FOUR products dV, dP, dQ, dK of 2 * 64 FLOP
            # per allowed pair and head (the recompute form executes seven: S and dP twice)
            timer.records.append(("attn_bwd_self" if ql is None else "attn_bwd_cross", s, e, ATTN_BWD_FLOP_PER_PAIR_HEAD * nq * pairs, 0.0))
            return r

        orig_af, orig_ab = ops.attn_fwd, ops.attn_bwd

        def timed_attn_fwd(q, ldq, k, ldk, v, ldv, kl, ql, *rest, **kw):
            nq = rest[4]
            pairs = timer.pairs["self" if ql is None else "cross"]
            timer._next = ("attn_fwd_self" if ql is None else "attn_fwd_cross", 4.0 * 64 * nq * pairs)
            return orig_af(q, ldq, k, ldk, v, ldv, kl, ql, *rest, **kw)

        def timed_attn_bwd(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, *rest, **kw):
            nq = rest[4]
            pairs = timer.pairs["self" if ql is None else "cross"]
            # the delta + dq + dkv launches share one entry point; the label goes to the first (delta),
            # so time the whole call as one record instead
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            was = timer.enabled and (timer.only is None or ("attn_bwd_self" if ql is None else "attn_bwd_cross") in timer.only)
            if not was:
                en = timer.enabled
                timer.enabled = False
                try:
                    return orig_ab(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, *rest, **kw)
                finally:
                    timer.enabled = en
            timer.enabled = False
            s.record()
            r = orig_ab(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, *rest, **kw)
            e.record()
            timer.enabled = True
            timer.records.append(("attn_bwd_self" if ql is None else "attn_bwd_cross", s, e, ATTN_BWD_FLOP_PER_PAIR_HEAD * nq * pairs, 0.0))
            return r

        orig_afs, orig_abs = ops.attn_fwd_split, ops.attn_bwd_split

        def timed_attn_fwd_split(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, *rest, **kw):
            pairs = timer.pairs["self" if ql is None else "cross"]
            timer._next = ("attn_fwd_self" if ql is None else "attn_fwd_cross", 4.0 * 64 * nq * pairs)
            return orig_afs(q, ldq, k, ldk, v, ldv, kl, ql, row_empty, B, S, nq, *rest, **kw)

        def timed_attn_bwd_split(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, row_empty, tile_empty, B, S, nq, *rest, **kw):
            label = "attn_bwd_self" if ql is None else "attn_bwd_cross"
            pairs = timer.pairs["self" if ql is None else "cross"]
            if not (timer.enabled and (timer.only is None or label in timer.only)):
                en = timer.enabled
                timer.enabled = False
                try:
                    return orig_abs(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, row_empty, tile_empty, B, S, nq, *rest, **kw)
                finally:
                    timer.enabled = en
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            timer.enabled = False
            s_.record()
            r = orig_abs(q, ldq, k, ldk, v, ldv, o, d_o, lse, kl, ql, row_empty, tile_empty, B, S, nq, *rest, **kw)
            e_.record()
            timer.enabled = True
            # algorithmic = SURVEY 8(d): backward = 2 x forward = four products (the two-kernel recompute form executes seven)
            timer.records.append((label, s_, e_, ATTN_BWD_FLOP_PER_PAIR_HEAD * nq * pairs, 0.0))
            return r

        ops.attn_fwd_split = timed_attn_fwd_split
        ops.attn_bwd_split = timed_attn_bwd_split
        ops.call = timed_call
        ops.gemm = timed_gemm
        ops.linear_dgrad_t = timed_dgrad_t
        ops.attn_fwd_bf16 = timed_attn_fwd16
        ops.attn_bwd_bf16 = timed_attn_bwd16
        ops.attn_fwd = timed_attn_fwd
        ops.attn_bwd = timed_attn_bwd

    def summary(self, steps: int):
        agg = {}
        for label, s, e, flops, nbytes in self.records:
            d = agg.setdefault(label, dict(ms=0.0, flops=0.0, launches=0, bytes=0.0))
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
            d["launches"] += 1
        out = []
        for label, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            row = dict(kernel=label, ms_per_step=d["ms"] / steps, launches_per_step=d["launches"] / steps,
                       avg_launch_ms=d["ms"] / d["launches"])
            if d["flops"] > 0:
                row["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12
            if d["bytes"] > 0:
                row["algorithmic_GBps"] = d["bytes"] / (d["ms"] * 1e-3) / 1e9      # operands read once + result written
            out.append(row)
        return out


def _kernel_choice_counters() -> dict:
    """Launch counters of the kernels that replace the 128 x 128 GEMM on their shapes (diagnostic exports of the library)."""
    import ctypes
    from gamer_amd import _lib
    lib, out = _lib.load(), {}
    for name in ("gemm_as", "gemm_os", "gemm_wg"):
        fn = getattr(lib, f"gamer_debug_{name}_launches")
        fn.restype, fn.argtypes = ctypes.c_longlong, []
        out[name] = int(fn())
    return out


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def log(msg: str):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def committed_traffic(kernel_substr, want: dict, per_call: bool = False):
    """HBM bytes per launch of a kernel - or of a family of kernels (a tuple of name fragments: the weight gradients run on
    gemm_wg_kernel where dW is whole 256 x 256 tiles and on gemm_f32_kernel elsewhere), averaged over their dispatches -
    from the committed rocprofv3 PMC summaries (profiles/<tag>_pmc_{FETCH,WRITE}_SIZE.csv;
    separate --pmc passes).  Only a profile set whose sidecar ``profiles/<tag>_pmc_meta.json`` (written by the profiling
    script next to the CSVs: the bench arguments of the profiled command) matches the CURRENT run's workload in every
    key of ``want`` (batch, items, dtype, matmul, variant, ragged) is used; the newest matching set wins.  No match ->
    None (``traffic: null``) rather than a figure from another shape.  gfx950 correction from MI355X_MICROARCH.md:
    FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled; WRITE_SIZE is exact; both in KiB."""
    import csv
    import glob
    prof = os.path.join(ROOT, "profiles")

    import re
    if isinstance(kernel_substr, re.Pattern):
        def hit(name):
            return kernel_substr.search(name) is not None
    else:
        subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)

        def hit(name):
            return any(sub in name for sub in subs)

    # per_call: the family is SEVERAL kernels launched once each per call of its entry point (attention backward: dQ + dK/dV) -
    # bytes per call = the sum over its kernels, calls = the largest dispatch count among them
    def avg(path, col):
        tot, n, most = 0.0, 0.0, 0.0
        for r in csv.DictReader(open(path)):
            if hit(r["kernel"]):
                d = float(r.get("dispatches", 1) or 1)
                tot += float(r[col]) * d
                n += d
                most = max(most, d)
        if per_call:
            return tot / most if most else None
        return tot / n if n else None

    # newest first, by what the profiling script RECORDED in the sidecar: `created` (UTC time stamp) where present, else `seq`;
    # sets without either sort before all others, by name (file times are meaningless in a fresh checkout, and tags such as
    # r04final / r04u do not sort by age)
    metas = []
    for meta_path in glob.glob(os.path.join(prof, "*_pmc_meta.json")):
        try:
            metas.append((json.load(open(meta_path)), meta_path))
        except (OSError, ValueError):
            continue
    metas.sort(key=lambda mp: (str(mp[0].get("created", "")), float(mp[0].get("seq", 0) or 0), os.path.basename(mp[1])), reverse=True)
    for meta, meta_path in metas:
        if any(meta.get(k) != v for k, v in want.items()):
            continue
        tag = meta_path[:-len("_pmc_meta.json")]
        fetch, write = tag + "_pmc_FETCH_SIZE.csv", tag + "_pmc_WRITE_SIZE.csv"
        if not (os.path.exists(fetch) and os.path.exists(write)):
            continue
        f = avg(fetch, "avg_FETCH_SIZE_KB_per_dispatch")
        w = avg(write, "avg_WRITE_SIZE_KB_per_dispatch")
        if f is None or w is None:
            continue
        return {"bytes_per_launch": (2.0 * f + w) * 1024.0, "source": os.path.basename(fetch) + " + " + os.path.basename(write),
                "fetch_kib_raw": f, "write_kib": w, "profiled_command": meta.get("cmd")}
    return None


def cpu_baseline(cfg_dict, seq_items: int, micro_batch: int = 32, timed_steps: int = 3, session_mean=None):
    """The CPU oracle's train step (fwd + bwd + clip + AdamW, dropout on) on this host's cores."""
    from oracle import qwen3multi_oracle as orc
    from gamer_amd import synthetic
    cores = usable_cpus()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
    ocfg = orc.OracleConfig.from_dict(cfg_dict)
    sd = orc.init_state_dict(ocfg, seed=0)
    params = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(x) for k, x in sd.items()}
    times = []
    for step in range(1 + timed_steps):
        batch = synthetic.make_batch(micro_batch, seq_items, 256, 3, seed=100 + step, behavior_probs=[0.7, 0.25, 0.05],
                                     session_mean=session_mean)
        t0 = time.perf_counter()
        _, grads, _ = orc.loss_and_grads(params, ocfg, batch, temperature=0.7, training=True,
                                         session=session_mean is not None)
        orc.clip_and_adamw(params, grads, m, v, step=step + 1, lr=5e-4)
        times.append(time.perf_counter() - t0)
        log(f"cpu baseline step {step}: {times[-1]:.2f} s")
        if step >= 1 and sum(times) > 150.0:       # a slow host: keep the default run within a few minutes
            break
    timed_steps = len(times) - 1
    mean = sum(times[1:]) / len(times[1:])
    quota = None
    try:
        q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q_ == "max" else float(q_) / float(p_)
    except (OSError, ValueError):
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    return dict(value=micro_batch / mean, unit="sequences/s", cores=cores, kind="port",
                # BASELINE.md section 3 asks for N = os.cpu_count(); this box reports more logical CPUs than the process may use:
                # threads = min(affinity mask, cgroup CPU quota), all three stated here
                os_cpu_count=os.cpu_count(), affinity_cpus=affinity, cgroup_quota_cpus=quota,
                sample=f"oracle/qwen3multi_oracle.py train step (fwd+bwd+clip+AdamW, dropout 0.2), micro-batch "
                       f"{micro_batch} x {seq_items * 5} tokens, 1 warm-up + {timed_steps} timed steps, "
                       f"torch.set_num_threads({cores})")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--global-batch", type=int, default=1024,
                    help="sequences per optimizer step over ALL GPUs (north star: 1024); per-GPU batch = global / N")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: every GPU takes --batch sequences (BASELINE configs[2]: 8 x 1024)")
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU micro-batch; implies --weak (default: --global-batch / --gpus)")
    ap.add_argument("--items", type=int, default=101, help="items per sequence (max_his_len + 1)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (headline); bf16 = the AMP variant of configs[2]")
    ap.add_argument("--path", choices=["engine", "module", "module-fused"], default="engine",
                    help="engine = Engine.train_step (the fused fast path, headline); module = the nn.Module plug-in surface "
                         "driven the way HF Trainer drives the reference (forward -> loss.backward() -> clip_grad_norm_ -> "
                         "torch AdamW -> zero_grad); module-fused = the same with the module's FusedClipAdamW optimizer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (bf16, split6, per-GPU batch 128) the default single-GPU run adds")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--matmul", choices=["f32", "split3", "split6", "split9"], default=None,
                    help="how fp32 matrix products are formed (--dtype f32 only; default split3, the engine's default): "
                         "split3 = every fp32 product from a two-way fp16 cut of both operands scaled per tensor, 3 piece products; "
                         "split6 / split9 = every fp32 product from an exact three-way bf16 cut of both operands, 6 / 9 piece "
                         "products on the bf16 pipe, fp32 accumulation (gamer_gemm_f32_split, gamer_attn_*_split; error against "
                         "fp64 at the fp32 MFMA's level, docs/DESIGN_rounds1-4.md section 13); f32 = v_mfma_f32_32x32x2_f32 throughout")
    ap.add_argument("--kernel-rows", type=int, default=12, help="rows of the per-family kernel table in the JSON line")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-launch HIP events (and the roofline object)")
    ap.add_argument("--ragged", action="store_true", help="secondary workload: n_items ~ U{2..items}, right padded")
    ap.add_argument("--accum", type=int, default=1,
                    help="gradient accumulation: one step = ACCUM micro-batches of --batch through Engine.train_window (one clip + "
                         "AdamW per window); BASELINE configs[3] is --batch 128 --accum 4 --augment 4")
    ap.add_argument("--deterministic", action="store_true",
                    help="Engine(deterministic=True): every reduction of the step in a fixed order whatever the environment says - the "
                         "default of the fp32 forms anyway; with --dtype bf16 the weight-gradient GEMMs take the two-pass form too")
    ap.add_argument("--augment", type=int, default=0,
                    help="secondary workload: rows shaped like tasks=smb_explicit_decoder_<AUGMENT> (thinned copies, cropped to "
                         "--items, right padded per micro-batch; synthetic.make_augmented_batch)")
    ap.add_argument("--variant", choices=["multi", "session"], default="multi",
                    help="secondary workload: 'session' = Qwen3SessionMulti (session-wise masks, same weights)")
    ap.add_argument("--session-mean", type=float, default=4.0, help="items per session for --variant session")
    args = ap.parse_args(argv)
    if args.matmul is None:
        args.matmul = "split3" if args.dtype == "f32" else "f32"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.batch is not None:
        args.weak = True
    elif args.weak:
        args.batch = 1024
    else:
        if args.global_batch % args.gpus != 0:
            ap.error(f"--global-batch {args.global_batch} is not divisible by --gpus {args.gpus}")
        args.batch = args.global_batch // args.gpus
    return args


def launcher_command(n: int, port: int, argv) -> list:
    """One rank per GPU on this node, as ref:scripts/train_SMB_decoder.sh:123-153 does with torchrun.
    GAMER_BENCH_LAUNCHER (a command prefix, shlex-split) replaces the torch launcher in tests."""
    import shlex
    override = os.environ.get("GAMER_BENCH_LAUNCHER")
    if override:
        return shlex.split(override) + [str(n), os.path.abspath(__file__), *argv]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def spawn_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes and relay rank 0's
    JSON line.  Runs before anything in this process has touched the GPU (device_count() does not initialise it);
    the children are new processes, never an exec of this one."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("GAMER_BENCH_LAUNCHER"):
        log(f"--gpus {args.gpus} but only {have} HIP device(s) are visible")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = launcher_command(args.gpus, port, argv)
    log("spawning: " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if r.returncode != 0:
        log(f"launcher exited with {r.returncode}")
        return r.returncode
    if line is None:
        log("no JSON line came back from rank 0")
        return 3
    got = json.loads(line).get("n_gpus")
    if got != args.gpus:
        log(f"asked for {args.gpus} ranks, the run reports n_gpus={got}")
        return 4
    print(line, flush=True)
    return 0


SECONDARY_LEGS = (
    # (name, dtype, matmul, per-GPU batch): what DESIGN.md / README.md quote beside the headline, timed by the SAME driver run
    ("f32_split6_b1024", "f32", "split6", 1024),   # Linear products as six bf16 piece products (the exact three-way cut)
    ("f32_mfma_b1024", "f32", "f32", 1024),        # the same fp32 step with every product on v_mfma_f32_32x32x2_f32
    ("bf16_b1024", "bf16", "f32", 1024),           # BASELINE configs[2]'s per-GPU shape (bf16 AMP)
    ("f32_split3_b128", "f32", "split3", 128),     # the headline form at the per-GPU batch of the 8-GPU north-star point
    ("f32_mfma_b128", "f32", "f32", 128),
    ("bf16_b128", "bf16", "f32", 128),
    # BASELINE configs[3]: per-GPU micro-batch 128 x gradient accumulation 4 on 4x-augmentation-shaped ragged rows (one step = one
    # optimizer step over 512 sequences)
    ("f32_split3_b128_accum4_ragged", "f32", "split3", 128, dict(accum=4, augment=4)),
    ("bf16_b128_accum4_ragged", "bf16", "f32", 128, dict(accum=4, augment=4)),
    # the Qwen3SessionMulti variant's train step (sessions of 4 items on average), default product form
    ("session_split3_b1024", "f32", "split3", 1024, dict(variant="session")),
)


def decode_leg(users: int = 256, beams: int = 20, his: int = 100, catalogue: int = 20000, iters: int = 4, cpu_users: int = 1):
    """BASELINE configs[4]'s shape on one GPU (ref:SeqRec/tasks/test_SMB_decoder.py:141-285): trie-
    constrained beam search of `users` users x `beams` beams over a 100-item history, 4 new tokens, shipped architecture, default
    product form.  Reports users/s, the prompt pass and the cached per-token step apart (a run with one new token = prompt pass +
    first selection; the three cached steps are the rest), the per-token step against the HBM roofline (algorithmic bytes: the
    prompt K / V of every attention call once per user + the fp32 parameters once), and the CPU decode oracle on `cpu_users`
    user(s) as the baseline (baseline only: a bounded sample)."""
    from gamer_amd import synthetic
    from gamer_amd.config import synthetic_config
    from gamer_amd.decode import ItemTrie, beam_search
    from gamer_amd.engine import Engine
    cfg = synthetic_config()
    eng = Engine(cfg, temperature=0.7)
    eng.init_weights(seed=0)
    cat = synthetic.make_catalogue(catalogue, 256, seed=3)
    tb = 2
    items = synthetic.item_tokens(cat, tb, 256).tolist()
    trie = ItemTrie(items)
    batch = synthetic.make_eval_batch(users, his, cat, tb, 256, 3, min_his=his, seed=5, behavior_probs=[0.7, 0.25, 0.05])

    def timed(new_tokens):
        run = lambda: beam_search(eng, batch["input_ids"], batch["attention_mask"], batch["actions"], trie, beams, new_tokens)
        # three untimed runs: an evaluation decodes many batches of one shape (with GAMER_DECODE_GRAPH=1 the cached per-token step
        # is a hipGraph replay from the third one on - gamer_amd/decode.py: _DecodeStatic) - the steady state is what is timed
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters, out
    ms_all, (seqs, scores) = timed(4)
    ms_first, _ = timed(1)
    per_token_ms = max(ms_all - ms_first, 1e-6) / 3.0
    L0 = batch["input_ids"].shape[1]
    n_attn = cfg.num_hidden_layers + len(cfg.cross_attention_decoder)
    kv_bytes = users * n_attn * L0 * cfg.num_key_value_heads * cfg.head_dim * 4 * 2          # prompt K and V, once per user and call
    param_bytes = sum(int(v.numel()) for v in eng.params.values()) * 4
    step_bytes = kv_bytes + param_bytes
    out = {"name": f"decode_bs{users}_beams{beams}", "metric": "evaluation users/s, Qwen3Multi SMB decoder, trie-constrained beam search",
           "value": users / (ms_all * 1e-3), "unit": "users/s", "ms_per_batch": ms_all, "prefill_plus_first_token_ms": ms_first,
           "per_token_step_ms": per_token_ms,
           "per_token_roofline": {"bound": "hbm", "algorithmic_bytes": step_bytes, "achieved": step_bytes / (per_token_ms * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / (per_token_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "workload": f"{users} users x {beams} beams, history {his} items (prompts of {L0} tokens), 4 new tokens, catalogue "
                       f"{catalogue} items, fp32 tensors (split3 products), K/V cache: prompt once per user, generated positions per beam"}
    if cpu_users > 0:
        from oracle import decode_oracle as dec, qwen3multi_oracle as orc
        ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
        sd = {k: v.detach().cpu().clone() for k, v in eng.params.items()}
        cores = usable_cpus()
        torch.set_num_threads(cores)
        otrie = dec.ItemTrie(items)
        t0 = time.perf_counter()
        oseq, osc = dec.beam_search(sd, ocfg, batch["input_ids"][:cpu_users], batch["attention_mask"][:cpu_users],
                                    batch["actions"][:cpu_users], otrie, beams, 4)
        cdt = time.perf_counter() - t0
        n = cpu_users * beams
        out["cpu_baseline"] = {"value": cpu_users / cdt, "unit": "users/s", "cores": cores, "kind": "port",
                               "sample": f"oracle/decode_oracle.py beam search on {cpu_users} of the {users} users ({cdt:.1f} s)",
                               "sequences_equal_gpu": bool(torch.equal(oseq, seqs[:n].cpu())),
                               "max_score_diff": float((osc - scores[:n].cpu()).abs().max())}
    del eng
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def run_leg(args, rank, world, local_rank, force_dist, timer):
    """One measurement: W untimed steps, K timed steps between barrier + synchronize (MAX over ranks).  Returns the result
    dict on rank 0 (None elsewhere).  Frees its engine before returning."""
    from gamer_amd import synthetic
    from gamer_amd.config import synthetic_config
    from gamer_amd.dp import GradAllReducer
    from gamer_amd.engine import Engine

    cfg = synthetic_config()
    if args.no_dropout:
        cfg.dropout_rate = 0.0
        cfg.attention_dropout = 0.0
    if args.matmul != "f32" and args.dtype != "f32":
        raise SystemExit("--matmul split3/split6/split9 is a form of the fp32 path (--dtype f32)")
    eng = Engine(cfg, device=f"cuda:{local_rank}", temperature=0.7, variant=args.variant, dtype=args.dtype,
                 matmul=args.matmul, deterministic=bool(getattr(args, "deterministic", False)) or None)
    smean = args.session_mean if args.variant == "session" else None
    eng.init_weights(seed=0)                     # identical replicas on every rank
    eng.base_seed = 0x5EED + rank                # independent dropout streams per rank
    reducer = GradAllReducer(eng.flat_g, eng.layout, cfg.num_hidden_layers) if (world > 1 or force_dist) else None
    if reducer is not None and force_dist:
        reducer.world = 2                        # take the collective code path even with one rank

    accum = max(1, int(args.accum))

    def make_cpu(seed):
        if args.augment:
            return synthetic.make_augmented_batch(args.batch, args.items, 256, 3, augment=args.augment, seed=seed,
                                                  behavior_probs=[0.7, 0.25, 0.05])
        return synthetic.make_batch(args.batch, args.items, 256, 3, ragged=args.ragged, seed=seed,
                                    behavior_probs=[0.7, 0.25, 0.05], session_mean=smean)

    # inputs are resident in HBM before the timed region starts; with --accum a "batch" is a window of ACCUM micro-batches
    n_batches = min(args.steps + args.warmup, 4)
    cpu_windows = [[make_cpu(20251114 + 1000 * rank + 16 * s + a) for a in range(accum)] for s in range(n_batches)]
    windows = [[{k: v.cuda(non_blocking=True) for k, v in b.items()} for b in w] for w in cpu_windows]
    batches = [w[0] for w in windows]
    flops = []
    for w in cpu_windows:
        fs = [algorithmic_flops(cfg, b, session=args.variant == "session") for b in w]
        flops.append({k: sum(f[k] for f in fs) for k in ("fwd", "step", "tokens", "p_self", "p_cross")})
    window_labels = [float(sum(int((b["labels"][:, 1:] != -100).sum()) for b in w)) for w in cpu_windows]   # HF's num_items_in_batch
    if accum > 1 and (args.path != "engine" or args.variant != "multi"):
        raise SystemExit("--accum is measured on the engine path (Engine.train_window) of the Qwen3Multi variant")
    timer.reset()
    lr = 5e-4
    grad_scale = 1.0                              # gradients are already global means (sum CE / global count)

    module = opt = None
    if args.path != "engine":
        if world > 1:
            raise SystemExit("--path module is a single-GPU measurement (HF Trainer would wrap the module in DDP)")
        from gamer_amd.modeling import Qwen3MultiWithTemperature, Qwen3SessionMultiWithTemperature
        cls = Qwen3MultiWithTemperature if args.variant == "multi" else Qwen3SessionMultiWithTemperature
        del eng
        torch.cuda.empty_cache()
        module = cls(cfg, device=f"cuda:{local_rank}", dtype=args.dtype, matmul=args.matmul)
        module.set_hyper(0.7)
        module.zero_copy_logits = True      # the loop below reads only the loss (HF Trainer.training_step does the same)
        module.train()
        eng = module.engine
        eng.base_seed = 0x5EED + rank
        if args.path == "module-fused":
            opt = module.fused_optimizer(lr=lr, weight_decay=0.01, max_grad_norm=1.0)
        else:
            # HF Trainer's parameter groups: no weight decay on the norm weights (get_decay_parameter_names)
            decay = [p for n, p in module.named_parameters() if not n.endswith("norm.weight")]
            nodecay = [p for n, p in module.named_parameters() if n.endswith("norm.weight")]
            opt = torch.optim.AdamW([dict(params=decay, weight_decay=0.01), dict(params=nodecay, weight_decay=0.0)], lr=lr,
                                    betas=(0.9, 0.999), eps=1e-8)

    def step(i):
        f = flops[i % n_batches]
        timer.pairs = {"self": f["p_self"], "cross": f["p_cross"]}
        if module is None and accum > 1:
            # (the per-family kernel table's attention TFLOP/s use the window's pair count / ACCUM per launch)
            timer.pairs = {"self": f["p_self"] / accum, "cross": f["p_cross"] / accum}
            return eng.train_window(windows[i % n_batches], lr, n_items=window_labels[i % n_batches], reducer=reducer)[-1]
        if module is None:
            return eng.train_step(batches[i % n_batches], lr, reducer=reducer, grad_scale=grad_scale)
        b = batches[i % n_batches]
        out = module(input_ids=b["input_ids"], attention_mask=b["attention_mask"], actions=b["actions"], labels=b["labels"],
                     session_ids=b.get("session_ids"), extended_session_ids=b.get("extended_session_ids"), split="train")
        out.loss.backward()
        if args.path == "module":
            torch.nn.utils.clip_grad_norm_(module.parameters(), 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return out.loss.detach()

    def barrier():
        if world > 1 or force_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Per-launch HIP events cost ~2.3 ms per step (an event record is a barrier packet between two kernels: measured at
    # per-GPU batch 128, 54.1 vs 51.6 ms fp32, 20.6 vs 18.3 ms bf16), so every launch is timed only in the WARM-UP steps
    # after the first (-> the "kernels" table) and the timed region carries events on the dominant kernel family alone
    # (-> "roofline", measured live in the timed region as the contract asks, ~100 events per step).
    log(f"engine ready ({args.dtype}/{args.matmul}, per-GPU batch {args.batch}), {n_batches} batches resident; warm-up")
    warm_kernels, warm_steps = [], 0
    for i in range(args.warmup):
        if i == 1 and not args.no_kernel_timing:
            timer.enabled = True
        loss = step(i)
    barrier()
    if timer.enabled:
        warm_steps = args.warmup - 1
        warm_kernels = timer.summary(warm_steps)
        # the timed region carries events on TWO families: the one with the largest time per step among every family with a
        # FLOP model (GEMMs and attention alike -> "roofline") and the largest GEMM family (-> "roofline.gemm")
        modelled = [k for k in warm_kernels if "tflops" in k]
        gw = [k for k in modelled if k["kernel"].startswith("gemm")]
        timer.only = {max(fam, key=lambda k: k["ms_per_step"])["kernel"] for fam in (modelled, gw) if fam}
        timer.records = []
    log("timed region")
    timer.enabled = not args.no_kernel_timing
    choice0 = _kernel_choice_counters()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    choice1 = _kernel_choice_counters()
    timer.enabled = False
    final_loss = float(loss)
    log(f"timed region done: {elapsed / args.steps * 1e3:.1f} ms/step")
    eng.check_inputs()
    if world > 1 or force_dist:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the collective on its own, outside the timed region: the whole flat gradient (98 MB fp32) in the per-layer
    # buckets of the train step, back to back; bus bandwidth = 2 (N-1)/N x bytes / time (ring-equivalent convention)
    allreduce = None
    if reducer is not None and world > 1:
        import torch.distributed as dist
        reps = 5
        for timed in (False, True):
            dist.barrier()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(reps if timed else 2):
                for l in reversed(range(cfg.num_hidden_layers)):
                    reducer.layer_done(l)
                reducer.finish()
            torch.cuda.synchronize()
            tb = time.perf_counter() - ta
        t = torch.tensor([tb], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nbytes = eng.flat_g.numel() * 4
        ms = float(t.item()) / reps * 1e3
        allreduce = {"ranks": dist.get_world_size(), "backend": "nccl (RCCL)", "bytes": nbytes, "buckets": cfg.num_hidden_layers + 2,
                     "ms": ms, "bus_GBps": 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9}

    # N > 1 (or the forced single-rank collective path): what the reducer costs and hides.  (a) every bucket alone; (b) the same
    # step WITHOUT any collective (local gradients only - not a valid training step, a timing reference): with - without = the
    # communication the backward did not hide.
    comm_diag = None
    if reducer is not None and (world > 1 or force_dist) and accum == 1 and module is None:
        import torch.distributed as dist
        from gamer_amd.dp import bucket_timings
        buckets = bucket_timings(reducer)
        n_ref = max(2, min(args.steps, 4))

        def timed_steps(red):
            barrier()
            ta = time.perf_counter()
            for i in range(n_ref):
                eng.train_step(batches[i % n_batches], lr, reducer=red, grad_scale=grad_scale)
            barrier()
            t = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) / n_ref * 1e3
        with_ms = timed_steps(reducer)
        without_ms = timed_steps(None)
        comm_diag = {"per_bucket": buckets, "sum_of_buckets_ms": sum(b["ms"] for b in buckets),
                     "step_ms_with_reducer": with_ms, "step_ms_without_collectives": without_ms,
                     "exposed_communication_ms": max(0.0, with_ms - without_ms), "steps_each": n_ref}

    result = None
    if rank == 0:
        seqs = args.batch * accum * world * args.steps
        step_flops = sum(flops[i % n_batches]["step"] for i in range(args.warmup, args.warmup + args.steps)) / args.steps
        timed = timer.summary(args.steps)                      # the dominant family, measured in the timed region
        kernels = warm_kernels if warm_kernels else timed      # every launch, measured in the warm-up steps
        gemm_rows = [k for k in kernels if k["kernel"].startswith("gemm")]
        timed_gemm = [k for k in timed if k["kernel"].startswith("gemm")]
        gdom = timed_gemm[0] if timed_gemm else (max(gemm_rows, key=lambda k: k["ms_per_step"]) if gemm_rows else None)
        modelled = [k for k in (timed if timed else kernels) if "tflops" in k]
        dom = max(modelled, key=lambda k: k["ms_per_step"]) if modelled else None
        gemm_ms = sum(k["ms_per_step"] for k in gemm_rows)
        gemm_tf = sum(k["tflops"] * k["ms_per_step"] for k in gemm_rows) / max(gemm_ms, 1e-9)
        ms_per_step = elapsed / args.steps * 1e3
        split_terms = {"f32": 0, "split3": 3, "split6": 6, "split9": 9}[args.matmul]
        peak = FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else BF16_MATRIX_PEAK_TFLOPS
        if split_terms:
            # every fp32 product is `terms` 16-bit MFMA products: the pipe that bounds the kernel is the 16-bit one, and its
            # peak in units of the ALGORITHMIC (fp32 problem) FLOPs is the 16-bit peak / terms
            peak = BF16_MATRIX_PEAK_TFLOPS / split_terms
        want = dict(batch=args.batch, items=args.items, dtype=args.dtype, matmul=args.matmul, variant=args.variant,
                    ragged=bool(args.ragged))

        def family_roofline(row):
            """`roofline` object of one kernel family (a row of the timed region's table): algorithmic FLOPs of its launches /
            their HIP-event time against the matrix peak of the form, HBM bytes per launch (per call for the attention backward:
            its kernels summed) from the newest committed PMC profile of this workload."""
            if row is None:
                return None
            fam = row["kernel"]
            if fam.startswith("gemm"):
                # (substring match: the template list ends with the matmul form, ", 0," = fp32 MFMA, then the ping-pong flag)
                kname = {"gemm_fwd": f"gemm_f32_kernel<true, true, 0, false, false, 2, 0, {split_terms},",
                         "gemm_dgrad": f"gemm_f32_kernel<true, false, 0, false, false, 2, 0, {split_terms},",
                         "gemm_wgrad": (f"gemm_f32_kernel<false, false, 1, false, false, 2, 0, {split_terms},", "gemm_wg_kernel<")}.get(fam)
                if args.dtype != "f32":
                    kname = None
                traffic = committed_traffic(kname, want) if kname else None
                label = ((f"gemm_wg_kernel + gemm_f32_kernel ({fam}: 256 x 256 tiles where dW is whole tiles, 128 x 128 elsewhere)"
                          if (fam == "gemm_wgrad" and split_terms == 3) else
                          f"gemm_{'f32' if args.dtype == 'f32' else 'bf16'}_kernel ({fam})"))
                note = None
            else:
                # attention: every form's kernels carry CROSS as their LAST template argument; the backward is two kernels per call
                import re
                stem = "attn_bwd_" if fam.startswith("attn_bwd") else "attn_fwd_"
                cross = "true" if fam.endswith("cross") else "false"
                traffic = committed_traffic(re.compile(stem + r"\w+<[^>]*\b" + cross + r">\("), want, per_call=True)
                label = (f"{stem}*_kernel<..., CROSS = {cross}> ({fam}: " +
                         ("dQ + dK/dV kernels of one call" if stem == "attn_bwd_" else "one kernel per call") + ")")
                note = ("algorithmic FLOPs per SURVEY.md 8(d): 1536 per allowed (query, key) pair forward, backward = 2 x forward "
                        "(four products; the two-kernel recompute form executes seven)")
            if split_terms:
                label += (f" SPLIT={split_terms}: {'fp16' if split_terms == 3 else 'bf16'} MFMA pipe, peak = "
                          f"{BF16_MATRIX_PEAK_TFLOPS:g} / {split_terms}")
            return {
                "bound": "mfma",
                "kernel": label,
                "achieved": row["tflops"],
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": row["tflops"] / peak,
                # the same launches against the HBM roofline (operands once + result): at K = 256 the bf16 GEMMs sit
                # below the MFMA/HBM ridge, so this is the bound that binds them
                "algorithmic_GBps": row.get("algorithmic_GBps"),
                "hbm_frac": (row["algorithmic_GBps"] / HBM_PEAK_GBS) if row.get("algorithmic_GBps") else None,
                "traffic": traffic["bytes_per_launch"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "traffic_profiled_command": traffic["profiled_command"] if traffic else None,
                "algorithmic_flop_per_launch": row["tflops"] * 1e12 * row["avg_launch_ms"] * 1e-3,
                "avg_launch_ms": row["avg_launch_ms"],
                "ms_per_step": row["ms_per_step"],
                "flop_convention": note,
            }
        roofline = family_roofline(dom)
        if roofline is not None:
            roofline.update({
                "gemm": family_roofline(gdom) if (gdom is not None and gdom is not dom) else None,
                "all_gemm_tflops": gemm_tf,
                "all_gemm_ms_per_step": gemm_ms,
                "step_algorithmic_tflops": step_flops / (ms_per_step * 1e-3) / 1e12,
                "step_frac_of_matrix_peak": step_flops / (ms_per_step * 1e-3) / 1e12 / peak,
                # what the 16-bit matrix pipe SUSTAINS with the whole chip busy depends on the operand data (power): measured
                # 1.43-1.68 PFLOP/s on random bits against the nominal 2.5 (tools/ubench_memtime.hip); `peak` above is nominal
                "pipe_sustained_random_operands_TFLOPs": ([1430.0 / split_terms, 1680.0 / split_terms] if split_terms else None),
                "pipe_sustained_source": "profiles/r03_mfma_sustained.txt" if split_terms else None,
            })
        result = {
            "metric": ("train-step sequences/sec, Qwen3Multi SMB decoder, his_len=100" if args.variant == "multi" else
                       "train-step sequences/sec, Qwen3SessionMulti SMB decoder, his_len=100"),
            "value": seqs / elapsed,
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            # the arithmetic the path computes in: fp32 tensors and accumulators throughout; for the split forms the PRODUCTS are
            # formed on the 16-bit matrix pipe from piece cuts of the fp32 operands - named here, not hidden behind "f32"
            "dtype": ({"f32": "f32", "split3": "f32 tensors/accumulators, products = 3 x fp16 pieces (split3)",
                       "split6": "f32 tensors/accumulators, products = 6 x bf16 pieces (split6, exact cut)",
                       "split9": "f32 tensors/accumulators, products = 9 x bf16 pieces (split9, exact cut)"}[args.matmul]
                      if args.dtype == "f32" else "bf16"),
            "data": "synthetic",
            "config": {
                "workload": ((f"[{args.matmul if args.dtype == 'f32' else 'bf16'}] ") +
                             (f"[accum {accum} x micro-batch {args.batch}] " if accum > 1 else "") +
                             (f"[{args.augment}x-augmentation-shaped ragged rows] " if args.augment else "") +
                             ("Qwen3Multi" if args.variant == "multi" else
                              f"Qwen3SessionMulti (sessions of {args.session_mean:g} items on average)") +
                             " SMB decoder train step (fwd+bwd+clip+AdamW, dropout 0.2), ShortVideoAD-shaped "
                             f"synthetic ids, per-GPU batch {args.batch} x {args.items * 5} tokens"
                             + (" ragged" if args.ragged else "") + ", V=1041, " +
                             ("fp32" if args.dtype == "f32" else "bf16 AMP (fp32 master weights / gradients / moments)") +
                             (" tensors and accumulation; every fp32 matrix product (Linear and attention) formed as 3 fp16 piece "
                              "products of a two-way cut of both operands, each operand tensor scaled by a power of two from its "
                              "largest magnitude (error vs fp64 at or below the fp32 MFMA's)" if split_terms == 3 else
                              f" tensors and accumulation; every fp32 matrix product (Linear and attention) formed as {split_terms} "
                              "exact bf16 piece products of a three-way cut of both operands (error vs fp64 at the fp32 MFMA's "
                              "level)" if split_terms else (" on the fp32 MFMA" if args.dtype == "f32" else ""))),
                "global_batch": args.batch * accum * world,
                "micro_batch": args.batch,
                "grad_accum": accum,
                "seq_len": args.items * 5,
                "parallelism": f"dp{world}",
                "path": args.path,
                "matmul": args.matmul,
                "deterministic_wgrad": bool(getattr(args, "deterministic", False)),
            },
            "roofline": roofline,
            "kernels": kernels[:args.kernel_rows],
            # which of the alternative kernels ran (launches per step inside the timed region) and the switches in effect: A/B numbers
            # are attributable to a kernel set from the line alone
            "kernel_choice": {"launches_per_step": {k: (choice1[k] - choice0[k]) / args.steps for k in choice1},
                              "split_inject": bool(getattr(eng, "split_inject", False)),
                              "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GAMER_")}},
            "kernels_measured_in": (f"warm-up steps 2..{args.warmup} (every launch between HIP events)" if warm_kernels
                                    else "timed region"),
            "loss": final_loss,
        }
        if allreduce is not None:
            result["allreduce"] = allreduce
        if comm_diag is not None:
            result["communication"] = comm_diag
        if args.no_kernel_timing:
            result["roofline"] = None
    # release this leg's engine (86 GB of activations at batch 1024 fp32) before the next leg allocates its own
    del eng, module, opt, batches, reducer
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return result


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("GAMER_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path on one GPU
    rccl_log = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        if "NCCL_DEBUG_FILE" not in os.environ and os.environ.get("NCCL_DEBUG", "").upper() not in ("TRACE",):
            # what RCCL decides (rings / channels / transports, algorithm and protocol per collective) goes into the JSON line:
            # every rank writes its INFO log to a file of its own, rank 0's is parsed after the run (gamer_amd.dp.parse_rccl_log).
            # (An inherited NCCL_DEBUG=WARN / VERSION is raised to INFO for this process; a user's own log file is left alone.)
            import tempfile
            log_dir = tempfile.mkdtemp(prefix="gamer_rccl_")
            rccl_log = os.path.join(log_dir, f"rank{rank}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING,ENV", NCCL_DEBUG_FILE=rccl_log)
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {world}")
        # every rank really joined RCCL on a device of its own: one tiny all-reduce counts them, one all-gather lists the devices
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)
        devs = [None] * world
        dist.all_gather_object(devs, (rank, torch.cuda.current_device(), torch.cuda.get_device_properties(local_rank).name))
        if int(probe.item()) != world or len({d[1] for d in devs}) != (world if torch.cuda.device_count() >= world else 1):
            raise SystemExit(f"RCCL group check failed: sum of ones = {int(probe.item())}, ranks/devices = {devs}")

    timer = KernelTimer()
    if not args.no_kernel_timing:
        timer.install()
    result = run_leg(args, rank, world, local_rank, force_dist, timer)

    # Secondary legs (N = 1, default workload only): the other step forms DESIGN.md quotes - bf16 AMP, fp32 Linear
    # products on the bf16 pipe (split6), and both dtypes at the per-GPU batch of the 8-GPU north-star point (128) -
    # timed by the same run, 2 warm-up + 5 timed steps each, reported under "secondary" on the ONE JSON line.
    default_workload = (world == 1 and not force_dist and args.path == "engine" and args.dtype == "f32" and args.matmul == "split3"
                        and args.batch == 1024 and args.items == 101 and args.variant == "multi" and not args.ragged
                        and not args.no_dropout and args.accum == 1 and not args.augment and not args.deterministic)
    if default_workload and not args.no_secondary and result is not None:
        import copy
        secondary = []
        for leg in SECONDARY_LEGS:
            name, dtype, matmul, batch = leg[:4]
            a = copy.copy(args)
            a.dtype, a.matmul, a.batch, a.steps, a.warmup = dtype, matmul, batch, 5, 2
            for k_, v_ in (leg[4] if len(leg) > 4 else {}).items():
                setattr(a, k_, v_)
            try:
                r = run_leg(a, rank, world, local_rank, force_dist, timer)
            except Exception as e:                                  # a failing extra leg must not cost the headline line
                log(f"secondary leg {name} failed: {e!r}")
                secondary.append({"name": name, "error": repr(e)})
                continue
            rf = r.get("roofline") or {}
            secondary.append({"name": name, "dtype": dtype, "matmul": matmul, "per_gpu_batch": batch, "grad_accum": a.accum,
                              "sequences_per_step": batch * a.accum, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": r["unit"],
                              "loss": r["loss"], "dominant_kernel": rf.get("kernel"), "achieved": rf.get("achieved"),
                              "peak": rf.get("peak"), "frac": rf.get("frac"), "hbm_frac": rf.get("hbm_frac"),
                              "step_algorithmic_tflops": rf.get("step_algorithmic_tflops"),
                              "all_gemm_ms_per_step": rf.get("all_gemm_ms_per_step"),
                              "kernels": [{k: row[k] for k in ("kernel", "ms_per_step", "avg_launch_ms", "tflops") if k in row}
                                          for row in r["kernels"][:8]]})
        result["secondary"] = secondary
        # the evaluation path (BASELINE configs[4]'s shape, one GPU) in the same driver run: under "secondary_eval"
        try:
            timer.reset()                                           # (no per-launch events in the evaluation leg)
            result["secondary_eval"] = [decode_leg(cpu_users=0 if args.no_cpu_baseline else 1)]
        except Exception as e:
            log(f"decode leg failed: {e!r}")
            result["secondary_eval"] = [{"name": "decode_bs256_beams20", "error": repr(e)}]

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            from gamer_amd.config import synthetic_config
            smean = args.session_mean if args.variant == "session" else None
            result["cpu_baseline"] = cpu_baseline(synthetic_config().to_dict(), args.items, session_mean=smean)
        if rccl_log is not None:
            from gamer_amd.dp import parse_rccl_log
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)           # RCCL writes its log through buffered C stdio: push it to the file first
            except Exception:
                pass
            try:
                result["rccl"] = parse_rccl_log(open(rccl_log, errors="replace").read())
            except OSError as e:
                result["rccl"] = {"error": repr(e)}
        elif world > 1 or force_dist:
            result["rccl"] = {"note": "NCCL_DEBUG_FILE / NCCL_DEBUG=TRACE set by the caller: log left where the caller put it"}
        # RCCL prints a version banner through C stdio; push it out first so that the JSON line is last
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
