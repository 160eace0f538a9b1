"""Resident-K/V attention kernels (csrc/attention_res.hip) against the tiled kernels of csrc/attention_split.hip on the same inputs:
largest differences of o / lse (forward) and dq / dk / dv (backward), and the launch times of both forms.
python tools/dev_attn_res.py [B] [fwd|bwd|all]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
what = sys.argv[2] if len(sys.argv) > 2 else "all"
cfg = synthetic_config(); dev = "cuda"
items, nq, nkv = 101, 6, 3
S = items * 5; T = B * S
batch = synthetic.make_batch(B, items, 256, 3, seed=3, behavior_probs=[0.7, 0.25, 0.05])
r = ops.alloc_router_outputs(B, S, dev)
ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), cfg.behavior_lut().to(dev), 5, 4, 8, r)
torch.manual_seed(0)
q = torch.randn(T, nq * 64, device=dev); k = torch.randn(T, nkv * 64, device=dev)
qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev); v = qkv[:, (nq + nkv) * 64:]; do = torch.randn(T, nq * 64, device=dev)
n_t = (S + 31) // 32
order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev))
ops.attn_row_order(r["empty_cross"], *order)


def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


ops.set_f32_matmul("split3")
with ops.amax_reuse(everything=True):
    for name, kl, ql, re_, te, od in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], None),
                                      ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], order)):
        for p in (0.2, 0.0):
            res = {}
            for form in ("0", "1"):
                os.environ["GAMER_ATTN_RES"] = form
                ops.reload_env()          # (the library caches its switches)
                o = torch.full((T, nq * 64), float("nan"), device=dev); lse = torch.full((B, nq, S), float("nan"), device=dev)
                fwd = lambda: ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od, h2=True)
                fwd(); torch.cuda.synchronize()
                out = {"o": o.clone(), "lse": lse.clone()}
                tf = timeit(fwd) if what in ("all", "fwd") else 0.0
                tb = 0.0
                if what in ("all", "bwd"):
                    delta = torch.zeros(B, nq, S, device=dev)
                    dq = torch.full((T, nq * 64), float("nan"), device=dev); dk = torch.full((T, nkv * 64), float("nan"), device=dev)
                    dqkv = torch.full_like(qkv, float("nan")); dv = dqkv[:, (nq + nkv) * 64:]
                    bwd = lambda: ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], out["o"], do, out["lse"], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7,
                                                     delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, h2=True)
                    bwd(); torch.cuda.synchronize()
                    out.update(dq=dq.clone(), dk=dk.clone(), dv=dv.clone())
                    tb = timeit(bwd)
                res[form] = (out, tf, tb)
            a, b_ = res["0"][0], res["1"][0]
            diffs = " ".join(f"{key} {rel(b_[key], a[key]):.2e}" + ("" if torch.isfinite(b_[key]).all() else " NONFINITE") for key in a)
            print(f"B={B} {name} p={p}: tiled fwd {res['0'][1]:.3f} bwd {res['0'][2]:.3f} | resident fwd {res['1'][1]:.3f} bwd {res['1'][2]:.3f} | {diffs}", flush=True)
