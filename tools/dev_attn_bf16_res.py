"""Resident-K/V bf16 attention kernels (csrc/attention_bf16.hip, attn_*_br_kernel) against the tiled ones on the same inputs:
outputs must be bit-identical (same arithmetic in the same order); launch times of both.  python tools/dev_attn_bf16_res.py [B] [fwd|all]"""
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
bf = torch.bfloat16
q = torch.randn(T, nq * 64, device=dev).to(bf); k = torch.randn(T, nkv * 64, device=dev).to(bf)
qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev).to(bf); v = qkv[:, (nq + nkv) * 64:]; do = torch.randn(T, nq * 64, device=dev).to(bf)
n_t = (S + 31) // 32
perm = torch.empty(B, S, dtype=torch.int32, device=dev); tk = torch.empty(B, n_t, dtype=torch.int32, device=dev); tm = torch.empty(B, n_t, dtype=torch.int32, device=dev)
ops.attn_row_order(r["empty_cross"], perm, tk, tm)


def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, kl, ql, od in (("self", r["kl_self"], None, None), ("cross", r["kl_cross"], r["ql_cross"], (perm, tm, r["empty_cross"]))):
    for p in (0.2, 0.0):
        res = {}
        for form in ("0", "1"):
            os.environ["GAMER_ATTN_RES"] = form
            ops.reload_env()          # (the library caches its switches)
            o = torch.full((T, nq * 64), float("nan"), device=dev, dtype=bf); lse = torch.full((B, nq, S), float("nan"), device=dev)
            fwd = lambda: ops.attn_fwd_bf16(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
            fwd(); torch.cuda.synchronize()
            out = {"o": o.clone(), "lse": lse.clone()}
            tf = timeit(fwd)
            tb = 0.0
            if what == "all":
                delta = torch.zeros(B, nq, S, device=dev)
                dq = torch.full((T, nq * 64), float("nan"), device=dev, dtype=bf); dk = torch.full((T, nkv * 64), float("nan"), device=dev, dtype=bf)
                dqkv = torch.full_like(qkv, float("nan")); dv = dqkv[:, (nq + nkv) * 64:]
                bwd = lambda: ops.attn_bwd_bf16(q, nq * 64, k, nkv * 64, v, qkv.shape[1], out["o"], do, out["lse"], kl, ql, B, S, nq, nkv, 0.125, p, 7,
                                                delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od)
                bwd(); torch.cuda.synchronize()
                out.update(dq=dq.clone(), dk=dk.clone(), dv=dv.clone())
                tb = timeit(bwd)
            res[form] = (out, tf, tb)
        a, b_ = res["0"][0], res["1"][0]
        diffs = " ".join(f"{key} {'EQUAL' if torch.equal(a[key], b_[key]) else 'max diff %.3e' % float((a[key].float() - b_[key].float()).abs().max())}" for key in a)
        print(f"B={B} {name} p={p}: tiled fwd {res['0'][1]:.3f} bwd {res['0'][2]:.3f} | resident fwd {res['1'][1]:.3f} bwd {res['1'][2]:.3f} | {diffs}", flush=True)
