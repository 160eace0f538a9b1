"""GPU: launch time of the bf16 attention backward (dQ + dK/dV of one call) at batch B x 505, self and cross masks, dropout 0.2:
python tools/time_attn_bf16_bwd.py [B]   (GAMER_LIB_PATH picks a variant build: tools/ablate_attn_bf16_dkv.sh)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
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
out = []
for name, kl, ql, od in (("self", r["kl_self"], None, None), ("cross", r["kl_cross"], r["ql_cross"], (perm, tm, r["empty_cross"]))):
    o = torch.empty(T, nq * 64, device=dev, dtype=bf); lse = torch.empty(B, nq, S, device=dev)
    ops.attn_fwd_bf16(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, B, S, nq, nkv, 0.125, 0.2, 7, o, lse, order=od)
    delta = torch.zeros(B, nq, S, device=dev)
    dq = torch.empty(T, nq * 64, device=dev, dtype=bf); dk = torch.empty(T, nkv * 64, device=dev, dtype=bf)
    dqkv = torch.empty_like(qkv); dv = dqkv[:, (nq + nkv) * 64:]
    bwd = lambda: ops.attn_bwd_bf16(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, B, S, nq, nkv, 0.125, 0.2, 7,
                                    delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od)
    for _ in range(3): bwd()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): bwd()
    e.record(); torch.cuda.synchronize()
    out.append(f"{name} {s.elapsed_time(e) / 10:.3f} ms")
print(os.environ.get("GAMER_LIB_PATH", "default"), "B =", B, "bwd (dQ + dK/dV):", ", ".join(out), flush=True)
