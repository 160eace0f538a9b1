"""Driver of tools/prof_attn_split.sh: six rounds of fp32-MFMA and split attention (forward, backward recompute, backward
with the dS spill) on one problem, self and cross."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
cfg = synthetic_config(); dev = "cuda"
B, items, nq, nkv, p = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 101, 6, 3, 0.2
S = items * 5; T = B * S
batch = synthetic.make_batch(B, items, 256, 3, seed=3, behavior_probs=[0.7, 0.25, 0.05])
r = ops.alloc_router_outputs(B, S, dev)
ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), cfg.behavior_lut().to(dev), 5, 4, 8, r)
q = torch.randn(T, nq * 64, device=dev); k = torch.randn(T, nkv * 64, device=dev)
qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev); v = qkv[:, (nq + nkv) * 64:]; do = torch.randn(T, nq * 64, device=dev)
n_t = (S + 31) // 32
order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev))
ops.attn_row_order(r["empty_cross"], *order)
o = torch.empty(T, nq * 64, device=dev); lse = torch.empty(B, nq, S, device=dev); delta = torch.zeros(B, nq, S, device=dev)
dq = torch.empty(T, nq * 64, device=dev); dk = torch.empty(T, nkv * 64, device=dev); dqkv = torch.empty_like(qkv); dv = dqkv[:, (nq + nkv) * 64:]
dsw = torch.empty(ops.attn_ds_work_numel(B, S, nq), device=dev)
for name, kl, ql, re_, te, od in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], None), ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], order)):
    for it in range(6):
        ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
        ops.attn_bwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, ds_work=dsw)
        ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
        ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od)
        ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, ds_work=dsw)
torch.cuda.synchronize()
