import sys, torch
sys.path.insert(0, "/root/repo")
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
torch.manual_seed(0)
cfg = synthetic_config()
dev = "cuda"
def run(B, items, nq, nkv, p, ragged):
    S = items * 5; T = B * S
    batch = synthetic.make_batch(B, items, 256, 3, ragged=ragged, seed=3, behavior_probs=[0.7, 0.25, 0.05])
    r = ops.alloc_router_outputs(B, S, dev)
    ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), cfg.behavior_lut().to(dev), 5, 4, 8, r)
    q = torch.randn(T, nq * 64, device=dev); k = torch.randn(T, nkv * 64, device=dev)
    qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev); v = qkv[:, (nq + nkv) * 64:]
    n_t = (S + 31) // 32
    order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev))
    ops.attn_row_order(r["empty_cross"], *order)
    for name, kl, ql, re_, te, od in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], None),
                                      ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], order),
                                      ("cross-noord", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], None)):
        o1 = torch.empty(T, nq * 64, device=dev); l1 = torch.empty(B, nq, S, device=dev)
        o2 = torch.full_like(o1, float("nan")); l2 = torch.full_like(l1, float("nan"))
        ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o1, l1, order=od)
        ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o2, l2, order=od)
        torch.cuda.synchronize()
        eo = float((o1 - o2).abs().max() / o1.abs().max()); el = float((l1 - l2).abs().max())
        print(f"B={B} S={S} nq={nq} nkv={nkv} p={p} ragged={ragged} {name}: o rel {eo:.2e} lse abs {el:.2e} nan={bool(torch.isnan(o2).any())}")
for args in ((2, 7, 2, 1, 0.0, True), (3, 14, 6, 3, 0.2, True), (4, 101, 6, 3, 0.2, False), (2, 40, 3, 3, 0.0, True)):
    run(*args)
