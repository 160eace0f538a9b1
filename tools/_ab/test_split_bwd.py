import sys, torch
sys.path.insert(0, "/root/repo")
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
torch.manual_seed(0)
cfg = synthetic_config()
dev = "cuda"
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
def run(B, items, nq, nkv, p, ragged, time=False):
    S = items * 5; T = B * S
    batch = synthetic.make_batch(B, items, 256, 3, ragged=ragged, seed=3, behavior_probs=[0.7, 0.25, 0.05])
    r = ops.alloc_router_outputs(B, S, dev)
    ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), cfg.behavior_lut().to(dev), 5, 4, 8, r)
    q = torch.randn(T, nq * 64, device=dev); k = torch.randn(T, nkv * 64, device=dev)
    qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev); v = qkv[:, (nq + nkv) * 64:]
    do = torch.randn(T, nq * 64, device=dev)
    n_t = (S + 31) // 32
    order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev))
    ops.attn_row_order(r["empty_cross"], *order)
    for name, kl, ql, re_, te, od in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], None),
                                      ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], order),
                                      ("cross-noord", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], None)):
        o = torch.empty(T, nq * 64, device=dev); lse = torch.empty(B, nq, S, device=dev)
        ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
        res = []
        for split in (False, True):
            delta = torch.zeros(B, nq, S, device=dev)
            dq = torch.full((T, nq * 64), float("nan"), device=dev); dk = torch.full((T, nkv * 64), float("nan"), device=dev)
            dqkv = torch.full_like(qkv, float("nan")); dv = dqkv[:, (nq + nkv) * 64:]
            if split:
                f = lambda: ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od)
                ff = lambda: ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
            else:
                dsw = torch.empty(ops.attn_ds_work_numel(B, S, nq), device=dev) if time else None
                f = lambda: ops.attn_bwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, ds_work=dsw)
                ff = lambda: ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
            f(); torch.cuda.synchronize()
            res.append((dq.clone(), dk.clone(), dv.clone(), delta.clone()))
            if time: print(f"   {name} {'split' if split else 'f32  '}: fwd {timeit(ff):.3f} ms  bwd {timeit(f):.3f} ms")
        (a, b_, c, d), (a2, b2, c2, d2) = res
        e = lambda x, y: float((x - y).abs().max() / x.abs().max())
        print(f"B={B} S={S} nq={nq} nkv={nkv} p={p} ragged={ragged} {name}: dq {e(a, a2):.2e} dk {e(b_, b2):.2e} dv {e(c, c2):.2e} delta {e(d, d2):.2e} nan={bool(torch.isnan(a2).any() or torch.isnan(b2).any() or torch.isnan(c2).any())}")
for args in ((2, 7, 2, 1, 0.0, True), (3, 14, 6, 3, 0.2, True), (4, 101, 6, 3, 0.2, False), (2, 40, 3, 3, 0.0, True), (3, 60, 6, 3, 0.2, True)):
    run(*args)
run(256, 101, 6, 3, 0.2, False, time=True)
