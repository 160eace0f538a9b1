"""Forward / backward times of the three-product (H2) split attention at batch B x 505, self and cross (row order), for the
current build or another build of the library: python tools/time_attn_h2.py [lib.so] [B]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else ""
B = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 256
from gamer_amd import _lib
if lib: _lib.LIB_PATH = os.path.abspath(lib)
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
cfg = synthetic_config(); dev = "cuda"
items, nq, nkv, p = 101, 6, 3, float(os.environ.get("P_DROP", "0.2"))
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
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
out = []
ops.set_f32_matmul("split3")
with ops.amax_reuse(everything=True):          # operand maxima measured once (kernel timing)
    for name, kl, ql, re_, te, od in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], None), ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], order)):
        ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
        tf = timeit(lambda: ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od, h2=True))
        ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od)
        tb = timeit(lambda: ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, h2=True))
        out.append(f"{name}: fwd {tf:.3f} bwd {tb:.3f}")
print(os.path.basename(lib) or "current", f"B={B} p={p}", " | ".join(out))
