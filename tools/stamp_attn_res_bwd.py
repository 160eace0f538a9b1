"""Where a wave of the resident dK/dV kernel (csrc/attention_res.hip, two heads per wave) spends its cycles; diagnostic build:
bash tools/build_variant.sh rstamp attention_res.hip -DRES_STAMP=1; python tools/stamp_attn_res_bwd.py tools/_ab/rstamp.so [B]."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
os.environ["GAMER_ATTN_RES_DQ"] = "0"
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
lib = _lib.load()
lib.gamer_debug_res_stamp.argtypes = [ctypes.c_void_p]
cfg = synthetic_config(); dev = "cuda"
items, nq, nkv, p = 101, 6, 3, 0.2
S = items * 5; T = B * S
batch = synthetic.make_batch(B, items, 256, 3, seed=3, behavior_probs=[0.7, 0.25, 0.05])
r = ops.alloc_router_outputs(B, S, dev)
ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), cfg.behavior_lut().to(dev), 5, 4, 8, r)
q = torch.randn(T, nq * 64, device=dev); k = torch.randn(T, nkv * 64, device=dev)
qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev); v = qkv[:, (nq + nkv) * 64:]; do = torch.randn(T, nq * 64, device=dev)
o = torch.empty(T, nq * 64, device=dev); lse = torch.empty(B, nq, S, device=dev); delta = torch.zeros(B, nq, S, device=dev)
dq = torch.empty(T, nq * 64, device=dev); dk = torch.empty(T, nkv * 64, device=dev); dqkv = torch.empty_like(qkv); dv = dqkv[:, (nq + nkv) * 64:]
n_t = (S + 31) // 32
order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev))
ops.attn_row_order(r["empty_cross"], *order)
names = ["stage block (loads, cut, LDS stores, barrier)", "key-tile prologue (K / V rows, carried sums)", "query-tile loop", "key-tile epilogue", "wait for the stage's slowest wave", "queue / loop control"]
ops.set_f32_matmul("split3")
with ops.amax_reuse(everything=True):
    for name, kl, ql, re_, te, od in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], None), ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], order)):
        ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od, h2=True)
        f = lambda: ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, h2=True)
        f(); f()
        buf = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
        assert lib.gamer_debug_res_stamp(buf.data_ptr()) == 0
        torch.cuda.synchronize()
        f(); torch.cuda.synchronize()
        lib.gamer_debug_res_stamp(None)
        rec = buf.view(-1, 8).cpu().double()
        rec = rec[rec[:, :6].sum(1) > 0]
        tot = rec[:, :6].sum(1)
        steps = rec[:, 6]
        print(f"{name}: {len(rec)} waves, {tot.mean():.0f} stamped counts per wave (min {tot.min():.0f} max {tot.max():.0f}), {steps.mean():.0f} (query tile x 2 heads) steps per wave (min {steps.min():.0f} max {steps.max():.0f})")
        for i, nm in enumerate(names):
            print(f"   {nm:52s} {100 * rec[:, i].sum() / tot.sum():5.1f} %   {rec[:, i].sum() / steps.sum():7.1f} counts per step")
