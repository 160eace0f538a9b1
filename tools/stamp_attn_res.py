"""Where a wave of the resident-K/V attention FORWARD (csrc/attention_res.hip) spends its cycles; diagnostic build with -DRES_STAMP=1:
bash tools/build_variant.sh rstamp attention_res.hip -DRES_STAMP=1; python tools/stamp_attn_res.py tools/_ab/rstamp.so [B].
s_memtime stamps around the phases of every wave, summed over its row tiles; read the SHARES, not the build's run time."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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
qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev); v = qkv[:, (nq + nkv) * 64:]
o = torch.empty(T, nq * 64, device=dev); lse = torch.empty(B, nq, S, device=dev)
n_t = (S + 31) // 32
order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev))
ops.attn_row_order(r["empty_cross"], *order)
names = ["stage block (loads, cut, LDS stores, barrier)", "row-tile prologue (q, row data, carried state)", "key loop", "row-tile epilogue", "wait for the block's slowest wave", "loop control / skipped tiles"]
ops.set_f32_matmul("split3")
with ops.amax_reuse(everything=True):
    for name, kl, ql, re_, od in (("self", r["kl_self"], None, r["empty_self"], None), ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], order)):
        f = lambda: ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, B, S, nq, nkv, 0.125, p, 7, o, lse, order=od, h2=True)
        f(); f()
        buf = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
        assert lib.gamer_debug_res_stamp(buf.data_ptr()) == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        lib.gamer_debug_res_stamp(None)
        rec = buf.view(-1, 8).cpu().double()
        rec = rec[rec[:, :6].sum(1) > 0]
        tot = rec[:, :6].sum(1)
        tiles = rec[:, 6]
        print(f"{name}: {s.elapsed_time(e):.3f} ms, {len(rec)} waves, {tot.mean():.0f} stamped counts per wave (min {tot.min():.0f} max {tot.max():.0f}), {tiles.mean():.0f} key tiles per wave (min {tiles.min():.0f} max {tiles.max():.0f})")
        for i, nm in enumerate(names):
            print(f"   {nm:52s} {100 * rec[:, i].sum() / tot.sum():5.1f} %   {rec[:, i].sum() / tiles.sum():7.1f} counts per key tile")
        # spread of the key-loop share between the eight waves of a workgroup
        per_wave = rec[:, 2].view(-1, 8) if len(rec) % 8 == 0 else None
        if per_wave is not None:
            print("   key-loop counts by wave index (mean over workgroups):", " ".join(f"{x:.0f}" for x in per_wave.mean(0).tolist()))
            pw4 = rec[:, 4].view(-1, 8)
            print("   wait-for-slowest counts by wave index:               ", " ".join(f"{x:.0f}" for x in pw4.mean(0).tolist()))
