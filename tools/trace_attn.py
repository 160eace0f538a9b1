#!/usr/bin/env python3
"""Residency study of the attention forward kernel: per-workgroup start/end/HW_ID trace."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops, synthetic, _lib
from gamer_amd.config import synthetic_config
import numpy as np

B, items = 256, 101
S, nq, nkv = items * 5, 6, 3
cfg = synthetic_config()
lib = _lib.load()
batch = synthetic.make_batch(B, items, 256, 3, seed=1, behavior_probs=[0.7, 0.25, 0.05])
r = ops.alloc_router_outputs(B, S, "cuda")
ops.router_fwd(batch["input_ids"].cuda(), batch["attention_mask"].cuda(), batch["actions"].cuda(), cfg.behavior_lut().cuda(), 5, 4, 8, r)
T = B * S
q = torch.randn(T, nq * 64, device="cuda"); k = torch.randn(T, nkv * 64, device="cuda")
qkv = torch.randn(T, 768, device="cuda"); v = qkv[:, 576:]
o = torch.empty(T, 384, device="cuda"); lse = torch.empty(B, nq, S, device="cuda")
nblk = ((B * nkv + 7) // 8) * 8 * 8
trace = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
f = lambda: ops.attn_fwd(q, 384, k, 192, v, 768, r["kl_self"], None, r["empty_self"], r["tile_empty_self"], B, S, nq, nkv, 0.125, 0.2, 7, o, lse)
for _ in range(2): f()
torch.cuda.synchronize()
lib.gamer_debug_set_trace.argtypes = [ctypes.c_void_p]
assert lib.gamer_debug_set_trace(trace.data_ptr()) == 0
f(); torch.cuda.synchronize()
lib.gamer_debug_set_trace(None)
t = trace.cpu().numpy().reshape(-1, 4)
t = t[t[:, 0] > 0]
start, end, hw, xcc = t[:, 0], t[:, 1], t[:, 2], t[:, 3] & 0xF
t0 = start.min()
dur_total = (end.max() - t0) / 100.0      # us (100 MHz)
life = (end - start) / 100.0
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
print(f"blocks {len(t)}  kernel span {dur_total:.1f} us  block life mean {life.mean():.1f} us  min {life.min():.1f} max {life.max():.1f}")
print("distinct CUs", len(np.unique(cuid)), "xccs", np.unique(xcc))
# residency: integrate number of live blocks per CU over time
occ = life.sum() / (dur_total * len(np.unique(cuid)))
print(f"mean resident blocks per CU: {occ:.2f}")
# per-XCD finish time
for x in np.unique(xcc):
    m = xcc == x
    print(f"xcc {x}: blocks {m.sum():5d} first start {(start[m].min()-t0)/100:.1f} last end {(end[m].max()-t0)/100:.1f} us  sum life {life[m].sum():.0f}")
# life by q-tile rank
rank = (np.arange(nblk)[: len(trace) // 4][trace.cpu().numpy().reshape(-1, 4)[:, 0] > 0] >> 3) % 8
for rk in range(8):
    m = rank == rk
    print(f"tile rank {rk} (q-tile {7-rk}): mean life {life[m].mean():.1f} us")
# gap analysis on one CU
c = np.unique(cuid)[0]
m = cuid == c
order = np.argsort(start[m])
print("one CU timeline (start, end) us:", [(round((a - t0) / 100, 1), round((b - t0) / 100, 1)) for a, b in zip(start[m][order][:12], end[m][order][:12])])
np.save(os.path.join(ROOT, "gpurun_out", "attn_trace.npy"), t)
per_cu = {}
for c in np.unique(cuid):
    m = cuid == c
    per_cu[c] = (m.sum(), life[m].sum(), (end[m].max() - t0) / 100.0)
vals = np.array(list(per_cu.values()))
print("per-CU blocks: min %d max %d ; busy-sum(us) min %.0f mean %.0f max %.0f ; last end min %.0f max %.0f" % (
    vals[:, 0].min(), vals[:, 0].max(), vals[:, 1].min(), vals[:, 1].mean(), vals[:, 1].max(), vals[:, 2].min(), vals[:, 2].max()))
simd = (hw >> 4) & 3
wave = hw & 0xF
print("wave-0 SIMD distribution", np.bincount(simd.astype(int)), "wave slot ids", np.bincount(wave.astype(int)))
