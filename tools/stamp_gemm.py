#!/usr/bin/env python3
"""Where a GEMM workgroup spends its cycles (diagnostic build with s_memtime stamps; shares only)."""
import ctypes, os, sys
os.environ["GAMER_GEMM_STAMP"] = "1"
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops, _lib
lib = _lib.load()
lib.gamer_debug_gemm_stamp.argtypes = [ctypes.c_void_p]
T, N, K = 512 * 505, 768, 256
x = torch.randn(T, K, device="cuda"); W = torch.randn(N, K, device="cuda"); y = torch.empty(T, N, device="cuda")
nblk = ((T + 127) // 128) * (N // 128)
buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
assert lib.gamer_debug_gemm_stamp(buf.data_ptr()) == 0
for _ in range(3):
    ops.linear_fwd(x, K, W, K, y, N, T, N, K)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); ops.linear_fwd(x, K, W, K, y, N, T, N, K); e.record(); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
print(f"kernel {s.elapsed_time(e):.3f} ms ({2.0*T*N*K/s.elapsed_time(e)/1e9:.1f} TF, stamped build)")
names = ["issue global loads", "LDS frag reads + 64 MFMA", "vmcnt wait + LDS stores", "barrier"]
tot = t[:, 4].mean()
for i, n in enumerate(names):
    print(f"  {n:28s} {t[:, i].mean() / 8:9.0f} cycles per K-step  ({100 * t[:, i].mean() / tot:5.1f} % of the K loop)")
print(f"  K loop {tot:.0f} cycles, epilogue {t[:, 5].mean():.0f} cycles; MFMA-only time would be {8 * 64 * 64} cycles")
life = (t[:, 7] - t[:, 6])
print(f"  workgroup lifetime mean {life.mean():.0f} cycles; kernel span {(t[:, 7].max() - t[:, 6].min()):.0f} cycles; blocks {len(t)}")
print(f"  resident workgroups per CU (sum life / span / 256): {life.sum() / (t[:, 7].max() - t[:, 6].min()) / 256:.2f}")
