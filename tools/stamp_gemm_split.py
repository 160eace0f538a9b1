#!/usr/bin/env python3
"""Where a split-GEMM workgroup (wave 0) spends its time: s_memtime stamps around the phases of the one-stage K loop
(GAMER_GEMM_STAMP=1 selects the stamped instantiation; argv[1] = another build of the library, e.g. one compiled with
-DSP_PINGPONG=1 and run with GAMER_GEMM_PP=1).  Shares only: the counter's rate is not the shader clock's."""
import ctypes, os, sys
os.environ["GAMER_GEMM_STAMP"] = "1"
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from gamer_amd import ops
lib = _lib.load()
lib.gamer_debug_gemm_stamp.argtypes = [ctypes.c_void_p]
T, N, K = 512 * 505, 768, 256
x = torch.randn(T, K, device="cuda"); W = torch.randn(N, K, device="cuda"); y = torch.empty(T, N, device="cuda")
nblk = ((T + 127) // 128) * (N // 128)
buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
assert lib.gamer_debug_gemm_stamp(buf.data_ptr()) == 0
with ops.f32_matmul("split6"):
    for _ in range(3):
        ops.linear_fwd(x, K, W, K, y, N, T, N, K)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.linear_fwd(x, K, W, K, y, N, T, N, K); e.record(); torch.cuda.synchronize()
ti = buf.cpu().numpy().reshape(-1, 8)
idx = np.arange(len(ti))
ok = ti[:, 6] != 0
ti, idx = ti[ok], idx[ok]
t = ti.astype(np.float64)
ms = s.elapsed_time(e)
print(f"kernel {ms:.3f} ms ({2.0*T*N*K/ms/1e9:.1f} TF, stamped build); {len(ti)} workgroups with stamps")
names = ["vmcnt wait + cut + LDS stores", "barrier A", "issue loads + frag reads + 48 MFMA", "barrier B"]
tot = t[:, 4].mean()
nk = K // 32
for i, n in enumerate(names):
    print(f"  {n:36s} {t[:, i].mean() / nk:9.0f} ticks per K-step  ({100 * t[:, i].mean() / tot:5.1f} % of the K loop)")
life = t[:, 7] - t[:, 6]
print(f"  K loop {tot:.0f} ticks, epilogue {t[:, 5].mean():.0f}, workgroup lifetime {life.mean():.0f} ticks (s_memtime counts)")
