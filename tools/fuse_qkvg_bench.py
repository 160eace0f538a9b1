"""Same-box micro-benchmark: the cross block's q|k|v (N = 768) and output-gate (N = 256) projections as two GEMMs against one fused
GEMM with N = 1024 (forward, input gradient with / without the accumulate pass, weight gradient).  python tools/fuse_qkvg_bench.py [batch]"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from gamer_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, H, N1, N2 = B * 505, 256, 768, 256
dev = "cuda"
ops.set_f32_matmul("split3")
x = torch.randn(T, H, device=dev)
W1, W2 = torch.randn(N1, H, device=dev) * 0.05, torch.randn(N2, H, device=dev) * 0.05
W12 = torch.cat([W1, W2]).contiguous()
y1, y2 = torch.empty(T, N1, device=dev), torch.empty(T, N2, device=dev)
y12 = torch.empty(T, N1 + N2, device=dev)
dx = torch.empty(T, H, device=dev)
dW1, dW2, dW12 = torch.zeros(N1, H, device=dev), torch.zeros(N2, H, device=dev), torch.zeros(N1 + N2, H, device=dev)
def timeit(fn, iters=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
with ops.amax_reuse(everything=True):
    def f2():
        ops.linear_fwd(x, H, W1, H, y1, N1, T, N1, H); ops.linear_fwd(x, H, W2, H, y2, N2, T, N2, H)
    def f1():
        ops.linear_fwd(x, H, W12, H, y12, N1 + N2, T, N1 + N2, H)
    print(f"fwd   768 + 256: {timeit(f2):.3f} ms   1024: {timeit(f1):.3f} ms")
    def d2():
        ops.linear_dgrad(y1, N1, W1, H, dx, H, T, N1, H); ops.linear_dgrad(y2, N2, W2, H, dx, H, T, N2, H, accumulate=True)
    def d1():
        ops.linear_dgrad(y12, N1 + N2, W12, H, dx, H, T, N1 + N2, H)
    print(f"dgrad 768 + 256: {timeit(d2):.3f} ms   1024: {timeit(d1):.3f} ms")
    def w2():
        ops.linear_wgrad(y1, N1, x, H, dW1, H, T, N1, H); ops.linear_wgrad(y2, N2, x, H, dW2, H, T, N2, H)
    def w1():
        ops.linear_wgrad(y12, N1 + N2, x, H, dW12, H, T, N1 + N2, H)
    print(f"wgrad 768 + 256: {timeit(w2):.3f} ms   1024: {timeit(w1):.3f} ms")
