"""Same-box micro-benchmark: the experts' gate and up projections as two GEMMs against one fused GEMM (forward, input gradient, weight
gradient; non-grouped approximation of the engine's shapes).  python tools/fuse_gu_bench.py [per-GPU batch]"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from gamer_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, H, I = B * 505, 256, 512
dev = "cuda"
ops.set_f32_matmul("split3")
x = torch.randn(T, H, device=dev)
Wg, Wu = torch.randn(I, H, device=dev) * 0.05, torch.randn(I, H, device=dev) * 0.05
Wgu = torch.cat([Wg, Wu]).contiguous()
g, u = torch.empty(T, I, device=dev), torch.empty(T, I, device=dev)
gu = torch.empty(T, 2 * I, device=dev)
dx = torch.empty(T, H, device=dev)
dWg, dWu, dWgu = torch.zeros(I, H, device=dev), torch.zeros(I, H, device=dev), torch.zeros(2 * I, H, device=dev)
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
        ops.linear_fwd(x, H, Wg, H, g, I, T, I, H); ops.linear_fwd(x, H, Wu, H, u, I, T, I, H)
    def f1():
        ops.linear_fwd(x, H, Wgu, H, gu, 2 * I, T, 2 * I, H)
    print(f"fwd   2 x N=512: {timeit(f2):.3f} ms   1 x N=1024: {timeit(f1):.3f} ms")
    def d2():
        ops.linear_dgrad(g, I, Wg, H, dx, H, T, I, H); ops.linear_dgrad(u, I, Wu, H, dx, H, T, I, H, accumulate=True)
    def d1():
        ops.linear_dgrad(gu, 2 * I, Wgu, H, dx, H, T, 2 * I, H)
    print(f"dgrad 2 x K=512: {timeit(d2):.3f} ms   1 x K=1024: {timeit(d1):.3f} ms")
    def w2():
        ops.linear_wgrad(g, I, x, H, dWg, H, T, I, H); ops.linear_wgrad(u, I, x, H, dWu, H, T, I, H)
    def w1():
        ops.linear_wgrad(gu, 2 * I, x, H, dWgu, H, T, 2 * I, H)
    print(f"wgrad 2 x N=512: {timeit(w2):.3f} ms   1 x N=1024: {timeit(w1):.3f} ms")
