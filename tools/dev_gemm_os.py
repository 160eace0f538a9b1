"""Input-gradient GEMMs with 256 input features at T = B x 505 tokens on the output-stationary kernel (csrc/gemm_os.hip, GAMER_GEMM_OS=1)
against the 128 x 128 kernel: launch time and difference.   python tools/dev_gemm_os.py [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, H, dev = B * 505, 256, "cuda"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


torch.manual_seed(0)
for name, N_out, ldy in (("qkv", 768, 768), ("head", 1041, 1056), ("gu-like", 1024, 1024)):
    flat = torch.randn(N_out * H + 8, device=dev) * 0.05
    W = flat[:N_out * H].view(N_out, H)
    dy = torch.randn(T, ldy, device=dev) * 1e-3
    cache = ops.amax_reuse(everything=True)
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=dev)
    res = {}
    with ops.f32_matmul("split3"), cache:
        for form in ("0", "1"):
            os.environ["GAMER_GEMM_OS"] = form
            ops.reload_env()          # (the library caches its switches)
            dx = torch.empty(T, H, device=dev)
            run = lambda: ops.linear_dgrad(dy, ldy, W, H, dx, H, T, N_out, H)
            cache.reset(); run(); cache.reset(); run()
            t = timeit(run)
            res[form] = (dx.clone(), t)
    d = float((res["0"][0] - res["1"][0]).abs().max() / res["0"][0].abs().max())
    print(f"{name:8s} dX [T x 256] = dY [T x {N_out}] W: {res['0'][1]:.3f} -> {res['1'][1]:.3f} ms   max diff / max {d:.2e}", flush=True)
