"""The weight-gradient shapes of the step (T = B x 505 tokens) on the 256 x 256-tile kernel (csrc/gemm_wg.hip, GAMER_GEMM_WG=1) against the
128 x 128 kernel: bit equality of dW at the same token chunk, launch time (GEMM + ordered reduce) over a sweep of chunks.
python tools/dev_gemm_wg.py [B] [shape ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
only = sys.argv[2:]
T = B * 505
dev = "cuda"
# name, N_out, K_in, ld of dY, experts
SHAPES = [("qkv", 768, 256, 768, 1), ("o", 256, 384, 256, 1), ("gate", 256, 256, 256, 1), ("head", 1041, 256, 1056, 1),
          ("down", 256, 512, 256, 6), ("gu320", 1024, 320, 1024, 6), ("gu256", 1024, 256, 1024, 6)]


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
for name, N, K, ldy, E in SHAPES:
    if only and name not in only:
        continue
    x = torch.randn(T, K, device=dev) * 2
    dy = torch.randn(T, ldy, device=dev) * 1e-3
    grp = {}
    if E > 1:
        cuts = sorted(torch.randint(0, T, (E - 1,)).tolist())
        offs = torch.tensor([0] + cuts + [T], dtype=torch.int32, device=dev)
        grp = dict(groups=E, group_offsets=offs, strideC=N * K)
    line = []
    with ops.f32_matmul("split3"), ops.amax_reuse(everything=True):
        best = {}
        for kchunk in [int(v) for v in os.environ.get("CHUNKS", "1024,2048,4096,8192").split(",")]:
            res = {}
            for form in ("0", "1"):
                os.environ["GAMER_GEMM_WG"] = form
                ops.reload_env()          # (the library caches its switches)
                dW = torch.zeros(E * N, K, device=dev)
                run = lambda: ops.linear_wgrad(dy, ldy, x, K, dW, K, T, N, K, kchunk=kchunk, **grp)
                run()
                torch.cuda.synchronize()
                res[form] = (dW.clone(), timeit(run))
                best[form] = min(best.get(form, 1e9), res[form][1])
            same = torch.equal(res["0"][0], res["1"][0])
            fin = bool(torch.isfinite(res["1"][0]).all())
            line.append(f"chunk {kchunk}: {res['0'][1]:.3f} -> {res['1'][1]:.3f} ms {'same bits' if same else 'DIFFERENT'}{'' if fin else ' NONFINITE'}")
        if E == 1 and B <= 64:
            ref = dy[:, :N].double().T @ x.double()
            err = float((res["1"][0].double() - ref).abs().max() / ref.abs().max())
            line.append(f"vs fp64 {err:.2e}")
    print(f"{name:6s} dW [{N} x {K}] x {E}: best {best['0']:.3f} -> {best['1']:.3f} ms | " + " | ".join(line), flush=True)
