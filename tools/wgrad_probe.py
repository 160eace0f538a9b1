import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/gamer_amd") else os.environ.get("GRAFT_REPO_ROOT","."))
from gamer_amd import ops
sys.path.insert(0, "tools")
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
T = 1024 * 505
dev = "cuda"
for N, K in ((512, 320), (512, 256), (256, 512), (512, 384)):
    y = torch.randn(T, N, device=dev); x = torch.randn(T, K, device=dev)
    dW = torch.zeros(6 * N, K, device=dev)
    offs = torch.tensor([0, 0, T // 5, 2 * (T // 5), 3 * (T // 5), 4 * (T // 5), T], dtype=torch.int32, device=dev)
    fl = 2.0 * T * N * K
    for kc in (None, 1024, 2048, 8192):
        t_plain = timeit(lambda: ops.linear_wgrad(y, N, x, K, dW, K, T, N, K, kchunk=kc))
        t_grp = timeit(lambda: ops.linear_wgrad(y, N, x, K, dW, K, T, N, K, groups=6, group_offsets=offs, strideC=N * K, kchunk=kc))
        print(f"N={N} K={K} kchunk={kc}: plain {t_plain:.3f} ms {fl/t_plain/1e9:.1f} TF | grouped {t_grp:.3f} ms {fl/t_grp/1e9:.1f} TF")
