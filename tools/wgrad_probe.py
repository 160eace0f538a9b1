#!/usr/bin/env python3
"""Token-chunk sweep of the split-K wgrad GEMM: python tools/wgrad_probe.py [--B 128] [--matmul f32|split6|split9].
Prints, per weight-gradient shape of the train step, the time of every candidate chunk and of the rule in
gamer_amd.ops.pick_kchunk (kchunk=None)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gamer_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--matmul", default="f32")
args = ap.parse_args()
ops.set_f32_matmul(args.matmul)


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


T, dev = args.B * 505, "cuda"
offs = torch.tensor([0, 0, T // 5, 2 * (T // 5), 3 * (T // 5), 4 * (T // 5), T], dtype=torch.int32, device=dev)
for name, N, K, grouped in (("qkv", 768, 256, False), ("o", 256, 384, False), ("gate_c", 256, 256, False),
                            ("head", 1041, 256, False), ("gate/up 320", 512, 320, True), ("gate/up 256", 512, 256, True),
                            ("down", 256, 512, True)):
    ldn = (N + 31) // 32 * 32
    y, x = torch.randn(T, ldn, device=dev), torch.randn(T, K, device=dev)
    dW = torch.zeros((6 if grouped else 1) * N, K, device=dev)
    grp = dict(groups=6, group_offsets=offs, strideC=N * K) if grouped else {}
    fl = 2.0 * T * N * K
    res = []
    for kc in (None, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072, 4096):
        t = timeit(lambda: ops.linear_wgrad(y, ldn, x, K, dW, K, T, N, K, kchunk=kc, **grp))
        res.append((kc, t))
    best = min(res[1:], key=lambda r: r[1])
    print(f"T={T} {name:12s} rule {res[0][1]:.3f} ms ({fl / res[0][1] / 1e9:.0f} TF) best kchunk={best[0]} {best[1]:.3f} ms "
          f"({fl / best[1] / 1e9:.0f} TF) | " + " ".join(f"{kc}:{t:.3f}" for kc, t in res[1:]))
