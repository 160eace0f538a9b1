#!/usr/bin/env python3
"""Does the row stride of the operands / the result matter (memory-channel collisions of power-of-two strides)?
python tools/ld_probe.py [--dtype bf16|f32]: the q|k|v forward GEMM (N=768, K=256) and the o_proj shape with padded strides."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gamer_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--matmul", default="f32")
args = ap.parse_args()
ops.set_f32_matmul(args.matmul)
DT = torch.bfloat16 if args.dtype == "bf16" else torch.float32
T, dev = 1024 * 505, "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for N, K in ((768, 256), (256, 384), (512, 256)):
    for pa in (0, 16, 32, 64):
        for pc in (0, 16, 32, 64):
            lda, ldc = K + pa, N + pc
            a = (torch.randn(T, lda, device=dev) * 0.5).to(DT)
            w = (torch.randn(N, K, device=dev) * 0.5).to(DT)
            c = torch.empty(T, ldc, dtype=DT, device=dev)
            ms = timeit(lambda: ops.gemm(a, lda, 1, w, K, 1, c, ldc, T, N, K))
            print(f"N={N} K={K} lda={lda} ldc={ldc}: {ms:.3f} ms  {2.0 * T * N * K / ms / 1e9:.0f} TF/s")
            del a, c
