#!/usr/bin/env python3
"""Micro-benchmark of the bf16 GEMM shapes of the train step (B = 1024 x 505 tokens unless --tokens is given):
python tools/kbench_bf16.py [--tokens T] [--iters N].  Prints ms, TFLOP/s and algorithmic GB/s per shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", type=str, default="", help="time another build of the library (tools/ablate_gemm.sh)")
ap.add_argument("--tokens", type=int, default=1024 * 505)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", type=str, default="")
args = ap.parse_args()
from gamer_amd import _lib  # noqa: E402
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from gamer_amd import ops  # noqa: E402
T, dev, BF = args.tokens, "cuda", torch.bfloat16


def timeit(fn, iters):
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


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(BF)


offs5 = torch.tensor([0] + [T // 5 * (i + 1) for i in range(5)] + [T // 5 * 5], dtype=torch.int32, device=dev)   # 6 experts (expert 0 empty)
offs5 = torch.tensor([0, 0] + [T // 5 * (i + 1) for i in range(4)] + [T], dtype=torch.int32, device=dev)
shapes = [("qkv fwd", T, 768, 256), ("o fwd(plain)", T, 256, 384), ("gate/up fwd 256 (grouped)", T, 512, 256),
          ("gate/up fwd 320 (grouped)", T, 512, 320), ("head fwd", T, 1041, 256), ("dqkv dgrad", T, 256, 768),
          ("dhm dgrad (grouped)", T, 512, 256), ("dhin dgrad (grouped)", T, 320, 512), ("head dgrad", T, 256, 1088)]
print(f"T = {T}")
for name, M, N, K in shapes:
    if args.only and args.only not in name:
        continue
    grouped = "grouped" in name
    a = rnd(M, K)
    w = rnd((6 if grouped else 1) * N, K)
    ldc = 1088 if N == 1041 else N
    c = torch.empty(M, ldc, dtype=BF, device=dev)
    kw = dict(groups=6, group_offsets=offs5, strideB=N * K) if grouped else {}
    ms = timeit(lambda: ops.gemm(a, K, 1, w, K, 1, c, ldc, M, N, K, **kw), args.iters)
    fl = 2.0 * M * N * K
    by = (M * K + N * K + M * N) * 2
    print(f"  {name:<28} M={M} N={N:4d} K={K:4d}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {by / ms / 1e6:7.0f} GB/s")
# residual epilogue (o_proj, down)
for name, N, K, grouped in (("o_proj fwd + resid", 256, 384, False), ("down fwd + resid (grouped, row_map)", 256, 512, True)):
    if args.only and args.only not in name:
        continue
    a, w = rnd(T, K), rnd((6 if grouped else 1) * N, K)
    res, out = torch.randn(T, N, device=dev), torch.empty(T, N, device=dev)
    perm = torch.randperm(T, device=dev).int() if grouped else None
    kw = dict(groups=6, group_offsets=offs5, strideB=N * K, row_map=perm) if grouped else {}
    ms = timeit(lambda: ops.gemm(a, K, 1, w, K, 1, out, N, T, N, K, resid=res, p_drop=0.2, seed=1, **kw), args.iters)
    fl, by = 2.0 * T * N * K, (T * K + N * K) * 2 + 2 * T * N * 4
    print(f"  {name:<28} M={T} N={N:4d} K={K:4d}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {by / ms / 1e6:7.0f} GB/s")
# wgrad
for name, M, N in (("qkv wgrad", 768, 256), ("o wgrad", 256, 384), ("gate wgrad (grouped)", 512, 320), ("down wgrad (grouped)", 256, 512),
                   ("head wgrad", 1041, 256)):
    if args.only and args.only not in name:
        continue
    grouped = "grouped" in name
    lda = 1088 if M == 1041 else M
    dy, x = rnd(T, lda), rnd(T, N)
    dw = torch.zeros((6 if grouped else 1) * M, N, device=dev)
    kw = dict(groups=6, group_offsets=offs5, strideC=M * N) if grouped else {}
    ms = timeit(lambda: ops.linear_wgrad(dy, lda, x, N, dw, N, T, M, N, **kw), args.iters)
    fl, by = 2.0 * T * M * N, (T * M + T * N) * 2
    print(f"  {name:<28} M={M:4d} N={N:4d} K={T}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {by / ms / 1e6:7.0f} GB/s")
