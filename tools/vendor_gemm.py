#!/usr/bin/env python3
"""Yardstick: the vendor library (hipBLASLt / rocBLAS through torch.mm) on the bf16 GEMM shapes of the train step, plus a
plain copy kernel of the same bytes (what the memory system gives a kernel that only reads A and writes C)."""
import torch

T, dev, BF = 1024 * 505, "cuda", torch.bfloat16


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


for name, N, K in (("qkv fwd", 768, 256), ("o fwd", 256, 384), ("gate/up", 512, 256), ("head fwd", 1041, 256),
                   ("dqkv dgrad", 256, 768), ("head dgrad", 256, 1088)):
    a = (torch.randn(T, K, device=dev) * 0.5).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.5).to(BF)
    c = torch.empty(T, N, dtype=BF, device=dev)
    ms = timeit(lambda: torch.mm(a, w.t(), out=c))
    src = torch.empty(T, K, dtype=BF, device=dev)
    dst = torch.empty(T, N, dtype=BF, device=dev)
    # the same bytes as a pure stream: read T x K, write T x N
    ms_copy = timeit(lambda: (dst.fill_(1.0), src.sum()))
    by = (T * K + N * K + T * N) * 2
    print(f"{name:12s} N={N:4d} K={K:4d}: torch.mm {ms:.3f} ms {2.0 * T * N * K / ms / 1e9:6.0f} TF/s {by / ms / 1e6:5.0f} GB/s | "
          f"fill C + reduce A {ms_copy:.3f} ms")
