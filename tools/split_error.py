#!/usr/bin/env python3
"""Error of the fp32 GEMM forms against an fp64 product of the same fp32 inputs, on the three layouts of the train step
(forward, dgrad, split-K wgrad): python tools/split_error.py [--tokens T].  Prints max and rms error relative to
sum_k |a_k b_k| (the scale fp32 rounding errors are proportional to) for f32 (v_mfma_f32_32x32x2_f32), split3 (the default: three fp16
piece products; the weight gradient in the order a1.b0, a0.b0, a0.b1 that gemm_wg's single fragment set needs), split6, split9."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gamer_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=8192)
args = ap.parse_args()
T, dev = args.tokens, "cuda"
torch.manual_seed(0)


def report(name, got, ref, scale):
    e = (got.double() - ref).abs() / scale
    print(f"  {name:8s} max {e.max().item():.3e}  rms {e.pow(2).mean().sqrt().item():.3e}")


for N, K in ((768, 256), (256, 512), (1041, 256)):
    x = torch.randn(T, K, device=dev) * torch.exp(torch.randn(T, K, device=dev))      # wide dynamic range
    W = torch.randn(N, K, device=dev) * 0.05
    ldn = (N + 3) // 4 * 4                           # leading dimensions are multiples of 4 (the tied head: 1041 -> 1044)
    dy_buf = torch.zeros(T, ldn, device=dev)
    dy_buf[:, :N] = torch.randn(T, N, device=dev)
    dy = dy_buf[:, :N]
    xd, Wd, dyd = x.double(), W.double(), dy.double()
    ref_f, sc_f = xd @ Wd.T, xd.abs() @ Wd.abs().T
    ref_d, sc_d = dyd @ Wd, dyd.abs() @ Wd.abs()
    ref_w, sc_w = dyd.T @ xd, dyd.abs().T @ xd.abs()
    print(f"T={T} N={N} K={K}")
    for mode in ("f32", "split3", "split6", "split9"):
        ops.set_f32_matmul(mode)
        y = torch.empty(T, ldn, device=dev)[:, :N]
        ops.linear_fwd(x, K, W, K, y, ldn, T, N, K)
        dx = torch.empty(T, K, device=dev)
        ops.linear_dgrad(dy, ldn, W, K, dx, K, T, N, K)
        dW = torch.zeros(N, K, device=dev)
        ops.linear_wgrad(dy, ldn, x, K, dW, K, T, N, K)
        torch.cuda.synchronize()
        print(f" {mode}")
        report("fwd", y, ref_f, sc_f)
        report("dgrad", dx, ref_d, sc_d)
        report("wgrad", dW, ref_w, sc_w)
