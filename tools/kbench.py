#!/usr/bin/env python3
"""Micro-benchmarks of single kernels at the bench shapes (for A/B work and rocprofv3 --pmc runs).

  python tools/kbench.py attn --B 256          attention fwd + bwd, self and cross, dropout 0.2
  python tools/kbench.py gemm --B 256          the GEMM shapes of one decoder layer (fwd/dgrad/wgrad)
  python tools/kbench.py elem --B 256          the HBM-bound kernels
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import _lib  # noqa: E402
if "--lib" in sys.argv:                                  # another build of the library (tools/ablate_gemm.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from gamer_amd import ops, synthetic  # noqa: E402
from gamer_amd.config import synthetic_config  # noqa: E402


def timeit(fn, iters=5, warm=2):
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


def pairs(batch):
    keep, a = batch["attention_mask"].bool(), batch["actions"]
    p_self = int(keep.long().cumsum(1).sum())
    p_cross = 0
    for lv in torch.unique(a).tolist():
        p_cross += int(((a < lv) & keep).long().cumsum(1)[a == lv].sum())
    return p_self, p_cross


def bench_attn(args):
    cfg = synthetic_config()
    B, S, nq, nkv = args.B, args.items * 5, 6, 3
    T = B * S
    batch = synthetic.make_batch(B, args.items, 256, 3, ragged=args.ragged, seed=1, behavior_probs=[0.7, 0.25, 0.05])
    p_self, p_cross = pairs(batch)
    dev = "cuda"
    r = ops.alloc_router_outputs(B, S, dev)
    ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev),
                   cfg.behavior_lut().to(dev), 5, 4, 8, r)
    q = torch.randn(T, nq * 64, device=dev)
    k = torch.randn(T, nkv * 64, device=dev)
    qkv = torch.randn(T, (nq + 2 * nkv) * 64, device=dev)
    v = qkv[:, (nq + nkv) * 64:]
    o = torch.empty(T, nq * 64, device=dev)
    do = torch.randn(T, nq * 64, device=dev)
    lse = torch.empty(B, nq, S, device=dev)
    delta = torch.empty(B, nq, S, device=dev)
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    dqkv = torch.empty_like(qkv)
    dv = dqkv[:, (nq + nkv) * 64:]
    p = args.p
    n_t = (S + 31) // 32
    order = (torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, n_t, dtype=torch.int32, device=dev),
             torch.empty(B, n_t, dtype=torch.int32, device=dev))
    ops.attn_row_order(r["empty_cross"], *order)
    for name, kl, ql, re_, te, npairs in (("self", r["kl_self"], None, r["empty_self"], r["tile_empty_self"], p_self),
                                          ("cross", r["kl_cross"], r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], p_cross)):
        if args.only and args.only != name:
            continue
        od = order if (name == "cross" and not args.no_order) else None
        f = lambda: ops.attn_fwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, re_, te, B, S, nq, nkv, 0.125, p, 7, o, lse,
                                 order=od)
        b = lambda: ops.attn_bwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv,
                                 0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od)
        dsw = torch.empty(ops.attn_ds_work_numel(B, S, nq), device=dev)
        b2 = lambda: ops.attn_bwd(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, re_, te, B, S, nq, nkv,
                                  0.125, p, 7, delta, dq, nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od, ds_work=dsw)
        tf, tb = timeit(f, args.iters), timeit(b, args.iters)
        tb2 = timeit(b2, args.iters)
        print(f"attn_{name}: bwd with dS spill {tb2:.3f} ms (recompute {tb:.3f} ms)")
        if args.bf16:
            BF = torch.bfloat16
            q16, k16, qkv16, do16 = q.to(BF), k.to(BF), qkv.to(BF), do.to(BF)
            v16 = qkv16[:, (nq + nkv) * 64:]
            o16, dq16, dk16, dqkv16 = torch.empty_like(q16), torch.empty_like(q16), torch.empty_like(k16), torch.empty_like(qkv16)
            ord16 = (od[0], od[2], re_) if od is not None else None
            f16 = lambda: ops.attn_fwd_bf16(q16, nq * 64, k16, nkv * 64, v16, qkv16.shape[1], kl, ql, B, S, nq, nkv, 0.125, p, 7,
                                            o16, lse, order=ord16)
            b16 = lambda: ops.attn_bwd_bf16(q16, nq * 64, k16, nkv * 64, v16, qkv16.shape[1], o16, do16, lse, kl, ql, B, S, nq, nkv,
                                            0.125, p, 7, delta, dq16, nq * 64, dk16, nkv * 64, dqkv16[:, (nq + nkv) * 64:],
                                            qkv16.shape[1], order=ord16)
            tf16, tb16 = timeit(f16, args.iters), timeit(b16, args.iters)
            print(f"attn_{name} bf16: fwd {tf16:.3f} ms  bwd {tb16:.3f} ms")
            f()                                    # restore the fp32 lse for the lines below
        causal_pairs = B * S * (S + 1) // 2
        print(f"attn_{name}: fwd {tf:.3f} ms ({4 * 64 * nq * npairs / tf / 1e9:.1f} TF alg, "
              f"{4 * 64 * nq * causal_pairs / tf / 1e9:.1f} TF causal-dense)  bwd {tb:.3f} ms "
              f"({10 * 64 * nq * npairs / tb / 1e9:.1f} TF alg, {10 * 64 * nq * causal_pairs / tb / 1e9:.1f} TF causal-dense)")


def bench_gemm(args):
    B, S = args.B, args.items * 5
    T = B * S
    dev = "cuda"
    shapes = [("qkv", 768, 256), ("o", 256, 384), ("gate/up", 512, 320), ("down", 256, 512), ("head", 1041, 256)]
    offs = torch.tensor([0] + [T // 5 * i for i in range(1, 6)] + [T], dtype=torch.int32)[[0, 0, 1, 2, 3, 4, 6]].contiguous()
    offs = torch.tensor([0, 0, T // 5, 2 * (T // 5), 3 * (T // 5), 4 * (T // 5), T], dtype=torch.int32, device=dev)
    for name, N, K in shapes:
        x = torch.randn(T, K, device=dev)
        ldn = (N + 31) // 32 * 32
        y = torch.empty(T, ldn, device=dev)
        grouped = name in ("gate/up", "down")
        E = 6 if grouped else 1
        W = torch.randn(E * N, K, device=dev)
        dW = torch.zeros(E * N, K, device=dev)
        dx = torch.empty(T, K, device=dev)
        grp = dict(groups=6, group_offsets=offs) if grouped else {}
        f = lambda: ops.linear_fwd(x, K, W, K, y, ldn, T, N, K, strideB=N * K if grouped else 0, **grp)
        d = lambda: ops.linear_dgrad(y, ldn, W, K, dx, K, T, N, K, strideB=N * K if grouped else 0, **grp)
        w = lambda: ops.linear_wgrad(y, ldn, x, K, dW, K, T, N, K, strideC=N * K if grouped else 0, **grp)
        fl = 2.0 * T * N * K
        if args.matmul == "split3":          # the GEMM kernels alone (operand maxima measured once), then the maxima
            with ops.amax_reuse(everything=True):
                tf, td, tw = timeit(f, args.iters), timeit(d, args.iters), timeit(w, args.iters)
            ta = timeit(lambda: (ops.absmax_slot(x, 1, 0, T, K, K), ops.absmax_slot(y, 1, 0, T, N, ldn)), args.iters)
            print(f"     absmax of x [T,{K}] + y [T,{N}] (ld {ldn}): {ta:.3f} ms  {T * (K + N) * 4 / ta / 1e6:.0f} GB/s")
        else:
            tf, td, tw = timeit(f, args.iters), timeit(d, args.iters), timeit(w, args.iters)
        print(f"gemm {name:8s} N={N:4d} K={K:3d}: fwd {tf:.3f} ms {fl / tf / 1e9:6.1f} TF | dgrad {td:.3f} ms "
              f"{fl / td / 1e9:6.1f} TF | wgrad {tw:.3f} ms {fl / tw / 1e9:6.1f} TF")


def bench_gemm_k(args):
    """Main-loop efficiency vs prologue/epilogue cost: the forward GEMM form at N = 768 with K from 128 to 4096
    (same number of output tiles, more K-steps per tile)."""
    T = args.B * args.items * 5
    T = min(T, 131072)
    dev = "cuda"
    N = 768
    for K in (128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(T, K, device=dev)
        W = torch.randn(N, K, device=dev)
        y = torch.empty(T, N, device=dev)
        t = timeit(lambda: ops.linear_fwd(x, K, W, K, y, N, T, N, K), args.iters)
        print(f"gemm fwd T={T} N={N} K={K:5d}: {t:.3f} ms {2.0 * T * N * K / t / 1e9:6.1f} TF  ({K // 32} K-steps per tile)")


def bench_elem(args):
    B, S, H, I = args.B, args.items * 5, 256, 512
    T = B * S
    dev = "cuda"
    x, y = torch.randn(T, H, device=dev), torch.empty(T, H, device=dev)
    w = torch.ones(H, device=dev)
    t = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, y), args.iters)
    print(f"rmsnorm_fwd {t:.3f} ms  {2 * T * H * 4 / t / 1e6:.0f} GB/s")
    part = torch.empty(2048, H, device=dev)          # (the engine's grid: ws.norm_partial)
    dx = torch.zeros(T, H, device=dev)
    t = timeit(lambda: ops.rmsnorm_bwd(x, w, y, H, 1e-6, dx, part, True), args.iters)
    print(f"rmsnorm_bwd {t:.3f} ms  {4 * T * H * 4 / t / 1e6:.0f} GB/s")
    t = timeit(lambda: ops.residual_dropout_fwd(x, y, 0.2, 3, None, dx), args.iters)
    print(f"residual_dropout_fwd {t:.3f} ms  {3 * T * H * 4 / t / 1e6:.0f} GB/s")
    g, u, hm = torch.randn(T, I, device=dev), torch.randn(T, I, device=dev), torch.empty(T, I, device=dev)
    t = timeit(lambda: ops.swiglu_fwd(g, u, T * I, 0.2, 3, hm), args.iters)
    print(f"swiglu_fwd {t:.3f} ms  {3 * T * I * 4 / t / 1e6:.0f} GB/s")
    t = timeit(lambda: ops.swiglu_bwd(g, u, hm, T * I, 0.2, 3), args.iters)
    print(f"swiglu_bwd {t:.3f} ms  {5 * T * I * 4 / t / 1e6:.0f} GB/s")
    nq, nkv = 6, 3
    qkv = torch.randn(T, 768, device=dev)
    q_rot, k_rot = torch.empty(T, 384, device=dev), torch.empty(T, 192, device=dev)
    cos, sin = torch.randn(S, 64, device=dev), torch.randn(S, 64, device=dev)
    w64 = torch.ones(64, device=dev)
    t = timeit(lambda: ops.qknorm_rope_fwd(qkv, S, nq, nkv, w64, w64, 1e-6, cos, sin, q_rot, k_rot), args.iters)
    print(f"qknorm_rope_fwd {t:.3f} ms  {2 * T * 576 * 4 / t / 1e6:.0f} GB/s")
    dqkv = torch.empty_like(qkv)
    dw1, dw2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    t = timeit(lambda: ops.qknorm_rope_bwd(qkv, q_rot, k_rot, S, nq, nkv, w64, w64, 1e-6, cos, sin, dqkv, dw1, dw2), args.iters)
    print(f"qknorm_rope_bwd {t:.3f} ms  {3 * T * 576 * 4 / t / 1e6:.0f} GB/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "gemm", "gemmk", "elem"])
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--items", type=int, default=101)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--p", type=float, default=0.2)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--bf16", action="store_true", help="attn: also time the bf16 kernels on the same inputs")
    ap.add_argument("--no-order", dest="no_order", action="store_true", help="cross attention without the row order")
    ap.add_argument("--lib", default="")
    ap.add_argument("--matmul", default="f32", choices=sorted(ops.MATMUL_MODES), help="fp32 GEMM form (gamer_gemm_f32_split)")
    args = ap.parse_args()
    ops.set_f32_matmul(args.matmul)
    {"attn": bench_attn, "gemm": bench_gemm, "gemmk": bench_gemm_k, "elem": bench_elem}[args.what](args)
