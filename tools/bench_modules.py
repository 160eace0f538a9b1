#!/usr/bin/env python3
"""Encoder of the discriminative baselines (gamer_amd.modules) on one MI355X: forward + backward of a SASRec-shaped
stack, HIP-event timed, next to the oracle (plain PyTorch on the host cores) on a bounded sample.

  python tools/bench_modules.py [--batch 2048] [--seq 50] [--hidden 64] [--heads 2] [--inner 256] [--layers 2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=50)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--inner", type=int, default=256)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from gamer_amd import modules as gm
    from oracle import modules_oracle as mo
    torch.manual_seed(0)
    enc = gm.TransformerEncoder(gm.TransformerEncoderLayer(a.hidden, a.heads, a.inner, dropout=0.2, activation="gelu",
                                                           layer_norm_eps=1e-12), a.layers).cuda().train()
    B, S, D = a.batch, a.seq, a.hidden
    x = torch.randn(B, S, D, device="cuda", requires_grad=True)
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool, device="cuda"))
    mask = torch.where(causal, 0.0, -10000.0)[None, None].expand(B, 1, S, S).contiguous()
    w = torch.randn(B, S, D, device="cuda")

    def step():
        out = enc(x, mask)
        (out * w).sum().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    T = B * S
    # algorithmic FLOP: projections + FFN (2 * MACs, x3 for fwd + bwd) and attention (4 * S^2 * D per sequence, x3.5)
    flop = a.layers * (3 * (2 * T * D * 4 * D + 2 * 2 * T * D * a.inner) + 3.5 * 4 * B * S * S * D)
    res = dict(metric="encoder fwd+bwd sequences/s (SeqRec.modules TransformerEncoder, SASRec shape)", value=B / (ms * 1e-3),
               unit="sequences/s", ms_per_step=ms, config=vars(a), algorithmic_tflops=flop / (ms * 1e-3) / 1e12)
    if not a.no_cpu:
        sd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
        cb = min(B, 256)
        xc = x[:cb].detach().cpu().requires_grad_(True)
        mc, wc = mask[:cb].cpu(), w[:cb].cpu()
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = mo.encoder_forward(leaves, xc, mc, a.layers, a.heads, "gelu", 1e-12)
            (out * wc).sum().backward()
            ts.append(time.perf_counter() - t0)
        res["cpu_baseline"] = dict(value=cb / min(ts[1:]), unit="sequences/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"oracle/modules_oracle.py fwd+bwd, batch {cb}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
