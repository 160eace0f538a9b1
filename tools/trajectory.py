#!/usr/bin/env python3
"""Training trajectories of the matmul / dtype forms side by side: the same weights, the same synthetic stream, the same
dropout seeds, N optimizer steps (fwd + bwd + clip + AdamW, cosine schedule off, lr 5e-4), loss logged every step.
python tools/trajectory.py [--steps 200] [--batch 64] > profiles/<tag>_trajectory.txt
Prints the losses of the fp32-MFMA run and, for split3 / split6 / split9 / bf16, the largest and the final |loss - loss_f32|."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gamer_amd import synthetic  # noqa: E402
from gamer_amd.config import synthetic_config  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402  (weight initialisation only)

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--items", type=int, default=101)
args = ap.parse_args()

cfg = synthetic_config()
sd = orc.init_state_dict(orc.OracleConfig.from_dict(cfg.to_dict()), seed=11)
batches = [synthetic.make_batch(args.batch, args.items, 256, 3, seed=1000 + s, behavior_probs=[0.7, 0.25, 0.05])
           for s in range(16)]                      # a 16-batch epoch, repeated: the loss has to come down
curves = {}
# "f32 again": the same fp32-MFMA configuration a second time - the split-K weight gradients add with fp32 atomics in
# whatever order the workgroups finish, so two identical runs drift apart too; that drift is the yardstick
for name, kw in (("f32", dict(matmul="f32")), ("f32 again", dict(matmul="f32")), ("split3", dict(matmul="split3")),
                 ("split6", dict(matmul="split6")), ("split9", dict(matmul="split9")), ("bf16", dict(dtype="bf16"))):
    eng = Engine(cfg, temperature=0.7, **kw)
    eng.load_state_dict(sd)
    losses = []
    for s in range(args.steps):
        losses.append(float(eng.train_step(batches[s % len(batches)], 5e-4)))
    curves[name] = losses
    del eng
    torch.cuda.empty_cache()
ref = curves["f32"]
print(f"batch {args.batch} x {args.items * 5} tokens, {args.steps} steps, dropout 0.2 (same masks in every run)")
print("step   f32        f32 again  split3     split6     split9     bf16")
for s in list(range(0, args.steps, max(1, args.steps // 20))) + [args.steps - 1]:
    print(f"{s:4d}  " + "  ".join(f"{curves[k][s]:9.6f}" for k in ("f32", "f32 again", "split3", "split6", "split9", "bf16")))
for k in ("f32 again", "split3", "split6", "split9", "bf16"):
    d = [abs(a - b) for a, b in zip(curves[k], ref)]
    print(f"{k:9s} max |loss - loss_f32| over the run {max(d):.3e} (step {d.index(max(d))}), at the last step {d[-1]:.3e}, "
          f"first step {d[0]:.3e}")
