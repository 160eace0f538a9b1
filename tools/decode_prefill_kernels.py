#!/usr/bin/env python3
"""Where the prompt pass of the cached decode path (DecodeSession.__init__: Engine.forward over users x L0 tokens, evaluation mode, K/V
caches filled, last-row logits) spends its GPU time: HIP events around every C-ABI launch (bench.py's KernelTimer) at the decode leg's
shape: python tools/decode_prefill_kernels.py [users] [beams]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gamer_amd import synthetic  # noqa: E402
from gamer_amd.config import synthetic_config  # noqa: E402
from gamer_amd.decode import DecodeSession  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 256
beams = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = synthetic_config()
eng = Engine(cfg, temperature=0.7)
eng.init_weights(seed=0)
cat = synthetic.make_catalogue(20000, 256, seed=3)
batch = synthetic.make_eval_batch(users, 100, cat, 2, 256, 3, min_his=100, seed=5, behavior_probs=[0.7, 0.25, 0.05])
timer = bench.KernelTimer()
timer.install()
for rep in range(4):
    if rep == 3:
        timer.reset()
        timer.enabled = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = DecodeSession(eng, batch["input_ids"], batch["attention_mask"], batch["actions"], beams, 4)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
timer.enabled = False
rows = timer.summary(1)
print(f"prompt pass with events: wall {wall:.2f} ms, sum of kernel times {sum(r['ms_per_step'] for r in rows):.3f} ms")
for r in rows:
    print(f"  {r['kernel']:24s} {r['ms_per_step'] * 1e3:9.1f} us  {r['launches_per_step']:5.0f} launches  {r['avg_launch_ms'] * 1e3:8.1f} us each")
