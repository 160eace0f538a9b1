#!/usr/bin/env python3
"""Evaluation-path throughput: users/s of trie-constrained beam search (his_len 100, 20 beams, shipped model).

  python tools/bench_decode.py --B 64 --beams 20 [--cpu-users 2]

Prints one JSON line; ``cpu_baseline`` times oracle/decode_oracle.py (the CPU restatement) on a few users.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import synthetic  # noqa: E402
from gamer_amd.config import synthetic_config  # noqa: E402
from gamer_amd.decode import ItemTrie, beam_search  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--beams", type=int, default=20)
    ap.add_argument("--his", type=int, default=100)
    ap.add_argument("--items", type=int, default=20000, help="catalogue size")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--cpu-users", type=int, default=1)
    ap.add_argument("--no-cache", dest="no_cache", action="store_true", help="re-run the whole sequence every step")
    args = ap.parse_args()
    cfg = synthetic_config()
    eng = Engine(cfg, temperature=0.7)
    eng.init_weights(seed=0)
    cat = synthetic.make_catalogue(args.items, 256, seed=3)
    tb = 2
    items = synthetic.item_tokens(cat, tb, 256).tolist()
    trie = ItemTrie(items)
    batch = synthetic.make_eval_batch(args.B, args.his, cat, tb, 256, 3, min_his=args.his, seed=5,
                                      behavior_probs=[0.7, 0.25, 0.05])
    run = lambda: beam_search(eng, batch["input_ids"], batch["attention_mask"], batch["actions"], trie, args.beams, 4,
                              use_cache=not args.no_cache)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        seqs, scores = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    out = dict(metric="evaluation users/s, Qwen3Multi SMB decoder, trie-constrained beam search", value=args.B / dt,
               unit="users/s", ms_per_batch=dt * 1e3,
               config=dict(workload=f"{args.B} users x {args.beams} beams, history {args.his} items, 4 new tokens, "
                                    f"catalogue {args.items} items, fp32, " +
                                    ("whole sequence re-run per step" if args.no_cache else
                                     "K/V cache: prompt once per sample, generated positions per beam")))
    if args.cpu_users > 0:
        from oracle import decode_oracle as dec, qwen3multi_oracle as orc
        ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
        sd = {k: v.detach().cpu().clone() for k, v in eng.params.items()}
        n = args.cpu_users
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        otrie = dec.ItemTrie(items)
        t0 = time.perf_counter()
        oseq, osc = dec.beam_search(sd, ocfg, batch["input_ids"][:n], batch["attention_mask"][:n], batch["actions"][:n],
                                    otrie, args.beams, 4)
        cdt = time.perf_counter() - t0
        same = bool(torch.equal(oseq, seqs[: n * args.beams].cpu()))
        out["cpu_baseline"] = dict(value=n / cdt, unit="users/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"oracle/decode_oracle.py beam search on {n} user(s)",
                                   sequences_equal_gpu=same,
                                   max_score_diff=float((osc - scores[: n * args.beams].cpu()).abs().max()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
