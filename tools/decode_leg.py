#!/usr/bin/env python3
"""bench.py's decode leg on its own (GAMER_DECODE_GRAPH=0/1): python tools/decode_leg.py [users]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = bench.decode_leg(users=users, cpu_users=0)
print(json.dumps({k: r[k] for k in ("value", "ms_per_batch", "prefill_plus_first_token_ms", "per_token_step_ms")}),
      r["per_token_roofline"]["frac"])
