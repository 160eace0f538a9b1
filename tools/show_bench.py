#!/usr/bin/env python3
"""Print the kernel table of bench.py JSON records: python tools/show_bench.py file.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f}: {r['value']:.0f} {r['unit']}  {r['ms_per_step']:.1f} ms/step  dtype {r['dtype']}  loss {r.get('loss')}")
    tot = 0.0
    for k in r["kernels"]:
        tot += k["ms_per_step"]
        bw = f"{k['algorithmic_GBps']:.0f} GB/s" if "algorithmic_GBps" in k else ""
        print(f"   {k['kernel']:<18} {k['ms_per_step']:7.2f} ms  n={k['launches_per_step']:5.1f}  avg {k['avg_launch_ms']:.3f} ms"
              f"  {k.get('tflops', 0):7.1f} TF  {bw}")
    print(f"   sum of listed kernels: {tot:.1f} ms")
