#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small tracked files under profiles/.

usage: tools/summarize_prof.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>]
"""
import collections
import csv
import glob
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    fetch_dir = sys.argv[3] if len(sys.argv) > 3 else None
    write_dir = sys.argv[4] if len(sys.argv) > 4 else None
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    ks = find(stats_dir, "kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "percent"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}", f"{float(r['AverageNs']) / 1e6:.4f}",
                        f"{float(r['MinNs']) / 1e6:.4f}", f"{float(r['MaxNs']) / 1e6:.4f}", r["Percentage"]])
    for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        if not d:
            continue
        cc = find(d, "counter_collection.csv")
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != name:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        with open(os.path.join(out_dir, f"{tag}_pmc_{name}.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "dispatches", f"sum_{name}_KB", f"avg_{name}_KB_per_dispatch"])
            for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, n, f"{v:.1f}", f"{v / n:.1f}"])
    print("wrote", sorted(os.listdir(out_dir)))


if __name__ == "__main__":
    main()
