#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (any number of passes)."""
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:48]
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
                dd = dur[(k, r["Counter_Name"])]
                dd[0] += 1
                dd[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, cs in agg.items():
    print(f"== {k}")
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:28s} n={n:3d} avg={v / n:.4e}")
    for (kk, c), (n, v) in dur.items():
        if kk == k:
            print(f"   duration_us[{c}] avg={v / n:.1f}")
