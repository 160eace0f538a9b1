#!/usr/bin/env python3
"""MFMA-pipe occupancy table from the SQ counter pass of a profile set (tools/prof_r03.sh): for every kernel with MFMA
instructions, busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), vector instructions per MFMA
instruction and the shader clock.  usage: tools/mfma_busy_table.py <tag> <counter dir> [...]  -> markdown on stdout"""
import collections
import csv
import glob
import os
import sys


def table(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dd = dur[k]
                dd[0] += 1
                dd[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = []
    for k, cs in agg.items():
        avg = {c: v / n for c, (n, v) in cs.items()}
        if avg.get("SQ_INSTS_MFMA", 0) <= 0:
            continue
        n, us = dur[k]
        us /= max(n, 1)
        gui = avg["GRBM_GUI_ACTIVE"]
        rows.append((us * n, k.split("(")[0].replace("void gamer::", "").replace("gamer::", "")[:64], n, us,
                     100.0 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui / 8), avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"],
                     gui / 8 / us / 1e3))
    rows.sort(reverse=True)
    out = ["| kernel | launches | avg us | MFMA busy % | VALU instr / MFMA instr | clock GHz |", "|---|---|---|---|---|---|"]
    for _, k, n, us, busy, ratio, ghz in rows:
        out.append(f"| `{k}` | {n} | {us:.0f} | {busy:.0f} | {ratio:.1f} | {ghz:.2f} |")
    return "\n".join(out)


if __name__ == "__main__":
    for d in sys.argv[1:]:
        print(f"## {os.path.basename(d.rstrip('/'))}\n")
        print(table(d) + "\n")
