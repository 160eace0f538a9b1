#!/usr/bin/env python3
"""Build gamer_amd/wgrad_chunks.json, the shipped table of split-K token chunks for the weight-gradient GEMMs (gamer_amd/ops.py:
linear_wgrad): for every step form (fp32 split3 / split6 / fp32 MFMA / bf16) and every per-GPU batch of the north star's strong scaling
(1024, 512, 256, 128) one single-process train step is run with GAMER_WGRAD_TUNE_FILE set, which measures each shape's chunk on
first use (ops._tune_kchunk) and records it; the records are merged into the table.  Under data parallelism the ranks never
measure (every rank must use the same chunk), they read this table.   usage (GPU box): python tools/wgrad_table.py [--forms f32:split3 ...] [--merge-only files...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gamer_amd", "wgrad_chunks.json")


def main():
    files = []
    if len(sys.argv) > 1 and sys.argv[1] == "--merge-only":
        files = sys.argv[2:]
    else:
        tmp = os.path.join(ROOT, "gpurun_out", "wgrad_table_tune.json")
        if os.path.exists(tmp):
            os.remove(tmp)
        env = dict(os.environ, GAMER_WGRAD_TUNE="1", GAMER_WGRAD_TUNE_FILE=tmp, GAMER_WGRAD_IGNORE_SHIPPED="1")
        forms = (("f32", "split3"), ("f32", "split6"), ("f32", "f32"), ("bf16", "f32"))
        if len(sys.argv) > 2 and sys.argv[1] == "--forms":          # e.g. --forms f32:split3 (after a change of that form's kernels)
            forms = tuple(tuple(f.split(":")) for f in sys.argv[2:])
        for dtype, matmul in forms:
            for batch in (1024, 512, 256, 128):
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", dtype, "--matmul", matmul, "--batch", str(batch),
                       "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-secondary"]
                r = subprocess.run(cmd, env=env, capture_output=True, text=True)
                print(dtype, matmul, batch, "rc", r.returncode, file=sys.stderr)
        files = [tmp]
    table = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for f in files:
        table.update({k: int(v) for k, v in json.load(open(f)).items()})
    with open(OUT, "w") as fh:
        json.dump(table, fh, indent=0, sort_keys=True)
    print(f"{len(table)} entries -> {OUT}")


if __name__ == "__main__":
    main()
