#!/usr/bin/env python3
"""Per-kernel roofline table of one profile set (profiles/<tag>_kernel_stats.csv + _pmc_{FETCH,WRITE}_SIZE.csv +
_bench_under_rocprof.json): average launch time, HBM bytes per launch from the PMC passes (FETCH doubled per the
gfx950 note of MI355X_MICROARCH.md, WRITE exact), achieved HBM rate against 8 TB/s, and for the MFMA kernels the
algorithmic TFLOP/s the bench computed against the fp32 matrix peak.

usage: tools/roofline_table.py r01j [bench.json]      (writes profiles/r01j_roofline.md; bench.json = the record whose
`kernels` table supplies the algorithmic TFLOP/s, default profiles/<tag>_bench_under_rocprof.json)
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_TBS, FP32_MFMA_PEAK = 8.0, 157.3


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)          # drop the argument list
    return name.replace("gamer::", "")


def main():
    tag = sys.argv[1]
    prof = os.path.join(ROOT, "profiles")
    stats = list(csv.DictReader(open(os.path.join(prof, f"{tag}_kernel_stats.csv"))))
    fetch = {r["kernel"]: float(r["avg_FETCH_SIZE_KB_per_dispatch"])
             for r in csv.DictReader(open(os.path.join(prof, f"{tag}_pmc_FETCH_SIZE.csv")))}
    write = {r["kernel"]: float(r["avg_WRITE_SIZE_KB_per_dispatch"])
             for r in csv.DictReader(open(os.path.join(prof, f"{tag}_pmc_WRITE_SIZE.csv")))}
    def load(path):
        txt = [l for l in open(path) if l.lstrip().startswith("{")]
        return json.loads(txt[-1])
    bench = load(os.path.join(prof, f"{tag}_bench_under_rocprof.json"))
    steps = bench["steps"] + bench["warmup"]
    tfsrc = load(sys.argv[2]) if len(sys.argv) > 2 else bench
    tf = {k["kernel"]: k.get("tflops") for k in (tfsrc.get("kernels") or [])}
    bf16 = tfsrc.get("dtype") == "bf16"
    mfma_peak, mfma_name = (2500.0, "bf16 MFMA") if bf16 else (FP32_MFMA_PEAK, "fp32 MFMA")
    terms = {"split3": 3, "split6": 6, "split9": 9}.get((tfsrc.get("config") or {}).get("matmul"), 0)
    if terms and not bf16:          # fp32 products as `terms` 16-bit piece products: that pipe's peak in fp32-problem FLOPs
        mfma_peak, mfma_name = 2500.0 / terms, f"{'fp16' if terms == 3 else 'bf16'} MFMA / {terms}"
    group = {"gemm_f32_kernel<true, true, 0, false, false, 2, 0": "gemm_fwd",
             "gemm_f32_kernel<true, true, 0, false, false, 2, 1": "gemm_fwd_resid",
             "gemm_f32_kernel<true, false, 0, false, false, 2, 2": "gemm_dgrad_delta",
             "gemm_f32_kernel<true, false, 0": "gemm_dgrad", "gemm_f32_kernel<false, false, 1": "gemm_wgrad",
             "gemm_bf16_wgrad_kernel": "gemm_wgrad", "gemm_bf16_kernel<1": "gemm_fwd_resid", "gemm_bf16_kernel<2": "gemm_dgrad_delta",
             "attn_fwd_kernel<2, true, false": "attn_fwd_self", "attn_fwd_kernel<2, true, true": "attn_fwd_cross",
             "attn_fwd_b_kernel<2, true, false, false": "attn_fwd_self",
             "attn_fwd_s_kernel<2, true, false": "attn_fwd_split_self", "attn_fwd_s_kernel<2, true, true": "attn_fwd_split_cross"}
    total = sum(float(r["total_ms"]) for r in stats)
    lines = [f"# Per-kernel roofline, profile set {tag}", "",
             f"`python3 bench.py --steps {bench['steps']} --warmup {bench['warmup']} --no-cpu-baseline` under rocprofv3 "
             f"(kernel trace; FETCH_SIZE and WRITE_SIZE in separate --pmc passes).  {steps} steps, "
             f"{total / steps:.1f} ms of kernel time per step.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE "
             f"(gfx950: FETCH_SIZE counts 64 B per 128-B request).  Peaks: HBM 8 TB/s (6.3 achievable), {mfma_name} "
             f"{mfma_peak:g} TFLOP/s.", "",
             f"| kernel | launches/step | avg ms | ms/step | HBM GB/launch | HBM TB/s | % of 8 TB/s | alg. TFLOP/s | % of {mfma_name} |",
             "|---|---|---|---|---|---|---|---|---|"]
    for r in stats:
        ms_step = float(r["total_ms"]) / steps
        if ms_step < 0.05:
            continue
        k = r["kernel"]
        avg = float(r["avg_ms"])
        gb = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / 1e9 if k in fetch or k in write else None
        tbs = gb / avg if gb is not None else None             # GB / ms = TB/s
        t = next((tf.get(v) for kk, v in group.items() if kk in k), None)
        lines.append("| `{}` | {:.1f} | {:.3f} | {:.2f} | {} | {} | {} | {} | {} |".format(
            short(k)[:70], int(r["calls"]) / steps, avg, ms_step,
            f"{gb:.2f}" if gb is not None else "-", f"{tbs:.2f}" if tbs is not None else "-",
            f"{100 * tbs / HBM_PEAK_TBS:.0f}" if tbs is not None else "-",
            f"{t:.0f}" if t else "-", f"{100 * t / mfma_peak:.0f}" if t else "-"))
    out = os.path.join(prof, f"{tag}_roofline.md")
    open(out, "w").write("\n".join(lines) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
