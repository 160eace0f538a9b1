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
    # Algorithmic FLOPs per LAUNCH of every kernel symbol of the fp32 forms at the shipped architecture (H 256, 6 / 3 heads of 64, I 512,
    # E 6, V 1041, 8 layers of which 4 with the cross block): the Linear launches of a step by the kernel that runs them (2 M N K each,
    # T = the sidecar's batch x items x 5 tokens), attention from the pair counts behind the bench record's family rows (SURVEY 8(d):
    # 1536 FLOP per allowed pair forward; the backward's dQ + dK/dV kernels together = 2 x forward, shown on the combined line).
    sym_flops = {}
    meta_path = os.path.join(prof, f"{tag}_pmc_meta.json")
    if not bf16 and os.path.exists(meta_path):
        meta = json.load(open(meta_path))
        T = float(meta["batch"]) * meta["items"] * 5
        mm = lambda n, k: 2.0 * T * n * k

        def avg(*terms):                              # (count, flops) pairs -> mean FLOPs per launch of the symbol
            n = sum(c for c, _ in terms)
            return sum(c * f for c, f in terms) / n
        sfx = f", {terms}," if terms else ", 0,"
        sym_flops = {
            "gemm_as_kernel<4, false, 0, 8>": avg((12, mm(768, 256)), (1, mm(1041, 256))),            # q|k|v, head
            "gemm_as_kernel<4, false, 0, 4>": mm(256, 256),                                          # cross gate (r06b and earlier: also the experts' gate|up)
            "gemm_as_kernel<4, false, 5, 4>": mm(1024, 256),                                         # experts' gate|up with the SwiGLU forward in its epilogue
            "gemm_as_kernel<4, true, 0, 4>": mm(256, 256),                                           # cross gate input gradient
            "gemm_os_kernel": avg((12, mm(768, 256)), (1, mm(1041, 256)), (8, mm(1024, 256))),       # input gradients with 256 input features
            "gemm_wg_kernel<true>": avg((12, mm(768, 256)), (4, mm(256, 256)), (8, mm(256, 512)), (8, mm(1024, 256))),
            "gemm_f32_kernel<true, true, 0, false, false, 2, 0" + sfx: mm(256, 384),                # cross o_proj
            "gemm_f32_kernel<true, true, 0, false, false, 2, 1" + sfx: avg((8, mm(256, 384)), (8, mm(256, 512))),   # o_proj, down + residual
            "gemm_f32_kernel<true, false, 0, false, false, 2, 2" + sfx: mm(384, 256),               # o_proj input gradient + delta
            "gemm_f32_kernel<true, false, 0, false, false, 2, 4" + sfx: mm(512, 256),               # down input gradient + SwiGLU backward
            "gemm_f32_kernel<false, false, 1, false, false, 2, 0" + sfx: avg((12, mm(256, 384)), (1, mm(1041, 256))),   # o_proj, head weight gradients
        }
        fam = {k["kernel"]: k for k in (tfsrc.get("kernels") or [])}
        per_launch = lambda f: fam[f]["tflops"] * 1e12 * fam[f]["ms_per_step"] * 1e-3 / fam[f]["launches_per_step"] if f in fam and "tflops" in fam[f] else None
        for stem, f in (("attn_fwd_r_kernel<2, true, false, false>", "attn_fwd_self"), ("attn_fwd_r_kernel<2, true, false, true>", "attn_fwd_cross"),
                        ("attn_fwd_kernel<2, true, false", "attn_fwd_self"), ("attn_fwd_kernel<2, true, true", "attn_fwd_cross")):
            if per_launch(f):
                sym_flops[stem] = per_launch(f)
    lines = [f"# Per-kernel roofline, profile set {tag}", "",
             f"`python3 bench.py --steps {bench['steps']} --warmup {bench['warmup']} --no-cpu-baseline` under rocprofv3 "
             f"(kernel trace; FETCH_SIZE and WRITE_SIZE in separate --pmc passes).  {steps} steps, "
             f"{total / steps:.1f} ms of kernel time per step.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE "
             f"(gfx950: FETCH_SIZE counts 64 B per 128-B request).  Peaks: HBM 8 TB/s (6.3 achievable), {mfma_name} "
             f"{mfma_peak:g} TFLOP/s.  alg. TFLOP/s = algorithmic FLOPs of the symbol's launches (SURVEY.md 8(d)) / its average launch time.", "",
             f"| kernel | launches/step | avg ms | ms/step | HBM GB/launch | HBM TB/s | % of 8 TB/s | alg. TFLOP/s | % of {mfma_name} |",
             "|---|---|---|---|---|---|---|---|---|"]
    bwd = {}                                           # attention backward: (dq + dkv) time per call, by self / cross
    for r in stats:
        ms_step = float(r["total_ms"]) / steps
        if ms_step < 0.05:
            continue
        k = r["kernel"]
        avg = float(r["avg_ms"])
        gb = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / 1e9 if k in fetch or k in write else None
        tbs = gb / avg if gb is not None else None             # GB / ms = TB/s
        fl = next((v for kk, v in sym_flops.items() if kk in k), None)
        t = fl / (avg * 1e-3) / 1e12 if fl else next((tf.get(v) for kk, v in group.items() if kk in k), None)
        m = re.search(r"attn_bwd_\w+<[^>]*\b(true|false)>\(", k)
        if m:
            d = bwd.setdefault("cross" if m.group(1) == "true" else "self", dict(ms=0.0, gb=0.0))
            d["ms"] += avg
            d["gb"] += gb or 0.0
        lines.append("| `{}` | {:.1f} | {:.3f} | {:.2f} | {} | {} | {} | {} | {} |".format(
            short(k)[:70], int(r["calls"]) / steps, avg, ms_step,
            f"{gb:.2f}" if gb is not None else "-", f"{tbs:.2f}" if tbs is not None else "-",
            f"{100 * tbs / HBM_PEAK_TBS:.0f}" if tbs is not None else "-",
            f"{t:.0f}" if t else "-", f"{100 * t / mfma_peak:.0f}" if t else "-"))
    fam = {k["kernel"]: k for k in (tfsrc.get("kernels") or [])}
    for which, d in sorted(bwd.items()):
        f = fam.get(f"attn_bwd_{which}")
        if f and "tflops" in f:
            fl = f["tflops"] * 1e12 * f["ms_per_step"] * 1e-3 / f["launches_per_step"]
            t = fl / (d["ms"] * 1e-3) / 1e12
            lines.append(f"| attention backward, {which} (the dQ + dK/dV kernels of one call together) | - | {d['ms']:.3f} | - | {d['gb']:.2f} | "
                         f"{d['gb'] / d['ms']:.2f} | {100 * d['gb'] / d['ms'] / HBM_PEAK_TBS:.0f} | {t:.0f} | {100 * t / mfma_peak:.0f} |")
    out = os.path.join(prof, f"{tag}_roofline.md")
    open(out, "w").write("\n".join(lines) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
