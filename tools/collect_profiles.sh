#!/bin/bash
# Host: condense what tools/prof_r06.sh <tag> left under gpurun_out/ into the tracked files under profiles/:
#   bash tools/collect_profiles.sh r06d      (kernel stats, FETCH / WRITE PMC sets + sidecars, SQ counters, roofline and MFMA-busy tables)
set -e
cd "$(dirname "$0")/.."
T=$1
for f in f32_split3 f32_f32 bf16_f32; do
  [ -d gpurun_out/prof_${T}_${f}_stats ] || continue
  python tools/summarize_prof.py ${T}_$f gpurun_out/prof_${T}_${f}_stats gpurun_out/prof_${T}_${f}_fetch gpurun_out/prof_${T}_${f}_write > /dev/null 2>&1
  cp gpurun_out/${T}_${f}_pmc_meta.json gpurun_out/${T}_${f}_bench_under_rocprof.json profiles/
  python tools/pmc_report.py gpurun_out/prof_${T}_${f}_sq > profiles/${T}_${f}_pmc_sq.txt 2>&1
done
cp gpurun_out/${T}_bench_default.json profiles/
python tools/mfma_busy_table.py gpurun_out/prof_${T}_f32_split3_sq gpurun_out/prof_${T}_f32_f32_sq gpurun_out/prof_${T}_bf16_f32_sq 2>&1 \
  | sed "s/^## prof_${T}_/## ${T}_/; s/_sq\$//" > profiles/${T}_mfma_busy.md
python tools/roofline_table.py ${T}_f32_split3 profiles/${T}_bench_default.json > /dev/null
python tools/roofline_table.py ${T}_f32_f32 profiles/${T}_bench_default.json > /dev/null
python tools/roofline_table.py ${T}_bf16_f32 > /dev/null
ls profiles | grep "^${T}_"
