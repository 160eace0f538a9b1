#!/bin/bash
# Collect the per-round profile set on the GPU box (run through gpurun):
#   gpurun --timeout 1500 -- bash tools/collect_profiles.sh r01e
# 1 kernel-trace pass + separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ set), then a plain bench run.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- $B > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- $B > /dev/null 2> $R/gpurun_out/prof_${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- $B > /dev/null 2> $R/gpurun_out/prof_${TAG}_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- $B > /dev/null 2> $R/gpurun_out/prof_${TAG}_sq.err
cd $R
python3 tools/summarize_prof.py ${TAG} gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write
python3 tools/pmc_report.py gpurun_out/prof_${TAG}_sq > gpurun_out/${TAG}_pmc_sq.txt
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
python3 bench.py 2> gpurun_out/${TAG}_bench_default.err | tail -1 > gpurun_out/${TAG}_bench_default.json
tail -c 600 gpurun_out/${TAG}_bench_default.json
