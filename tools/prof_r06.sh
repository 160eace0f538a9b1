# Round-6 measurement pass (the passes of tools/prof_r05.sh; the PMC sidecar records when the set was made, which is what bench.py orders sets by) (run on the GPU box through gpurun): the driver's default bench line, then rocprofv3 kernel-trace
# summaries and HBM byte counters (separate --pmc passes: FETCH_SIZE, WRITE_SIZE; SQ set) of the default (fp32 / split3) step
# and of the fp32-MFMA and bf16 steps.  Every PMC set gets a <tag>_pmc_meta.json naming the profiled command and workload.
# usage: bash tools/prof_r06.sh <tag> ["f32 split3" "f32 f32" ...]     (outputs under gpurun_out/<tag>_*; default: the three sets)
set -x
TAG=${1:-r06z}
shift
if [ $# -eq 0 ]; then set -- "f32 split3" "f32 f32" "bf16 f32"; fi
CFGS=("$@")
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
python bench.py --steps 20 --warmup 5 --kernel-rows 60 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -2 $O/${TAG}_bench_default.err
export TMPDIR=/tmp; cd /tmp
export GAMER_WGRAD_TUNE_FILE=$O/${TAG}_wgrad_tune.json
for cfg in "${CFGS[@]}"; do set -- $cfg; python3 $GRAFT_REPO_ROOT/bench.py --dtype $1 --matmul $2 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary > /dev/null 2>&1; done
for cfg in "${CFGS[@]}"; do
  set -- $cfg; dt=$1; mm=$2; tag=${dt}_${mm}
  B="python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --matmul $mm --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_${tag}_stats -o p -- $B > $O/${TAG}_${tag}_bench_under_rocprof.json 2>/dev/null
  B2="python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --matmul $mm --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_${TAG}_${tag}_fetch -o p -- $B2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_${TAG}_${tag}_write -o p -- $B2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prof_${TAG}_${tag}_sq -o p -- $B2 > /dev/null 2>&1
  echo "{\"created\": \"$(date -u +%Y-%m-%dT%H:%M:%SZ)\", \"cmd\": \"$B2 (tools/prof_r06.sh)\", \"batch\": 1024, \"items\": 101, \"dtype\": \"$dt\", \"matmul\": \"$mm\", \"variant\": \"multi\", \"ragged\": false}" > $O/${TAG}_${tag}_pmc_meta.json
done
cd $O; rm -f prof_*_stats/*kernel_trace.csv prof_*/*agent*; du -sh .
