# Final round-2 measurement pass (run on the GPU box through gpurun): bench records of every configuration quoted in
# DESIGN.md / README.md, rocprofv3 kernel-trace summaries of the fp32, bf16 and fp32/split6 steps, and the HBM byte counters
# (separate --pmc passes, FETCH_SIZE / WRITE_SIZE) of the fp32 and bf16 steps.  Outputs under gpurun_out/r02z_*.
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r02z_bench_default.json 2> $O/r02z_bench_default.err; tail -2 $O/r02z_bench_default.err
python bench.py --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02z_bench_batch128.json 2>/dev/null
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02z_bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02z_bench_bf16_batch128.json 2>/dev/null
python bench.py --matmul split6 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02z_bench_split6.json 2>/dev/null
python bench.py --matmul split6 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02z_bench_split6_batch128.json 2>/dev/null
python bench.py --path module --steps 10 --warmup 3 --no-cpu-baseline > $O/r02z_bench_module.json 2>/dev/null
python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline > $O/r02z_bench_ragged.json 2>/dev/null
python bench.py --variant session --steps 10 --warmup 3 --no-cpu-baseline > $O/r02z_bench_session.json 2>/dev/null
export TMPDIR=/tmp; cd /tmp
# the wgrad chunk sweep of a first step would sit in the profiled kernel statistics: measure once, reuse in the profiled runs
export GAMER_WGRAD_TUNE_FILE=$O/r02z_wgrad_tune.json
for cfg in "f32 f32" "bf16 f32" "f32 split6"; do set -- $cfg; python3 $GRAFT_REPO_ROOT/bench.py --dtype $1 --matmul $2 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1; done
for cfg in "f32 f32" "bf16 f32" "f32 split6"; do
  set -- $cfg; dt=$1; mm=$2; tag=${dt}_${mm}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --matmul $mm --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/r02z_bench_${tag}_under_rocprof.json 2>/dev/null
done
for dt in f32 bf16; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_${dt}_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_${dt}_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
done
cd $O; rm -f prof_*_stats/*kernel_trace.csv prof_*_stats/*agent*; ls prof_*/ | head -40; du -sh .
