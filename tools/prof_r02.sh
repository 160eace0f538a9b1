set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_torch_ops.py -q -x 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r02c_bench_default.json 2> gpurun_out/r02c_bench_default.err; tail -3 gpurun_out/r02c_bench_default.err
python bench.py --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_batch128.json 2>/dev/null
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_bf16_batch128.json 2>/dev/null
python bench.py --path module --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02c_bench_module.json 2>/dev/null
python bench.py --path module-fused --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02c_bench_module_fused.json 2>/dev/null
export TMPDIR=/tmp; cd /tmp
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${dt}_stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02c_bench_${dt}_under_rocprof.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${dt}_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${dt}_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT/gpurun_out; rm -f prof_*/*kernel_trace.csv prof_*_stats/*agent* ; ls -la prof_*/ | head -40; du -sh .
