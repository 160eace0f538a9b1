# MFMA-busy / VALU instruction counters of the fp32, fp32/split6 and bf16 steps (one --pmc pass each, kernel trace only)
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp
export GAMER_WGRAD_TUNE_FILE=$O/r02z_wgrad_tune.json
for cfg in "f32 f32" "f32 split6" "bf16 f32"; do
  set -- $cfg; tag=${1}_${2}
  python3 $GRAFT_REPO_ROOT/bench.py --dtype $1 --matmul $2 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prof_${tag}_sq -o p -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $1 --matmul $2 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $O/prof_${tag}_sq.err
  python3 $GRAFT_REPO_ROOT/tools/pmc_report.py $O/prof_${tag}_sq > $O/r02z_${tag}_pmc_sq.txt 2>&1
  head -20 $O/r02z_${tag}_pmc_sq.txt
done
