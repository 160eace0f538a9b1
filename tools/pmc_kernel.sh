#!/bin/bash
# usage: tools/pmc_kernel.sh <out-tag> <kbench args...>   (run on the GPU box through gpurun)
# Collects SQ counters for the kernels of a kbench run in two passes (8 SQ slots per pass).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 tools/kbench.py "$@" > /dev/null 2> gpurun_out/pmc_${TAG}_a.err || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 tools/kbench.py "$@" > /dev/null 2> gpurun_out/pmc_${TAG}_b.err || true
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -- python3 tools/kbench.py "$@" > /dev/null 2> gpurun_out/pmc_${TAG}_c.err || true
python3 tools/pmc_report.py gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b gpurun_out/pmc_${TAG}_c
