#!/bin/bash
# usage (GPU box): bash tools/pmc_attn_h2.sh <tag> [B]   - SQ counters of the three-product attention kernels (two passes of 8 SQ slots)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; B=${2:-256}
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 tools/time_attn_h2.py $B > /dev/null 2> gpurun_out/pmc_${TAG}_a.err || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 tools/time_attn_h2.py $B > /dev/null 2> gpurun_out/pmc_${TAG}_b.err || true
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -- python3 tools/time_attn_h2.py $B > /dev/null 2> gpurun_out/pmc_${TAG}_c.err || true
python3 tools/pmc_report.py gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b gpurun_out/pmc_${TAG}_c | grep -A30 "attn_" 
