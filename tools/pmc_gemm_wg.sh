# HBM bytes (FETCH_SIZE, WRITE_SIZE) and SQ counters of the weight-gradient kernels of tools/dev_gemm_wg.py (run on the GPU box):
#   bash tools/pmc_gemm_wg.sh "<shapes>" "<chunks>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_wg; rm -rf $O; mkdir -p $O
export CHUNKS=${2:-3104}
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  t=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$t -o p -- python3 $R/tools/dev_gemm_wg.py 1024 $1 > $O/$t.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "gemm" not in k and "wgrad" not in k: continue
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
