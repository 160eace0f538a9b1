# Per-kernel times (rocprofv3 kernel trace) of the fp32-MFMA and the split attention kernels on one (batch, 505) problem:
#   bash tools/prof_attn_split.sh [batch]
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_attn_split_driver.py ${1:-256} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ps/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attn" in r["Name"]:
        print(f"{r['Name'][:60]:60s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
