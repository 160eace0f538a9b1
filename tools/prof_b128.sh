# Kernel-trace summary of the per-GPU batch-128 step (the rank shape of the 8-GPU north-star point) next to the batch-1024 one:
# which kernels do not shrink with the batch.  usage (gpurun): bash tools/prof_b128.sh <tag>   -> gpurun_out/<tag>_b128_vs_b1024.txt
set -x
TAG=${1:-r04q}
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp
for b in 128 1024; do
  B="python3 $GRAFT_REPO_ROOT/bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary"
  $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_b$b -o p -- $B > $O/${TAG}_b${b}_bench.json 2>/dev/null
  rm -f $O/prof_${TAG}_b$b/*kernel_trace.csv $O/prof_${TAG}_b$b/*agent*
done
python3 - <<PY
import csv, json
def load(b):
    rows = list(csv.DictReader(open("$O/prof_${TAG}_b%d/p_kernel_stats.csv" % b)))
    return {r["Name"]: (float(r["TotalDurationNs"]) / 8e6, int(r["Calls"]) // 8, float(r["AverageNs"]) / 1e3) for r in rows}
a, c = load(128), load(1024)
out = ["kernel | ms/step B=128 | ms/step B=1024 / 8 | excess ms | avg us B=128 | avg us B=1024"]
tot = [0.0, 0.0]
for k, (ms, n, us) in sorted(a.items(), key=lambda kv: -(kv[1][0] - c.get(kv[0], (0, 0, 0))[0] / 8)):
    ms2, n2, us2 = c.get(k, (0.0, 0, 0.0))
    tot[0] += ms; tot[1] += ms2 / 8
    if ms > 0.02: out.append(f"{k[:100]} | {ms:.3f} | {ms2 / 8:.3f} | {ms - ms2 / 8:+.3f} | {us:.1f} | {us2:.1f}")
out.append(f"TOTAL | {tot[0]:.2f} | {tot[1]:.2f} | {tot[0] - tot[1]:+.2f}")
open("$O/${TAG}_b128_vs_b1024.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:30]))
PY
