# Launch-gap probe for the small-batch step (run on the GPU box): rocprofv3 kernel trace of N steps, then
# sum(kernel durations) per step against the wall time per step that bench.py reports.
# usage: bash tools/gap_probe.sh <tag> <bench args...>
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out
cd /tmp
export GAMER_WGRAD_TUNE_FILE=$O/${TAG}_wgrad_tune.json
python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary > /dev/null 2>&1
python3 $R/bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/${TAG}_plain.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -o p -- python3 $R/bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/${TAG}_under_rocprof.json 2>/dev/null
cd $R
python3 - <<PY
import csv, glob, json
f = glob.glob("$O/prof_${TAG}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
calls = sum(int(r["Calls"]) for r in rows)
plain = json.load(open("$O/${TAG}_plain.json"))
steps = 23
print(f"kernel time {tot / steps:.2f} ms/step over {calls / steps:.0f} launches/step; wall {plain['ms_per_step']:.2f} ms/step (plain run)")
for r in rows[:25]:
    print(f"  {r['Name'][:70]:70s} n/step={int(r['Calls']) / steps:6.1f} avg_us={float(r['AverageNs']) / 1e3:8.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:6.2f}")
PY
rm -rf $O/prof_${TAG}
