#!/bin/bash
# per-kernel average durations of a bench.py run under a set of environment switches:
#   tools/kstats.sh <tag> "<VAR=1 ...>|-" <bench.py arguments...>     (rocprofv3 --kernel-trace --stats; prints the top kernels)
TAG=$1; E=$2; shift 2
[ "$E" = "-" ] && E=""
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in $E; do export $kv; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 10 --warmup 2 "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
print("== $TAG [$E]")
for r in rows[:12]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
