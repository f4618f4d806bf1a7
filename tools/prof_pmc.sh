#!/bin/bash
# PMC pass for one kernel: tools/prof_pmc.sh <tag> <kernel-regex> "<counters>" [bench args...]
set -o pipefail
TAG=$1; KRE=$2; CTR=$3; shift 3
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $CTR --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - <<PY
import csv,collections,glob
f=glob.glob("$OUT/pmc/*counter_collection.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items(): print(k, sum(v)/len(v))
PY
