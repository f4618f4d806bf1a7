#!/bin/bash
# kernel-trace stats of bench.py for an arbitrary configuration: tools/prof_cfg.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
cut -d, -f1-5 $OUT/kernel_stats.csv | cut -c1-150
