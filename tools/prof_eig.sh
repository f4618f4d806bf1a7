#!/bin/bash
# kernel-trace stats of the eigensolver phases at n = 500 / 1000 / 79(dc): tools/prof_eig.sh <tag>
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
set -o pipefail
TAG=${1:-eig}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "500 50000 2500 null-exact" "1000 100000 1250 perms"; do
  set -- $cfg
  python3 bench.py --no-cpu-baseline --no-host-api --n $1 --p $2 --m $3 --method $4 --steps 5 --warmup 2 > $OUT/bench_n$1.json 2>$OUT/bench_n$1.err || { tail -3 $OUT/bench_n$1.err; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_n$1 -o trace -- python3 bench.py --no-cpu-baseline --no-host-api --n $1 --p $2 --m $3 --method $4 --steps 5 --warmup 1 > $OUT/trace_n$1.log 2>&1 || { tail -5 $OUT/trace_n$1.log; exit 1; }
  f=$(find $OUT/trace_n$1 -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_n$1.csv
  echo "== n=$1"; python3 -c "import json;j=json.loads(open('$OUT/bench_n$1.json').read().strip().splitlines()[-1]);print(round(j['ms_per_step'],3),{k:round(v,3) for k,v in j['phases_ms'].items()})"
  cut -d, -f1-4 $OUT/kernel_stats_n$1.csv | head -24 | cut -c1-110
done
BLMM_EIGEN=dc python3 bench.py --no-cpu-baseline --no-host-api > $OUT/bench_n79dc.json 2>$OUT/bench_n79dc.err
python3 -c "import json;j=json.loads(open('$OUT/bench_n79dc.json').read().strip().splitlines()[-1]);print('n=79 dc',round(j['ms_per_step'],3),{k:round(v,3) for k,v in j['phases_ms'].items()})"
BLMM_EIGEN=dc rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_n79 -o trace -- python3 bench.py --no-cpu-baseline --no-host-api --steps 5 --warmup 1 > $OUT/trace_n79.log 2>&1
f=$(find $OUT/trace_n79 -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_n79.csv; cut -d, -f1-4 $OUT/kernel_stats_n79.csv | head -24 | cut -c1-110
