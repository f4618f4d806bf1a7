#!/bin/bash
# A/B of run-time switches on one box: tools/env_sweep.sh <rounds> "<bench.py arguments>" "VAR=1 VAR2=3" "VAR=2" ...
# ("-" = no extra environment); round-robin so that clock drift hits every setting alike
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e=""
    env $e python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form $ARGS > /tmp/es.json 2> /tmp/es.err || { tail -3 /tmp/es.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/es.json')); print('[$v]', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phases_ms'].items()}, flush=True)"
  done
done
