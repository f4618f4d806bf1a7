#!/bin/bash
# tools/ab_env.sh OUT ROUNDS "ENVA" "ENVB" [bench args...]: alternate bench.py runs under two environments on one box,
# ms_per_step of each run into OUT (A/B testing of run-time switches without rebuilding)
out=$1; rounds=$2; ea=$3; eb=$4; shift 4
: > "$out"
for r in $(seq 1 "$rounds"); do
  for tag in A B; do
    if [ $tag = A ]; then e=$ea; else e=$eb; fi
    v=$(env $e python bench.py --no-host-api --no-cpu-baseline --no-all-rank-form --steps 40 "$@" 2>>"$out.err" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d.get('phases_ms')))") || exit 1
    echo "$r $tag [$e] $v" >> "$out"
  done
done
