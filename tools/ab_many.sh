#!/bin/bash
# A/B/C.. of prebuilt library variants on one box: tools/ab_many.sh <rounds> "<bench.py arguments>" ab_libs/a.so ab_libs/b.so ...
# (each copied over the in-tree library in turn, round-robin, so that clock drift hits every variant alike)
LIB=bulklmm.jl_amd/csrc/libbulklmm_hip.so
cp $LIB /tmp/lib_orig.so
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for v in "$@"; do
    cp $v $LIB
    python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form $ARGS > /tmp/ab.json 2> /tmp/ab.err || { tail -3 /tmp/ab.err; cp /tmp/lib_orig.so $LIB; exit 1; }
    grep "lr phase\|lr diag" /tmp/ab.err | tail -4
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phases_ms'].items()}, flush=True)"
  done
done
cp /tmp/lib_orig.so $LIB
