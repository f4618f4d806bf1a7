#!/bin/bash
# A/B of prebuilt library variants on one box: tools/ab_libs.sh ab_libs/a.so ab_libs/b.so [rounds [bench.py arguments...]]  (each copied over the in-tree library in turn)
LIB=bulklmm.jl_amd/csrc/libbulklmm_hip.so
cp $LIB /tmp/lib_orig.so
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq $R); do
  for v in $A $B; do
    cp $v $LIB
    python3 bench.py --no-cpu-baseline --no-host-api "$@" > /tmp/ab.json 2> /tmp/ab.err || { tail -3 /tmp/ab.err; cp /tmp/lib_orig.so $LIB; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phases_ms'].items()})"
  done
done
cp /tmp/lib_orig.so $LIB
