#!/bin/bash
# eigen phase of one bulkscan against the leaf size of the divide-and-conquer eigensolver (BLMM_EIG_LEAF), over n
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
for n in 130 200 300 500 700 1000 1400; do
  for leaf in 6 8 12 16 24 32; do
    BLMM_EIG_LEAF=$leaf python3 bench.py --no-cpu-baseline --no-host-api --n $n --p 2000 --m 512 --steps 10 --warmup 2 > /tmp/sl.json 2>/tmp/sl.err || { tail -3 /tmp/sl.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/sl.json')); print('n', $n, 'leaf', $leaf, 'eigen ms', round(d['phases_ms']['eigen'],4))"
  done
done
