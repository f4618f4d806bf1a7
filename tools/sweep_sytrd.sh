#!/bin/bash
# eigen phase time vs the workgroup count / thread count of k_sytrd (G = 0: the default choice)
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
for n in ${NS:-500 1000}; do
  for G in ${GS:-0 32 48 64 80 96 128 160}; do
    for NT in ${NTS:-512}; do
      if [ $n = 1000 ]; then args="--n 1000 --p 20000 --m 256 --method perms"; else args="--n $n --p 20000 --m 512"; fi
      r=$(BLMM_SYTRD_G=$G BLMM_SYTRD_NT=$NT python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form $args --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['phases_ms']['eigen'],3))")
      echo "n=$n G=$G NT=$NT eigen_ms=$r"
    done
  done
done
