#!/bin/bash
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
cd $GRAFT_REPO_ROOT
for n in 130 200 333; do
  for e in jacobi rocsolver; do
    echo "n=$n $e $(BLMM_EIGEN=$e python3 bench.py --no-cpu-baseline --n $n --p 2000 --m 2000 --steps 3 --warmup 2 2>/dev/null | grep -o '"eigen": [0-9.]*')"
  done
done
