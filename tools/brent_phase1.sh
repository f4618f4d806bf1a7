#!/bin/bash
# A/B of BRENT_PHASE1 (iterations of the first kernel of the split h2 search before the unfinished traits are handed to k_brent2):
# rebuild kernels_prep.o per value on the GPU box, two default bench runs each.
cd $GRAFT_REPO_ROOT
for it in ${*:-26 22 20 18 16 14 26}; do
  rm -f bulklmm.jl_amd/csrc/kernels_prep.o
  make -C bulklmm.jl_amd/csrc EXTRA="-DBRENT_PHASE1_IT=$it" > /dev/null 2>&1 || { echo build failed; exit 1; }
  for r in 1 2; do
    echo "PHASE1=$it $(python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['phases_ms'].items()})")"
  done
done
rm -f bulklmm.jl_amd/csrc/kernels_prep.o
make -C bulklmm.jl_amd/csrc > /dev/null 2>&1
