#!/bin/bash
# A/B the k_brent / k_brent2 build knobs on the GPU box: rebuild kernels_prep.o per variant, run the default bench.
cd $GRAFT_REPO_ROOT
for v in "3 1 2 5" "3 1 2 1" "3 1 2 2" "3 2 2 5" "2 5 2 5" "3 1 1 5"; do
  set -- $v
  rm -f bulklmm.jl_amd/csrc/kernels_prep.o
  make -C bulklmm.jl_amd/csrc EXTRA="-DBRENT_MINW=$1 -DBRENT_UNROLL=$2 -DBRENT_MINW2=$3 -DBRENT_UNROLL2=$4" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "MINW=$1 UNROLL=$2 MINW2=$3 UNROLL2=$4 $(python3 bench.py --no-cpu-baseline --steps 20 2>/dev/null | grep -o '"h2": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')"
done
rm -f bulklmm.jl_amd/csrc/kernels_prep.o
