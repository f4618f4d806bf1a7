#!/bin/bash
# per-class workgroup durations and per-XCD end times of k_scan_lr (diagnostic build): tools/lr_diag.sh
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
set -o pipefail
ROOT=$(pwd)
cd bulklmm.jl_amd/csrc && touch kernels_scan.hip && make EXTRA=-DLR_DIAG -j8 > /dev/null 2>&1; cd $ROOT
for t in 0 1e-13 8.4 1e300; do
  echo "BLMM_LR_TOL=$t"
  if [ "$t" = "0" ]; then export BLMM_LR_SHARED=0; t=1e-13; else unset BLMM_LR_SHARED; fi
  BLMM_LR_TOL=$t python3 bench.py --steps 3 --warmup 1 --no-host-api 2>&1 | grep "lr diag" | tail -2
done
cd bulklmm.jl_amd/csrc && touch kernels_scan.hip && make -j8 > /dev/null 2>&1
