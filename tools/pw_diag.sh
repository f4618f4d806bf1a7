#!/bin/bash
# when the workgroups of k_lr_panels_w start, how long their staging and their whole life take (diagnostic build)
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
ROOT=$(pwd)
cd bulklmm.jl_amd/csrc && touch kernels_lowrank.hip && make EXTRA=-DPW_DIAG -j8 > /dev/null 2>&1; cd $ROOT
BLMM_LR_SPLIT=${BLMM_LR_SPLIT:-1} python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-api "$@" 2>&1 | grep -E "panels_w diag|pw wg|ms_per_step" | cut -c1-400 | tail -24
cd bulklmm.jl_amd/csrc && touch kernels_lowrank.hip && make -j8 > /dev/null 2>&1
