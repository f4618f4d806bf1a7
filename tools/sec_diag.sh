#!/bin/bash
# iterations per root of the secular-equation kernel, per merge level (diagnostic build): tools/sec_diag.sh N
ROOT=$(pwd)
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make EXTRA=-DSEC_DIAG -j8 > /dev/null 2>&1; cd $ROOT
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-api --n ${1:-500} --p 2000 --m 512 2>&1 | grep "secular diag" | tail -8
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make -j8 > /dev/null 2>&1
