#!/bin/bash
# per-wave cycle stamps of one round of k_jacobi_lds (s_memtime; diagnostic build): tools/jac_prof.sh
set -o pipefail
cd bulklmm.jl_amd/csrc && touch kernels_prep.hip && make EXTRA=-DJAC_PROF -j8 > /dev/null 2>&1 && cd ../..
python3 tools/jacobi_probe.py 2>&1 | head -60
cd bulklmm.jl_amd/csrc && touch kernels_prep.hip && make -j8 > /dev/null 2>&1
