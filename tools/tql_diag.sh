#!/bin/bash
ROOT=$(pwd)
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make EXTRA=-DTQL_DIAG -j8 > /dev/null 2>&1; cd $ROOT
python3 tools/dbg_eig.py 2>&1 | grep -v -E "Warn|  j " | head -40
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make -j8 > /dev/null 2>&1
