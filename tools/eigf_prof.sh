#!/bin/bash
# per-phase cycle counts of k_eigf_pairs and k_backtransform (the fast eigen path, kernels_eig.hip; BLMM_BT_STAGE=0: reflectors streamed) on the BXD kinship (diagnostic build -DEIGF_PROF)
export BLMM_DEV_ENV=1
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make EXTRA=-DEIGF_PROF -j8 > /dev/null 2>&1 && cd ../..
for st in 1 0; do
echo "== BLMM_BT_STAGE=$st"
BLMM_BT_STAGE=$st python3 - <<'PY'
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bulklmm_jl_amd as blmm
from common import bxd_kinship, make_data
K = bxd_kinship(); n = K.shape[0]
ctx = blmm.Context(0)
for _ in range(3):
    blmm.transform_rotation(np.eye(n), np.ones((n, 2)), K, addIntercept=False, ctx=ctx)
PY
done
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make -j8 > /dev/null 2>&1
