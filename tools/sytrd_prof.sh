#!/bin/bash
# per-phase cycle counts of k_sytrd (s_memtime stamps; diagnostic build): tools/sytrd_prof.sh
export BLMM_DEV_ENV=1   # the BLMM_* switches below are developer switches: the library reads them only with this set
set -o pipefail
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make EXTRA=-DSYTRD_PROF -j8 > /dev/null 2>&1 && cd ../..
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bulklmm_jl_amd as B
for n in (79, 500, 1000):
    os.environ["BLMM_EIGEN"] = "dc"
    rng = np.random.default_rng(n)
    X = rng.random((n, 2 * n)) - 0.5
    K = 2 * X @ X.T / X.shape[1] + 0.5
    for nt in ("256", "512"):
        os.environ["BLMM_SYTRD_NT"] = nt
        B.transform_rotation(np.eye(n)[:, :2], np.ones((n, 2)), K)
        B.transform_rotation(np.eye(n)[:, :2], np.ones((n, 2)), K)
PY
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bulklmm_jl_amd as B
from common import bxd_kinship
K = bxd_kinship()
for _ in range(2):
    B.transform_rotation(np.eye(79)[:, :2], np.ones((79, 2)), K)
PY
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make -j8 > /dev/null 2>&1
