import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as B
from common import make_data
Y, G, K, Cov = make_data(n=125, p=1, m=2, seed=1000 + 63 + 7919 * 6, ncov=0, bxd=False)
print("K distinct values", np.unique(np.round(K, 12))[:10], "eigvals", np.round(np.linalg.eigvalsh(K)[[0, 1, 2, -3, -2, -1]], 6))
for eig in ("dc",):
    os.environ["BLMM_EIGEN"] = eig
    try:
        ctx = B.Context(0)
        r = B.transform_rotation(Y, np.hstack([np.ones((125, 1)), G]), K, addIntercept=False, ctx=ctx)
        lam = np.asarray(r[2])
        print(eig, "ok", np.abs(np.sort(lam) - np.linalg.eigvalsh(K)).max())
    except Exception as e:
        print(eig, "FAIL", repr(e)[:200])
