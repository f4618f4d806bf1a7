import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as blmm
import oracle.bulklmm_oracle as O
from common import make_data
n = 125
for sd in (35, 36, 37, 38, 40):
    gk = (np.random.default_rng(sd).random((n, 3)) < 0.5).astype(np.float64)
    K = O.calcKinship(gk[:, :1])
    ctx = blmm.Context(0)
    Ut, X0, lam = blmm.transform_rotation(np.eye(n), np.ones((n, 2)), K, addIntercept=False, ctx=ctx)
    U = np.asarray(Ut).T
    lam = np.asarray(lam)
    print(sd, "orth", np.abs(U.T @ U - np.eye(n)).max(), "resid", np.abs(K - (U * lam) @ U.T).max(), "eig err", np.abs(np.sort(lam) - np.linalg.eigvalsh(K)).max())
    ctx.close()
