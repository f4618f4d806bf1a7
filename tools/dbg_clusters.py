import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as blmm
from bulklmm_jl_amd import api as A, _lib as L
rng = np.random.default_rng(3)
n = 64
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
lam = np.repeat(rng.uniform(0.1, 10.0, 4), 16)
K = (Q * lam) @ Q.T; K = (K + K.T) / 2
ctx = blmm.Context(0)
Ut, _, lam_d = blmm.transform_rotation(np.eye(n), np.ones((n, 2)), K, addIntercept=False, ctx=ctx)
U = np.asarray(Ut).T; lam_d = np.asarray(lam_d)
M = U.T @ K @ U
off = M - np.diag(np.diag(M))
print("stop2", os.environ.get("BLMM_JAC_STOP2"), "resid", np.abs(K - (U * lam_d) @ U.T).max() / np.abs(K).max(), "max off of U'KU", np.abs(off).max(), "orth", np.abs(U.T @ U - np.eye(n)).max())
i, j = np.unravel_index(np.argmax(np.abs(off)), off.shape)
print("worst pair", i, j, "diag", M[i, i], M[j, j], "lam", lam_d[i], lam_d[j])
