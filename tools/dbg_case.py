import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as blmm
import oracle.bulklmm_oracle as O
from common import make_data
seed0, case, n, p, m = 8, 119, 8, 63, 2
Y, G, K, Cov = make_data(n=n, p=p, m=m, seed=1000 + case + 7919 * seed0, ncov=0, bxd=False)
got = blmm.bulkscan_null(Y, G, K, optim_interval=2)
ref = O.bulkscan_null(Y, G, K, optim_interval=2)
print("h2 gpu", got.h2_null_list, "oracle", ref.h2_null_list)
d = got.L - ref.L
print("sum d^2 per trait", (d ** 2).sum(axis=0))
j = int(np.argmax((d ** 2).sum(axis=0)))
i = int(np.argmax(np.abs(d[:, j])))
print("worst marker", i, "gpu", got.L[i, j], "oracle", ref.L[i, j])
pin = O.bulkscan_null(Y, G, K, h2_override=got.h2_null_list)
print("vs oracle at gpu h2: max rel", np.nanmax(np.abs(got.L - pin.L) / np.maximum(np.abs(pin.L), 1e-300)), "max abs", np.abs(got.L - pin.L).max())
print("max LOD", np.nanmax(ref.L), "eig K", np.linalg.eigvalsh(K))
