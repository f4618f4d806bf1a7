#!/bin/bash
# the tridiagonal matrix k_eigf_reduce (small_sytrd) produces, against LAPACK's eigenvalues of the input (diagnostic build -DSMALL_SYTRD_DBG)
export BLMM_DEV_ENV=1
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make EXTRA=-DSMALL_SYTRD_DBG -j8 > /dev/null 2>&1 && cd ../..
for n in ${NS:-4 6 13 70}; do
python3 - $n > /tmp/dbg_sy.out 2>&1 <<'PY'
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bulklmm_jl_amd as blmm
n = int(sys.argv[1])
rng = np.random.default_rng(n)
X = rng.standard_normal((n, 2 * n)); K = X @ X.T / n
np.save("/tmp/dbg_K.npy", K)
ctx = blmm.Context(0)
blmm.transform_rotation(np.eye(n), np.ones((n, 2)), K, addIntercept=False, ctx=ctx)
PY
python3 - $n <<'PY'
import sys, re, numpy as np
n = int(sys.argv[1]); K = np.load("/tmp/dbg_K.npy")
d = np.zeros(n); e = np.zeros(n)
for line in open("/tmp/dbg_sy.out"):
    m = re.match(r"sytrd_dbg n (\d+) j (\d+) d (\S+) e (\S+)", line)
    if m and int(m.group(1)) == n: d[int(m.group(2))] = float(m.group(3)); e[int(m.group(2))] = float(m.group(4))
T = np.diag(d) + np.diag(e[:-1], 1) + np.diag(e[:-1], -1)
print("n", n, "max |eig(T) - eig(K)|", np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(K)).max(), "trace diff", abs(d.sum() - np.trace(K)))
if n <= 6: print(np.round(d, 6), np.round(e, 6))
PY
done
cd bulklmm.jl_amd/csrc && touch kernels_eig.hip && make -j8 > /dev/null 2>&1
