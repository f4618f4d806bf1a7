"""Child process of tests/test_gpu_guard.py::test_fullsize_h2_audit_all_traits: the oracle's fitlmm (src/lmm.jl:56-86)
for every column of a rotated trait matrix, spread over a process pool.  It runs as its own program (started with
subprocess by the test, like tests/c_abi/c_abi_smoke.c) so that no worker is forked from a process that holds the GPU.
    python fitlmm_pool.py in.npz out.npy nproc"""
import multiprocessing as mp
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def fit_block(args):
    from oracle import bulklmm_oracle as O
    Y0, Z0, lam, prior = args
    out = np.empty((Y0.shape[1], 2))
    for j in range(Y0.shape[1]):
        est = O.fitlmm(Y0[:, j], Z0, lam, prior)
        out[j] = (est.h2, est.ell)
    return out


if __name__ == "__main__":
    z = np.load(sys.argv[1])
    nproc = int(sys.argv[3])
    Y0, Z0, lam, prior = z["Y0"], z["Z0"], z["lam"], list(z["prior"])
    blocks = np.array_split(np.arange(Y0.shape[1]), nproc * 4)
    with mp.get_context("spawn").Pool(nproc) as pool:
        res = np.vstack(pool.map(fit_block, [(np.ascontiguousarray(Y0[:, b]), Z0, lam, prior) for b in blocks]))
    np.save(sys.argv[2], res)
