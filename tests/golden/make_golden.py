#!/usr/bin/env python3
"""Regenerates tests/golden/bulkscan_small.npz: seeded inputs (BXD kinship fixture of the reference's tests,
synthetic genotypes/traits) and the CPU oracle's outputs for every bulkscan method.  The reference itself is pure
Julia and cannot be run here, so these vectors come from the oracle, which is pinned by the
reference's own KATs (tests/test_oracle_kats.py).  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from common import make_data  # noqa: E402
from oracle import bulklmm_oracle as O  # noqa: E402

Y, G, K, Cov = make_data(n=79, p=64, m=48, seed=20240, ncov=1)
grid = [i / 10.0 for i in range(10)]
ex = O.bulkscan_null(Y, G, K, prior_variance=1.0, prior_sample_size=0.1)
exc = O.bulkscan_null(Y, G, K, Covar=Cov, reml=True)
gr = O.bulkscan_null_grid(Y, G, K, grid)
al = O.bulkscan_alt_grid(Y, G, K, grid)
pidx = O.make_perm_idx(79, 16, 3)
np.savez_compressed(
    os.path.join(HERE, "bulkscan_small.npz"), Y=Y, G=G, K=K, Cov=Cov, grid=np.array(grid),
    exact_L=ex.L, exact_h2=ex.h2_null_list, exact_cov_reml_L=exc.L, exact_cov_reml_h2=exc.h2_null_list,
    grid_L=gr.L, grid_h2=gr.h2_null_list, alt_L=al.L, alt_h2=al.h2_panel, perm_idx=pidx)
print("wrote bulkscan_small.npz")
