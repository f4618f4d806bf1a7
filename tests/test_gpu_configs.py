"""BASELINE.json configs[2] and configs[4] at the size ONE of the 8 ranks runs, through whatever code path the library
picks by default (the suite sets no BLMM_* variable).

  configs[2]  synthetic large eQTL   n=500,  p=50000,  m=20000 / 8 = 2500 traits, fp64: null-exact and null-grid
  configs[4]  permutation test       n=1000, p=100000, 10000 / 8 = 1250 permutations, fp32 and fp64

The oracle needs ~0.5 s per trait at p=50000, so it checks sampled trait columns (every marker) for configs[2] and
sampled markers (every permutation) for configs[4]; the rest is covered by size-independent properties: a column block
scanned alone is bit-identical (the multi-GPU sharding contract), a duplicated marker gets an identical row, everything
is finite and non-negative.  Reference: src/bulkscan.jl:212-314, src/bulkscan_helpers.jl:239-292, src/scan.jl:485-557."""
import numpy as np
import pytest

from common import assert_lod_close, make_data
from oracle import bulklmm_oracle as O

pytestmark = pytest.mark.gpu


def MARKERS(p):
    """Sampled markers for the oracle: both ends (ragged tile edges) and every 97th in between."""
    return np.unique(np.concatenate([np.arange(0, 300), np.arange(p - 300, p), np.arange(300, p - 300, 97)]))


@pytest.fixture(scope="module")
def large():
    """configs[2] shard: n=500, p=50000, m=2500."""
    Y, G, K, _ = make_data(n=500, p=50000, m=2500, seed=20242, bxd=False)
    G = G.copy()
    G[:, 40001] = G[:, 17]          # a duplicated marker
    return Y, G, K


def test_config2_null_exact_shard(blmm, large):
    Y, G, K = large
    n, m = Y.shape
    p = G.shape[1]
    L, h2, st = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1,
                                        "eigen", 0, None, return_status=True)
    assert L.shape == (p, m) and np.isfinite(L).all() and (L >= -1e-9).all()
    assert st.lowrank_rank > 0 and st.n_nan_lod == 0
    assert st.lowrank_resid <= 1e-12 or st.lowrank_fallback > 0   # the guard: large residuals are re-scanned full rank
    assert np.array_equal(L[40001], L[17])
    # the oracle on sampled traits x sampled markers (a marker's LOD depends on no other marker; K stays the full one)
    cols = [0, 1, 63, 64, 1279, m - 2, m - 1]
    idx = MARKERS(p)
    ref = O.bulkscan_null(Y[:, cols], G[:, idx], K, h2_override=h2[cols])
    assert_lod_close(L[np.ix_(idx, cols)], ref.L)
    own = O.bulkscan_null(Y[:, cols[:2]], G[:, idx], K)
    assert np.abs(own.h2_null_list - h2[cols[:2]]).max() <= 1e-6
    assert np.sum((own.L - L[np.ix_(idx, cols[:2])]) ** 2, axis=0).max() <= 1e-7
    # shard contract: a column block scanned alone is bit-identical
    lo, hi = blmm.trait_shard(m, 2, 5)
    part = blmm.bulkscan_null(Y[:, lo:hi], G, K)
    assert np.array_equal(part.L, L[:, lo:hi]) and np.array_equal(part.h2_null_list, h2[lo:hi])


def test_config2_null_grid_shard(blmm, large):
    Y, G, K = large
    n, m = Y.shape
    grid = [i / 10.0 for i in range(10)]
    g = blmm.bulkscan_null_grid(Y, G, K, grid)
    assert np.isfinite(g.L).all() and (g.L >= -1e-9).all()
    assert set(np.unique(g.h2_null_list)).issubset(set(grid))
    assert np.array_equal(g.L[40001], g.L[17])
    cols = [0, 127, 128, 2047, m - 1]
    idx = MARKERS(G.shape[1])
    ref = O.bulkscan_null_grid(Y[:, cols], G[:, idx], K, grid)
    assert np.array_equal(ref.h2_null_list, g.h2_null_list[cols])
    assert_lod_close(g.L[np.ix_(idx, cols)], ref.L)
    lo, hi = blmm.trait_shard(m, 7, 8)
    part = blmm.bulkscan_null_grid(Y[:, lo:hi], G, K, grid)
    assert np.array_equal(part.L, g.L[:, lo:hi])


@pytest.fixture(scope="module")
def perm_cfg():
    """configs[4] shard: n=1000, p=100000, one trait, 1250 permutations."""
    Y, G, K, _ = make_data(n=1000, p=100000, m=1, seed=20244, bxd=False)
    pidx = O.make_perm_idx(1000, 1250, 44)
    return Y[:, 0].copy(), G, K, pidx


def test_config4_permutations_fp64_and_fp32(blmm, perm_cfg):
    y, G, K, pidx = perm_cfg
    n, p = G.shape
    nperms = pidx.shape[1]
    g64 = blmm.scan(y, G, K, permutation_test=True, nperms=nperms, perm_idx=pidx)
    g32 = blmm.scan(y, G, K, permutation_test=True, nperms=nperms, perm_idx=pidx, perm_precision="f32")
    for r in (g64, g32):
        assert r["L_perms"].shape == (p, nperms) and np.isfinite(r["L_perms"]).all() and (r["L_perms"] >= -1e-4).all()
        assert np.isfinite(r["lod"]).all()
    assert g32["L_perms"].dtype == np.float32 and g64["L_perms"].dtype == np.float64
    # the null model stays fp64; the original trait's LOD keeps an fp64 numerator, its marker norms come from the fp32-rotated
    # markers (round 4: the rotation runs on the fp32 matrix cores too): 1e-6, not bit for bit
    assert g32["h2_null"] == g64["h2_null"]
    assert_lod_close(g32["lod"], g64["lod"], what="lod of the fp32 permutation path")
    # the oracle on sampled markers (a marker's LOD row depends on no other marker), all permutations, shared rotation
    idx = MARKERS(p)
    y0, X0, lam = blmm.transform_rotation(y.reshape(-1, 1), np.hstack([np.ones((n, 1)), G]), K, addIntercept=False)
    X0s = np.hstack([X0[:, :1], X0[:, 1 + idx]])
    del X0
    pin = O.scan(y, G[:, idx], K, covar=np.ones((n, 1)), addIntercept=False, permutation_test=True, nperms=nperms,
                 perm_idx=pidx, h2_override=g64["h2_null"], rotation_override=(y0, X0s, lam))
    assert_lod_close(g64["lod"][idx], pin["lod"])
    assert_lod_close(g64["L_perms"][idx], pin["L_perms"])
    ref = pin["L_perms"]
    err = np.abs(g32["L_perms"][idx].astype(np.float64) - ref)
    assert np.all(err <= 1e-3 * np.abs(ref) + 1e-4), float(err.max())
    # own h2 against the oracle's Brent on its own LAPACK rotation
    own = O.scan(y, G[:, :64], K)
    assert abs(own["h2_null"] - g64["h2_null"]) <= 1e-6
    assert abs(own["sigma2_e"] - g64["sigma2_e"]) <= 1e-6 * own["sigma2_e"]
    # permutation shards are independent columns: a block of them alone is bit-identical
    part = blmm.scan(y, G, K, permutation_test=True, nperms=300, perm_idx=pidx[:, 500:800])
    assert np.array_equal(part["L_perms"], g64["L_perms"][:, 500:800])
    part32 = blmm.scan(y, G, K, permutation_test=True, nperms=300, perm_idx=pidx[:, 500:800], perm_precision="f32")
    assert np.array_equal(part32["L_perms"], g32["L_perms"][:, 500:800])


def test_sharded_marker_rotation_equals_the_replicated_one():
    """One process per GPU: every rank prepares (eigen replicated), rotates ITS marker block, the blocks are all-gathered, and
    the scan takes the gathered blocks (blmm_prepare_dev / blmm_rotate_block_dev / blmm_bulkscan_prerotated_dev) -- at n = 500
    the replicated rotation of G is 0.75 ms of a rank's 6.5 ms step.  Rehearsed with R = 3 contexts on the one GPU, the
    all-gather emulated by writing the blocks into one buffer: every rank's LOD block must equal blmm_bulkscan_dev's bits
    (null-exact, null-grid and alt-grid), and the same for the permutation test (blmm_scan_perms_prerotated_dev, fp64 and fp32,
    the permutations sharded over the ranks).  Own program: torch has to initialise the GPU before the library does."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    run = subprocess.run([sys.executable, os.path.join(here, "helpers", "sharded_rotation_check.py")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert run.stdout.count("sharded rotation ok") == 5
