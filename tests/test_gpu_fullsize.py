"""Full-size GPU checks at BASELINE.json's BXD shape (n=79, p=7321, m=35554), through size-independent
properties plus the oracle on a sample of trait columns (the oracle needs ~10 ms per trait at p=7321)."""
import numpy as np
import pytest

from common import assert_lod_close, make_data
from oracle import bulklmm_oracle as O

pytestmark = pytest.mark.gpu

N, P, M = 79, 7321, 35554


@pytest.fixture(scope="module")
def bxd():
    return make_data(n=N, p=P, m=M, seed=20241)[:3]


@pytest.fixture(scope="module")
def exact(blmm, bxd):
    Y, G, K = bxd
    return blmm.bulkscan_null(Y, G, K)


def test_fullsize_null_exact_sampled_columns(blmm, bxd, exact):
    Y, G, K = bxd
    assert exact.L.shape == (P, M) and np.isfinite(exact.L).all() and (exact.L >= -1e-9).all()
    cols = [0, 1, 63, 64, 17777, M - 2, M - 1]
    ref = O.bulkscan_null(Y[:, cols], G, K, h2_override=exact.h2_null_list[cols])
    assert_lod_close(exact.L[:, cols], ref.L)
    own = O.bulkscan_null(Y[:, cols[:3]], G, K)
    assert np.abs(own.h2_null_list - exact.h2_null_list[cols[:3]]).max() <= 1e-6
    # independent RSS form (src/scan.jl:341-351) on one column
    s = O.scan(Y[:, 17777], G, K, prior_variance=1.0)
    assert np.sum((s["lod"] - exact.L[:, 17777]) ** 2) <= 1e-7


def test_fullsize_shard_consistency_and_trait_invariances(blmm, bxd, exact):
    """Traits are independent: scanning a column block alone gives bit-identical columns (the multi-GPU sharding
    contract), and with prior_sample_size = 0 the LODs are invariant to shifting and scaling a trait."""
    Y, G, K = bxd
    lo, hi = blmm.trait_shard(M, 3, 8)
    part = blmm.bulkscan_null(Y[:, lo:hi], G, K)
    assert np.array_equal(part.L, exact.L[:, lo:hi]) and np.array_equal(part.h2_null_list, exact.h2_null_list[lo:hi])
    sub = slice(1000, 1128)
    tr = blmm.bulkscan_null(3.5 * Y[:, sub] - 2.0, G, K)
    assert np.abs(tr.h2_null_list - exact.h2_null_list[sub]).max() <= 1e-6
    assert np.sum((tr.L - exact.L[:, sub]) ** 2, axis=0).max() <= 1e-7


def test_fullsize_duplicated_markers_and_grid(blmm, bxd):
    Y, G, K = bxd
    Ys = Y[:, :4096]
    grid = [i / 16.0 for i in range(16)]
    G2 = G.copy()
    G2[:, 5000] = G2[:, 11]  # a duplicated marker must get an identical LOD row
    g = blmm.bulkscan_null_grid(Ys, G2, K, grid)
    assert np.array_equal(g.L[5000], g.L[11])
    assert set(np.unique(g.h2_null_list)).issubset(set(grid))
    cols = [0, 2047, 4095]
    ref = O.bulkscan_null_grid(Ys[:, cols], G2, K, grid)
    assert np.array_equal(ref.h2_null_list, g.h2_null_list[cols])
    assert_lod_close(g.L[:, cols], ref.L)
