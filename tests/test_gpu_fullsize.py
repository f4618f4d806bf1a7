"""Full-size GPU checks at BASELINE.json's BXD shape (n=79, p=7321, m=35554), through size-independent
properties plus the oracle on a sample of trait columns (the oracle needs ~10 ms per trait at p=7321)."""
import numpy as np
import pytest

from common import assert_h2_panel_ties_only, assert_lod_close, make_data
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
    # end to end, each side with its OWN h2 estimate, on 24 columns spread over the matrix: the north-star 1e-6 relative
    # wherever the two Brent runs agree on h2 to 1e-8 (they stop at x_tol ~ 1.5e-8 |x| on a likelihood that is flat to
    # rounding, so ~1e-7 is what two correct searches share); the measured worst case is printed
    more = list(range(100, M, M // 24))[:24]
    own2 = O.bulkscan_null(Y[:, more], G, K)
    dh = np.abs(own2.h2_null_list - exact.h2_null_list[more])
    rel = np.abs(exact.L[:, more] - own2.L) / np.maximum(np.abs(own2.L), 1e-4)
    close = dh <= 1e-8
    print(f"end to end at full size: max |dh2| {dh.max():.2e}; worst LOD rel error {rel.max():.2e} over {len(more)} traits, "
          f"{rel[:, close].max() if close.any() else 0.0:.2e} over the {int(close.sum())} with |dh2| <= 1e-8")
    assert dh.max() <= 1e-6
    if close.any():
        assert_lod_close(exact.L[:, np.asarray(more)[close]], own2.L[:, close], rtol=1e-6, atol=1e-9, what="LOD end to end")
    assert np.sum((exact.L[:, more] - own2.L) ** 2, axis=0).max() <= 1e-7
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


# ---- BASELINE.json configs[3] at its size: the 16-point heritability grid on n=79, p=7321, m=35554 ------------------------
GRID16 = [i / 16.0 for i in range(16)]


def test_fullsize_config3_null_grid_every_trait(blmm, bxd):
    """null-grid (the config's primary method) over the WHOLE trait matrix: the grid choice of every one of the 35,554 traits
    against the oracle's wls_multivar table (src/bulkscan_helpers.jl:267-269, find_optim_h2 :204-211; vectorised over the
    traits, so all of them are checked), LODs on sampled columns, and the ragged 4445 / 4439 shards of the 8-GPU run scanned
    alone (bit-identical: the sharding contract)."""
    Y, G, K = bxd
    g = blmm.bulkscan_null_grid(Y, G, K, GRID16)
    assert g.L.shape == (P, M) and np.isfinite(g.L).all() and (g.L >= -1e-9).all()
    Y0, X0, lam = O.transform_rotation(Y, G[:, :1], K)
    Ell = np.vstack([O.wls_multivar(Y0, X0[:, :1], O.makeweights(h, lam), [1.0, 0.0]).Ell for h in GRID16])   # 16 x M
    pick = np.argmax(Ell, axis=0)                       # first maximum wins
    ref_h2 = np.asarray(GRID16)[pick]
    bad = np.flatnonzero(g.h2_null_list != ref_h2)
    for j in bad:                                       # only a tie in the oracle's own table may resolve differently
        gi = GRID16.index(float(g.h2_null_list[j]))
        assert abs(Ell[gi, j] - Ell[pick[j], j]) <= 1e-12 * max(1.0, abs(Ell[pick[j], j])), (j, g.h2_null_list[j], ref_h2[j])
    print(f"configs[3] null-grid: {bad.size} of {M} grid choices differ from the oracle's (all ties)")
    assert bad.size <= 1e-4 * M
    cols = [0, 1, 63, 64, 4444, 4445, 17777, M - 4439, M - 2, M - 1]
    ref = O.bulkscan_null_grid(Y[:, cols], G, K, GRID16)
    assert np.array_equal(ref.h2_null_list, g.h2_null_list[cols])
    assert_lod_close(g.L[:, cols], ref.L)
    for r in (0, 7):
        lo, hi = blmm.trait_shard(M, r, 8)
        assert hi - lo == (4445 if r == 0 else 4439)
        part = blmm.bulkscan_null_grid(Y[:, lo:hi], G, K, GRID16)
        assert np.array_equal(part.L, g.L[:, lo:hi]) and np.array_equal(part.h2_null_list, g.h2_null_list[lo:hi])


def test_fullsize_config3_alt_grid_sampled_columns(blmm, bxd):
    """alt-grid (the config's secondary method) over the whole matrix -- L and h2_panel, 2 x 2.08 GB -- against
    O.bulkscan_alt_grid on sampled trait columns (every marker), the arg-max panel equal up to ties in the oracle's own logL1
    table, and the last ragged shard scanned alone."""
    Y, G, K = bxd
    a = blmm.bulkscan_alt_grid(Y, G, K, GRID16)
    assert a.L.shape == (P, M) and a.h2_panel.shape == (P, M) and np.isfinite(a.L).all()
    assert set(np.unique(a.h2_panel[:, ::97])).issubset(set(GRID16))
    cols = [0, 63, 64, 17777, M - 1]
    ref, tab = O.bulkscan_alt_grid(Y[:, cols], G, K, GRID16, return_tables=True)
    assert_lod_close(a.L[:, cols], ref.L, atol=1e-9)
    nt = assert_h2_panel_ties_only(a.h2_panel[:, cols], ref.h2_panel, tab, GRID16)
    print(f"configs[3] alt-grid: {nt} of {ref.h2_panel.size} sampled h2_panel entries differ from the oracle's (all ties)")
    assert nt <= 1e-3 * ref.h2_panel.size
    lo, hi = blmm.trait_shard(M, 7, 8)
    part = blmm.bulkscan_alt_grid(Y[:, lo:hi], G, K, GRID16)
    assert np.array_equal(part.L, a.L[:, lo:hi]) and np.array_equal(part.h2_panel, a.h2_panel[:, lo:hi])
