"""The null-exact LOD kernel runs a low-rank form of the per-trait weights (kernels_lowrank.hip).  These tests pin its
guard: every trait's expansion residual is measured on the device, traits above 1e-13 are re-scanned from the full-length
sums (k_scan_fix), and blmm_status says how many.  Adversarial spectra and heritabilities, the re-scan kernel compared
with the oracle as a whole (tuning lr_tol = 0 flags every trait), and the full-size audit of all 35,554 h2 estimates against
the oracle's own Brent search.  Reference: src/bulkscan_helpers.jl:127-150, src/lmm.jl:15-33,56-86."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import assert_lod_close, make_data
from oracle import bulklmm_oracle as O

pytestmark = pytest.mark.gpu


def oracle_given_h2(Y0, X0, lam, h2, c=1):
    """univar_liteqtl's scan part (src/bulkscan_helpers.jl:138-146) column by column at the given h2."""
    return np.hstack([O.univar_liteqtl(Y0[:, j], X0[:, :c], X0[:, c:], lam, h2_override=float(h2[j]))[0] for j in range(Y0.shape[1])])


def rotated_problem(n, p, m, lam, seed, c=1):
    rng = np.random.default_rng(seed)
    Y0 = rng.standard_normal((n, m)) * np.sqrt(np.abs(lam)[:, None] * rng.uniform(0, 2, m)[None, :] + 1.0)
    X0 = rng.standard_normal((n, c + p))
    return Y0, X0


SPECTRA = {
    "bxd_like": lambda n, rng: np.sort(np.exp(rng.uniform(np.log(0.02), np.log(40.0), n))),
    "two_tight_clusters": lambda n, rng: np.sort(np.concatenate([0.5 + 1e-9 * rng.standard_normal(n // 2),
                                                                 7.0 + 1e-9 * rng.standard_normal(n - n // 2)])),
    "rank_deficient": lambda n, rng: np.sort(np.concatenate([np.zeros(n // 3), rng.uniform(0.1, 30.0, n - n // 3)])),
    "tiny_negative_inside_positive_branch": lambda n, rng: np.sort(np.concatenate([[-3e-14, -1e-15, 2e-16],
                                                                                    rng.uniform(0.05, 20.0, n - 3)])),
    "negative_eigenvalue": lambda n, rng: np.sort(np.concatenate([[-1e-8], rng.uniform(0.05, 20.0, n - 1)])),
    "one_huge": lambda n, rng: np.sort(np.concatenate([rng.uniform(0.01, 1.0, n - 1), [1e6]])),
}


@pytest.mark.parametrize("spectrum", sorted(SPECTRA))
def test_lowrank_guard_on_adversarial_spectra_and_h2(blmm, spectrum):
    """liteqtl_given_h2 takes the same kernel choice as bulkscan(null-exact): heritabilities at both ends of [0, 1)
    (1e-15, 1 - 1e-12: delta = 1e12), mid-range ones, and spectra the basis was not tuned on."""
    n, p, m = 90, 140, 24
    rng = np.random.default_rng(100 + sorted(SPECTRA).index(spectrum))
    lam = SPECTRA[spectrum](n, rng)
    Y0, X0 = rotated_problem(n, p, m, lam, 7)
    h2 = np.concatenate([[0.0, 1e-15, 1e-9, 1.0 - 1e-12, 1.0 - 1e-9, 1.0 - 1e-6, 0.999], rng.uniform(0, 1, m - 7)])
    got = blmm.liteqtl_given_h2(Y0, X0, lam, h2)
    ref = oracle_given_h2(Y0, X0, lam, h2)
    assert np.isfinite(got).all()
    # delta = 1e12 next to exact-zero / negative eigenvalues makes the statistic itself ill-conditioned (weights span 12
    # decades and the projected marker norm cancels): both sides then agree to 1e-5, everything else to the usual bar
    hard = (h2 > 1 - 1e-8) & (lam.min() <= 1e-12)
    assert_lod_close(got[:, ~hard], ref[:, ~hard])
    if hard.any():
        assert_lod_close(got[:, hard], ref[:, hard], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("ncov", [0, 2])
def test_rescan_kernel_equals_oracle_when_every_trait_is_flagged(blmm, ncov, monkeypatch):
    Y, G, K, Cov = make_data(p=333, m=77, seed=4100 + ncov, ncov=ncov)
    base = blmm.bulkscan_null(Y, G, K, Cov)
    dctx = blmm.default_context()
    dctx.set_tuning("lr_tol", 0.0)      # (was BLMM_LR_TOL=0: numerics-changing switches are context tuning since 0.2.3)
    try:
        L, h2, st = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, Cov, None, True, None, 1.0, 0.0, False, 1,
                                            "eigen", 0, None, return_status=True)
    finally:
        dctx.set_tuning("defaults", 0)
    assert st.lowrank_fallback == 77 and np.array_equal(h2, base.h2_null_list)
    pin = O.bulkscan_null(Y, G, K, Covar=Cov, h2_override=h2)
    assert_lod_close(L, pin.L)
    assert_lod_close(L, base.L, rtol=1e-9, atol=1e-12)      # the MFMA low-rank form and the plain full-length sums
    L2, _, st2 = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, Cov, None, True, None, 1.0, 0.0, False, 1,
                                         "eigen", 0, None, return_status=True)
    assert st2.lowrank_fallback == 0 and st2.lowrank_resid <= 1e-13 and np.array_equal(L2, base.L)


def test_rank_deficient_kinship_end_to_end(blmm):
    """K = G G'/p from p < n markers: n - p exact-zero eigenvalues (+- rounding), traits with h2 at the upper bound."""
    rng = np.random.default_rng(77)
    n, p0 = 100, 60
    A = rng.standard_normal((n, p0))
    K = A @ A.T / p0
    G = rng.random((n, 200))
    Y = A @ rng.standard_normal((p0, 30)) * 3.0 + 0.05 * rng.standard_normal((n, 30)) + 5.0   # nearly all variance genetic
    Y[:, 20:] = rng.standard_normal((n, 10))                                                    # and some pure noise
    L, h2, st = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1,
                                        "eigen", 0, None, return_status=True)
    ref = O.bulkscan_null(Y, G, K)
    assert np.abs(h2 - ref.h2_null_list).max() <= 1e-6
    pin = O.bulkscan_null(Y, G, K, h2_override=h2)
    assert_lod_close(L, pin.L, rtol=1e-6, atol=1e-9)
    assert st.lowrank_resid <= 1e-13 or st.lowrank_fallback > 0


def test_fullsize_h2_audit_all_traits(blmm, tmp_path):
    """Every one of the 35,554 BXD-shaped traits: GPU h2 (k_brent / k_brent2) against the oracle's fitlmm.  Brent is a
    LOCAL method (src/gridbrent.jl:9-24 runs it once over [0, 1] with optim_interval = 1): on a profile likelihood with
    two local maxima, rounding-level differences in the first evaluations decide which one a run ends in, for the
    reference's own arithmetic as much as for ours.  So a trait whose two h2 differ by more than 1e-6 must be either a
    tie (same log-likelihood to 1e-9 relative) or sit on a genuine local maximum of the ORACLE's likelihood function --
    anything else is a divergence and fails -- and such traits must be rare (measured: 2 of 35,554)."""
    N, P, M = 79, 64, 35554
    Y, G, K, _ = make_data(n=N, p=P, m=M, seed=20241)
    got = blmm.bulkscan_null(Y, G, K)
    Y0, X0, lam = O.transform_rotation(Y, G, K)
    Z0 = X0[:, :1]
    prior = [1.0, 0.0]
    nproc = min(16, len(os.sched_getaffinity(0)))
    # the CPU pool runs as its own program: no worker is forked from this process, which holds the GPU
    np.savez(tmp_path / "in.npz", Y0=Y0, Z0=Z0, lam=lam, prior=np.array(prior))
    here = os.path.dirname(os.path.abspath(__file__))
    run = subprocess.run([sys.executable, os.path.join(here, "helpers", "fitlmm_pool.py"), str(tmp_path / "in.npz"),
                          str(tmp_path / "out.npy"), str(nproc)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    res = np.load(tmp_path / "out.npy")
    h2_o, ell_o = res[:, 0], res[:, 1]
    dh = np.abs(got.h2_null_list - h2_o)
    bad = np.flatnonzero(dh > 1e-6)
    ties = other_optimum = 0
    for j in bad:
        ell = lambda h: O.wls(Y0[:, j], Z0, O.makeweights(h, lam), prior).ell   # noqa: E731
        hg = float(got.h2_null_list[j])
        eg = ell(hg)
        if abs(eg - ell_o[j]) <= 1e-9 * abs(ell_o[j]):
            ties += 1
            continue
        # a different optimum: it has to be a local maximum of the oracle's own likelihood
        slack = 1e-9 * abs(eg)
        assert eg >= ell(min(hg + 1e-3, 1 - 1e-9)) - slack and eg >= ell(max(hg - 1e-3, 0.0)) - slack, (j, hg, h2_o[j], eg, ell_o[j])
        other_optimum += 1
        print(f"trait {j}: GPU h2 {hg:.6g} (ell {eg:.9g}) vs oracle h2 {h2_o[j]:.6g} (ell {ell_o[j]:.9g}): two local maxima")
    print(f"h2 audit: {bad.size} of {M} traits differ by more than 1e-6 (max |dh2| {dh.max():.3e}): {ties} ties, "
          f"{other_optimum} on the other local maximum of a bimodal profile")
    assert bad.size <= 0.0005 * M


@pytest.mark.parametrize("ncov", [0, 1, 2])
def test_shared_weights_class(blmm, ncov):
    """Traits whose weights are 1 to within the guard's tolerance (h2 at the 0 boundary) take the shared-weights path of
    k_scan_lr: no rank-R phase, per-marker denominators of the unweighted model.  Class sizes that are not tile multiples,
    every covariate count of the low-rank form; the class size reported in blmm_status against the criterion
    delta_j sqrt(sum lambda^2 / n) <= 1e-13 evaluated here, and the LODs against the oracle at the device's h2."""
    Y, G, K, Cov = make_data(p=300, m=333, seed=5200 + ncov, ncov=ncov)
    rng = np.random.default_rng(9)
    noise = rng.permutation(333)[:150]
    Y[:, noise] = rng.standard_normal((Y.shape[0], 150))                 # no heritable part: most of these land on h2 = 0
    L, h2, st = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, Cov, None, True, None, 1.0, 0.0, False, 1,
                                        "eigen", 0, None, return_status=True)
    lam = np.linalg.eigvalsh(K)
    want = int((h2 / (1.0 - h2) * np.sqrt((lam ** 2).sum() / lam.size) <= 1e-13).sum())
    assert 0 < want < 333 and want % 32 != 0
    assert st.lowrank_shared == want and st.lowrank_fallback == 0
    pin = O.bulkscan_null(Y, G, K, Covar=Cov, h2_override=h2)
    assert_lod_close(L, pin.L)
    own = O.bulkscan_null(Y, G, K, Covar=Cov)
    assert np.abs(h2 - own.h2_null_list).max() <= 1e-6


@pytest.mark.parametrize("h2val,m", [(0.0, 130), (0.0, 64), (0.37, 130), (None, 1)])
def test_shared_weights_class_extremes(blmm, h2val, m):
    """All traits in the class, none, and a single trait (h2 = 0 exactly is what a grid or a caller may pass)."""
    n, p = 79, 200
    rng = np.random.default_rng(17 + m)
    lam = SPECTRA["bxd_like"](n, rng)
    Y0, X0 = rotated_problem(n, p, m, lam, 3)
    h2 = np.full(m, 0.0 if h2val is None else h2val)
    got = blmm.liteqtl_given_h2(Y0, X0, lam, h2)
    assert_lod_close(got, oracle_given_h2(Y0, X0, lam, h2))


@pytest.mark.parametrize("segments", ["default", "2", "8"])
def test_segmented_weight_basis_equals_the_single_basis(blmm, segments, monkeypatch):
    """The heritability axis is cut into segments with one weight basis each (kernels_lowrank.hip, LrSeg: rank 11-12 per segment on
    the BXD spectrum against 23 for the whole family, so the rank-R phase of k_scan_lr runs half the K steps).  Same h2 (the search
    does not see the basis), LODs equal to the single-basis form's to rounding, the guard flags nothing, and the profile of what
    was executed (blmm_lowrank_profile) shows several segments of lower rank whose traits add up."""
    Y, G, K, _ = make_data(n=79, p=300, m=2500, seed=8101, bxd=True)
    ctx = blmm.Context(0)
    ctx.set_tuning("lr_segments", 1)
    one = blmm.bulkscan_null(Y, G, K, ctx=ctx)
    shared1, prof1 = ctx.lowrank_profile()
    assert len(prof1) == 1
    ctx.set_tuning("lr_segments", 0 if segments == "default" else int(segments))
    seg = blmm.bulkscan_null(Y, G, K, ctx=ctx)
    shared, prof = ctx.lowrank_profile()
    assert np.array_equal(seg.h2_null_list, one.h2_null_list)
    assert np.abs(seg.L - one.L).max() <= 1e-11 * max(1.0, np.abs(one.L).max())
    assert shared == shared1 and shared + sum(c for c, _ in prof) == Y.shape[1]
    assert len([1 for c, _ in prof if c > 0]) >= 2 and max(r for _, r in prof) < prof1[0][1]
    check = O.bulkscan_null(Y[:, :40], G, K, h2_override=seg.h2_null_list[:40])
    assert_lod_close(seg.L[:, :40], check.L)
    ctx.close()


def test_split_h2_search_equals_the_single_region_form(blmm, monkeypatch):
    """Two panel regions (the traits the first Brent kernel finished are scanned while the second kernel finishes the others; the
    default from 8192 traits) against one region (below): the same per-trait arithmetic, so h2 and every LOD bit for bit."""
    Y, G, K, _ = make_data(n=79, p=260, m=3000, seed=8202, bxd=True)
    ctx = blmm.Context(0)
    ctx.set_tuning("lr_split", 0)
    one = blmm.bulkscan_null(Y, G, K, ctx=ctx)
    ctx.set_tuning("lr_split", 1)
    two = blmm.bulkscan_null(Y, G, K, ctx=ctx)
    ctx.set_tuning("lr_split", -1)
    auto = blmm.bulkscan_null(Y, G, K, ctx=ctx)
    assert np.array_equal(one.h2_null_list, two.h2_null_list) and np.array_equal(one.L, two.L) and np.array_equal(one.L, auto.L)
    shared, prof = ctx.lowrank_profile()
    assert shared + sum(c for c, _ in prof) == Y.shape[1]
    ctx.close()
