"""CPU tests: the oracle against every known-answer test / fixture of the reference that can be run without the
(absent) BXD genotype and phenotype CSVs, plus the reference's cross-path identities re-created on synthetic
data with the real BXD kinship.  No GPU needed."""
import os
import warnings

import numpy as np
import pytest

from common import GOLDEN, bxd_kinship, make_data, make_geno
from oracle import bulklmm_oracle as O


def test_r2lod_roundtrip_kat():
    # test/bulkscan_test.jl:9-19
    assert abs(O.r2lod(O.lod2r(3.0, 79), 79) - 3.0) <= 1e-7


def test_computeR_LMM_equals_pearson():
    # test/bulkscan_test.jl:25-54
    rng = np.random.default_rng(1234)
    X = rng.standard_normal((100, 100))
    Y = rng.standard_normal((100, 100))
    R = O.computeR_LMM(Y, X, np.ones((100, 1)))
    C = np.corrcoef(X.T, Y.T)[:100, 100:]
    assert np.sum((R - C) ** 2) <= 1e-8
    x = rng.standard_normal((100, 1))
    y = 3 * x + rng.standard_normal()
    assert abs(O.computeR_LMM(y, x, np.ones((100, 1)))[0, 0] - np.corrcoef(x[:, 0], y[:, 0])[0, 1]) <= 1e-7


def test_resid_rss_vs_backslash():
    # test/wls_basic_test.jl:30-74
    rng = np.random.default_rng(7)
    X = rng.standard_normal((60, 4))
    Y = rng.standard_normal((60, 3))
    b = np.linalg.lstsq(X, Y, rcond=None)[0]
    assert np.abs(O.resid(Y, X) - (Y - X @ b)).max() <= 1e-10
    assert np.abs(O.rss(Y, X) - np.sum((Y - X @ b) ** 2, axis=0)).max() <= 1e-8
    assert np.abs(O.resid(Y, X, method="cholesky") - O.resid(Y, X, method="qr")).max() <= 1e-8


def test_wls_vs_scaled_ols():
    # test/wls_results_test.jl:89-117
    rng = np.random.default_rng(11)
    n = 200
    X = np.hstack([np.ones((n, 1)), rng.standard_normal((n, 2))])
    beta = np.array([[1.0], [2.0], [-1.0]])
    w = rng.uniform(0.2, 3.0, n)
    y = X @ beta + rng.standard_normal((n, 1)) / np.sqrt(w)[:, None]
    est = O.wls(y, X, w, [0.0, 0.0])
    sw = np.sqrt(w)[:, None]
    b = np.linalg.lstsq(X * sw, y * sw, rcond=None)[0]
    assert np.abs(est.b - b).max() <= 1e-4
    assert np.abs(est.b - beta).max() <= 0.3
    mv = O.wls_multivar(np.hstack([y, 2 * y]), X, w, [1.0, 0.1], reml=True)
    one = O.wls(2 * y, X, w, [1.0, 0.1], reml=True)
    assert abs(mv.Ell[0, 1] - one.ell) <= 1e-9 and abs(mv.Sigma2[0, 1] - one.sigma2) <= 1e-12


def test_gridbrent_kat():
    # test/gridbrent_test.jl:1-12
    f = lambda x: -(x ** 3 + 0.2 * (x - 2) ** 2 + 3)  # noqa: E731
    assert abs(O.gridbrent(f, -3.0, 1.0, 100).minimizer - 1.0) <= 1e-6
    # a plain quadratic: Brent must land on the minimiser to sqrt(eps)
    r = O.brent_optim(lambda x: (x - 0.3) ** 2, 0.0, 1.0)
    assert r.converged and abs(r.minimizer - 0.3) <= 1e-7


def test_makeweights_error_string():
    # test/lmm_test.jl:12-18
    with pytest.raises(O.BulkLMMError) as e:
        O.makeweights(1.0, [0.0])
    assert e.value.msg == "Heritability of 1 is not allowed."
    assert np.allclose(O.makeweights(0.5, [0.0, 1.0, 3.0]), [1.0, 0.5, 0.25])


def test_bxd_kinship_fixture():
    # test/kinship_test.jl:5-7 -- the stored Helium matrix (79 x 79), rounded to 12 digits
    K = bxd_kinship()
    assert K.shape == (79, 79) and np.array_equal(K, K.T) and np.array_equal(np.diag(K), np.ones(79))
    lam = np.linalg.eigvalsh(K)
    assert abs(lam[0] - 0.02146) <= 1e-4 and abs(lam[-1] - 40.62) <= 1e-2
    assert np.array_equal(O.read_he(os.path.join(GOLDEN, "bxd_kinship_ref.he")).round(12), K)


def test_calc_kinship_properties():
    rng = np.random.default_rng(5)
    G = make_geno(30, 400, rng)
    K = O.calcKinship(G)
    assert np.array_equal(np.diag(K), np.ones(30)) and np.allclose(K, K.T)
    assert abs(K[0, 1] - (2 * np.dot(G[0] - .5, G[1] - .5) / 400 + .5)) <= 1e-14


def test_transform_rotation_checks():
    # test/transform_helpers_test.jl:12-53
    Y, G, K, _ = make_data(p=20, m=3)
    with pytest.raises(O.BulkLMMError) as e:
        O.transform_rotation(Y[:-1], G, K)
    assert e.value.msg == "Dimension mismatch."
    with pytest.raises(O.BulkLMMError) as e:
        O.transform_rotation(Y, G, K, decomp_scheme="lu")
    assert e.value.msg == "Please choose either `eigen` or `svd` for decomposition of the kinship matrix."
    Y0, X0, lam = O.transform_rotation(Y, G, K)
    vals, vecs = np.linalg.eigh(K)
    assert np.allclose(Y0, vecs.T @ Y) and np.allclose(X0[:, 1:], vecs.T @ G) and np.allclose(lam, vals)


def test_bulkscan_null_equals_scan_null():
    # test/bulkscan_test.jl:60-80 incl. the prior's scale equivariance
    Y, G, K, _ = make_data(p=120, m=4, seed=31)
    got = O.bulkscan_null(O.colStandardize(Y), O.colStandardize(G), K, prior_variance=1.0, prior_sample_size=0.1)
    for j in (0, 3):
        y = Y[:, [j]]
        s = O.scan(y, G, K, prior_variance=float(np.var(y, ddof=1)), prior_sample_size=0.1)
        assert np.sum((s["lod"] - got.L[:, j]) ** 2) <= 1e-7


def test_null_grid_with_exact_h2():
    # test/bulkscan_test.jl:86-107
    Y, G, K, _ = make_data(p=80, m=3, seed=52)
    ex = O.bulkscan_null(Y, G, K)
    grid = list(np.arange(0.0, 1.0, 0.05)) + list(ex.h2_null_list)
    gr = O.bulkscan_null_grid(Y, G, K, grid)
    same = gr.h2_null_list == ex.h2_null_list
    assert same.any()
    assert np.abs(gr.L[:, same] - ex.L[:, same]).max() <= 1e-12


def test_weights_equal_prescaled_and_svd_equal_eigen():
    # test/weighted_error_test.jl:42-127, test/scan_covar_test.jl:27-40
    Y, G, K, _ = make_data(p=60, m=3, seed=81)
    n = Y.shape[0]
    w = np.random.default_rng(1).uniform(0.5, 2.0, n)
    W = np.diag(w)
    a = O.bulkscan_null(Y, G, K, weights=w)
    b = O.bulkscan_null(W @ Y, W @ G, W @ K @ W, Covar=W @ np.ones((n, 1)), addIntercept=False)
    assert np.abs(a.L - b.L).max() <= 1e-3
    e = O.bulkscan_null(Y, G, K, decomp_scheme="eigen")
    s = O.bulkscan_null(Y, G, K, decomp_scheme="svd")
    assert np.abs(e.L - s.L).mean() <= 1e-8


def test_alt_grid_dominates_null_grid_and_dispatcher():
    # alt-grid maximises over the same grid per marker, so it can only raise the log-likelihood ratio
    Y, G, K, _ = make_data(p=50, m=5, seed=61)
    grid = [i / 10.0 for i in range(10)]
    al = O.bulkscan_alt_grid(Y, G, K, grid)
    gr = O.bulkscan_null_grid(Y, G, K, grid)
    assert (al.L >= gr.L - 1e-9).all()
    assert set(np.unique(al.h2_panel)).issubset(set(grid))
    d = O.bulkscan(Y, G, K, method="null-grid")
    assert np.array_equal(d["L"], O.bulkscan_null_grid(Y, G, K, grid).L)
    with pytest.raises(O.BulkLMMError):
        O.bulkscan(Y, G, K, method="bogus")


def test_scan_errors_and_perms():
    Y, G, K, _ = make_data(p=40, m=2, seed=9)
    with pytest.raises(O.BulkLMMError) as e:
        O.scan(Y, G, K, permutation_test=True)
    assert e.value.msg == "Can only handle one trait."
    with pytest.raises(O.BulkLMMError) as e:
        O.scan(Y[:, 0], G, K, addIntercept=False)
    assert e.value.msg == "Intercept has to be added when no other covariate is given."
    with pytest.raises(O.BulkLMMError) as e:
        O.scan(Y[:, 0], G, K, assumption="x")
    assert e.value.msg == "Assumption keyword is not supported. Please enter null or alt."
    r = O.scan(Y[:, 0], G, K, permutation_test=True, nperms=5)
    s = O.scan(Y[:, 0], G, K)
    assert r["L_perms"].shape == (40, 5) and np.sum((r["lod"] - s["lod"]) ** 2) <= 1e-7


def test_zero_column_raises():
    # src/util.jl:69-71 via computeR_LMM's colDivide!
    Y, G, K, _ = make_data(p=20, m=2, seed=3)
    G = G.copy()
    G[:, 4] = 0.0
    with pytest.raises(O.BulkLMMError) as e:
        O.bulkscan_null_grid(Y, G, K, [0.0, 0.5])
    assert e.value.msg == "Dividing by zeros: the input vector can not contain any zeros!"


def test_golden_fixture_reproduces():
    # tests/golden/bulkscan_small.npz (tests/golden/make_golden.py)
    z = np.load(os.path.join(GOLDEN, "bulkscan_small.npz"))
    ex = O.bulkscan_null(z["Y"][:, :6], z["G"], z["K"], prior_variance=1.0, prior_sample_size=0.1)
    assert np.abs(ex.L - z["exact_L"][:, :6]).max() <= 1e-9 and np.abs(ex.h2_null_list - z["exact_h2"][:6]).max() <= 1e-9
    gr = O.bulkscan_null_grid(z["Y"], z["G"], z["K"], list(z["grid"]))
    assert np.array_equal(gr.h2_null_list, z["grid_h2"]) and np.abs(gr.L - z["grid_L"]).max() <= 1e-10


def dense_gls_lod(Y, G, K, h2, Covar=None):
    """An oracle that shares NOTHING with the rotation / weights algebra of the reference or of the restatement above:
    the model of README.md:18-33 written out densely.  V = h2 K + (1 - h2) I (the error covariance up to sigma^2),
    V = C C' (Cholesky), whiten y and [Z g_i] by C^-1, ordinary least squares with and without the marker, and the
    likelihood-ratio statistic LOD = (n/2) log10(rss0 / rss1)."""
    n, m = Y.shape
    Z = np.ones((n, 1)) if Covar is None else np.hstack([np.ones((n, 1)), Covar])
    L = np.empty((G.shape[1], m))
    for j in range(m):
        V = h2[j] * K + (1.0 - h2[j]) * np.eye(n)
        C = np.linalg.cholesky(V)
        yt = np.linalg.solve(C, Y[:, j])
        Zt = np.linalg.solve(C, Z)
        Gt = np.linalg.solve(C, G)
        Q, _ = np.linalg.qr(Zt)
        r0 = yt - Q @ (Q.T @ yt)
        rss0 = r0 @ r0
        Gr = Gt - Q @ (Q.T @ Gt)                      # markers with the null covariates projected out
        num = (Gr.T @ r0) ** 2 / np.sum(Gr * Gr, axis=0)
        L[:, j] = 0.5 * n * np.log10(rss0 / (rss0 - num))
    return L


@pytest.mark.parametrize("ncov", [0, 2, 7])   # 7: the oracle that pins the c = 5..8 kernels is itself pinned to the model
def test_oracle_equals_dense_gls_model(ncov):
    """Pins the restatement to the MODEL (y = X b + e, V(e) = s2g K + s2e I, README.md:18-33) independently of the
    eigen-rotation and LiteQTL weights algebra that bulkscan_null and scan_null share: same h2 in, LODs equal to 1e-8."""
    Y, G, K, Cov = make_data(p=60, m=7, seed=606 + ncov, ncov=ncov)
    h2 = np.array([0.0, 0.05, 0.3, 0.5, 0.77, 0.9, 0.999])
    ref = O.bulkscan_null(Y, G, K, Covar=Cov, h2_override=h2)
    gls = dense_gls_lod(Y, G, K, h2, Cov)
    assert np.abs(ref.L - gls).max() <= 1e-8 * max(1.0, np.abs(gls).max())
    # and the null log-likelihood the h2 search maximises: ell(h2) of wls on rotated data == the dense Gaussian
    # log-likelihood with sigma^2 profiled out (same h2 ranking => same optimiser target)
    y0, X0, lam = O.transform_rotation(Y[:, :1], G, K) if Cov is None else O.transform_rotation(Y[:, :1], np.hstack([Cov, G]), K)
    c = 1 + ncov
    n = Y.shape[0]
    Z = np.ones((n, 1)) if Cov is None else np.hstack([np.ones((n, 1)), Cov])
    ells, dense = [], []
    for h in (0.1, 0.4, 0.8):
        ells.append(O.wls(y0, X0[:, :c], O.makeweights(h, lam), [0.0, 0.0]).ell)
        # V(e)/s2e = delta K + I with delta = h/(1-h): the parametrisation makeweights uses (src/lmm.jl:15-33)
        V = (h / (1 - h)) * K + np.eye(n)
        C = np.linalg.cholesky(V)
        yt, Zt = np.linalg.solve(C, Y[:, 0]), np.linalg.solve(C, Z)
        b = np.linalg.lstsq(Zt, yt, rcond=None)[0]
        rss = np.sum((yt - Zt @ b) ** 2)
        s2 = rss / n
        dense.append(-0.5 * (n * np.log(s2) + 2 * np.sum(np.log(np.diag(C))) + rss / s2))
    assert np.allclose(ells, dense, rtol=1e-10, atol=1e-9)


BXD_PHENO = "/root/reference/data/bxdData/spleen-pheno-nomissing.csv"
BXD_GENO = "/root/reference/data/bxdData/spleen-bxd-genoprob.csv"


@pytest.mark.skipif(not (os.path.exists(BXD_PHENO) and os.path.exists(BXD_GENO)),
                    reason="the BXD spleen CSVs are listed in the reference's .MISSING_LARGE_BLOBS; this KAT activates when they appear")
def test_bxd_readme_and_lmmlite_kats():
    """End-to-end known answers of the reference, usable the moment its two data files exist: README.md:215
    (sigma2_e, h2) = (0.0942525841453798, 0.850587848871709) for trait 1112 and the R/lmmlite LODs of trait 7919
    (test/run-lmmlite_R/output/result.lmmlite_{ML,REML}.csv; tolerance of test/scan_test_lmmlite.jl:27-32).
    Loader as test/generate_test_bxdData.jl:4-14."""
    import csv
    ph = np.genfromtxt(BXD_PHENO, delimiter=",", skip_header=1)[:, 1:-1]
    ge = np.genfromtxt(BXD_GENO, delimiter=",", skip_header=1)[:, 0::2]
    K = np.round(O.calcKinship(ge), 12)
    r = O.scan(ph[:, 1111], ge, K)
    assert abs(r["sigma2_e"] - 0.0942525841453798) <= 1e-8 and abs(r["h2_null"] - 0.850587848871709) <= 1e-6
    for reml, name in ((False, "ML"), (True, "REML")):
        rows = list(csv.reader(open(f"/root/reference/test/run-lmmlite_R/output/result.lmmlite_{name}.csv")))[2:]
        lods = np.array([float(x[4]) for x in rows])
        got = O.scan(ph[:, 7918], ge, K, reml=reml)["lod"]
        assert np.max((got - lods) ** 2) <= 1e-9 and np.sum((got - lods) ** 2) <= np.sqrt(1e-9)


@pytest.mark.skipif(__import__("shutil").which("gcc") is None, reason="no gcc")
@pytest.mark.parametrize("ncov,reml,prior,oi", [(0, False, (1.0, 0.0), 1), (2, True, (1.0, 0.1), 1), (1, False, (0.0, 0.0), 3)])
def test_c_openmp_restatement_equals_numpy_restatement(ncov, reml, prior, oi):
    """oracle/bulkscan_null_ref.c (the CPU baseline bench.py times) against oracle/bulklmm_oracle.py: two independently
    written restatements of src/bulkscan.jl:212-314 (own Jacobi eigensolver and Householder QR there, LAPACK here)."""
    from oracle import cref
    Y, G, K, Cov = make_data(p=90, m=12, seed=909 + ncov, ncov=ncov)
    L, h2 = cref.bulkscan_null(Y, G, K, Cov, prior_variance=prior[0], prior_sample_size=prior[1], reml=reml, optim_interval=oi, nthreads=2)
    ref = O.bulkscan_null(Y, G, K, Covar=Cov, prior_variance=prior[0], prior_sample_size=prior[1], reml=reml, optim_interval=oi)
    assert np.abs(h2 - ref.h2_null_list).max() <= 1e-6
    pin = O.bulkscan_null(Y, G, K, Covar=Cov, prior_variance=prior[0], prior_sample_size=prior[1], reml=reml, h2_override=h2)
    assert np.abs(L - pin.L).max() <= 1e-9 * max(1.0, np.abs(pin.L).max())
    assert np.sum((L - ref.L) ** 2, axis=0).max() <= 1e-7
    # the override used by the every-entry GPU comparisons (tests/test_gpu_fullmatrix.py): L at heritabilities handed in -- here
    # values that are NOT the search's result -- equals the NumPy restatement at the same values; the search still runs beside it,
    # or is skipped
    rng = np.random.default_rng(5 + ncov)
    hov = np.clip(h2 + rng.uniform(-0.2, 0.2, h2.shape), 0.0, 0.95)
    hov[0] = 0.0
    hov[1] = 1e-15
    L2, h2own = cref.bulkscan_null(Y, G, K, Cov, prior_variance=prior[0], prior_sample_size=prior[1], reml=reml, optim_interval=oi,
                                   nthreads=2, h2_override=hov)
    assert np.array_equal(h2own, h2)
    pin2 = O.bulkscan_null(Y, G, K, Covar=Cov, prior_variance=prior[0], prior_sample_size=prior[1], reml=reml, h2_override=hov)
    assert np.abs(L2 - pin2.L).max() <= 1e-9 * max(1.0, np.abs(pin2.L).max())
    L3, h3 = cref.bulkscan_null(Y, G, K, Cov, prior_variance=prior[0], prior_sample_size=prior[1], reml=reml, nthreads=2,
                                h2_override=hov, skip_search=True)
    assert np.array_equal(L3, L2) and np.array_equal(h3, hov)


def test_oracle_scan_alt_is_the_profile_likelihood_ratio_and_restates_the_closing_calls():
    """scan_alt (src/scan.jl:397-453) in the oracle: with `true_weights` the LOD is the ratio of the two maximised
    log-likelihoods (fitlmm's own `ell`), per marker; as written in the reference -- the closing `wls` calls get
    sqrt.(makeweights) as their weights -- it is the same expression with the square-rooted weights.  Both against direct
    evaluation, so that the device's parity target is pinned to something independent of the function under test."""
    from common import make_data
    Y, G, K, _ = make_data(n=40, p=12, m=6, seed=77, bxd=False)
    j = int(np.argmin(np.abs(O.bulkscan_null(Y, G, K, prior_variance=1.0, prior_sample_size=0.1).h2_null_list - 0.5)))
    y = Y[:, [j]]
    prior = [1.0, 0.1]
    y0, X0, lam = O.transform_rotation(y, G, K, addIntercept=True)
    null = O.fitlmm(y0, X0[:, :1], lam, prior)
    a_true = O.scan(y, G, K, assumption="alt", prior_variance=1.0, prior_sample_size=0.1, true_weights=True)
    a_ref = O.scan(y, G, K, assumption="alt", prior_variance=1.0, prior_sample_size=0.1)
    assert abs(a_true["h2_null"] - null.h2) < 1e-12
    for i in range(G.shape[1]):
        Xi = X0[:, [0, 1 + i]]
        alt = O.fitlmm(y0, Xi, lam, prior)
        assert abs(a_true["h2_each_marker"][i] - alt.h2) < 1e-12
        assert abs(a_true["lod"][i] - (alt.ell - null.ell) / np.log(10)) < 1e-10
        q1 = O.wls(y0, Xi, np.sqrt(O.makeweights(alt.h2, lam)), prior).ell
        q0 = O.wls(y0, X0[:, :1], np.sqrt(O.makeweights(null.h2, lam)), prior).ell
        assert abs(a_ref["lod"][i] - (q1 - q0) / np.log(10)) < 1e-10
    assert np.all(a_true["lod"] >= -1e-9)          # a likelihood ratio of nested models
