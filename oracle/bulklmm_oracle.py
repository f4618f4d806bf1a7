"""
CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the bulkscan hot path of senresearch/BulkLMM.jl v1.2.0.

This file is a NumPy restatement, operation by operation, of the reference's Julia algorithm.  It is
the *checker*: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it.  The product path (`bulklmm.jl_amd/`, `libbulklmm_hip.so`) never imports, links or calls
anything in `oracle/`.

Pinning status
--------------
* No Julia exists in the build container, so the reference itself cannot be executed
  (SURVEY.md §8(c)); the oracle is pinned by the reference's own known-answer tests and fixtures that
  do not need the (absent) BXD genotype/phenotype CSVs:
    r2lod round trip            test/bulkscan_test.jl:9-19
    computeR_LMM == Pearson cor test/bulkscan_test.jl:25-54
    resid / rss vs `\\`          test/wls_basic_test.jl:30-74
    wls vs scaled OLS           test/wls_results_test.jl:89-117
    gridbrent KAT               test/gridbrent_test.jl:1-12
    makeweights error string    test/lmm_test.jl:12-18
    kinship fixture             test/ref_data_for_tests/kinship_ref.he == processed_bxdData/BXDkinship.csv
    bulkscan_null == scan_null  test/bulkscan_test.jl:60-80 (identity re-created on synthetic data + BXD kinship)
  (see tests/test_oracle_*.py).
* The univariate optimiser is Optim.jl's `Brent()` (Project.toml:13,22 compat "1.7, 2"; third-party,
  source not under /root/reference).  `brent_optim` below restates its published algorithm; the only
  reference test pinning it is test/gridbrent_test.jl.  Per-trait h2 values are therefore
  "PARITY UNPINNED" beyond that KAT (LOD sensitivity to a 1e-8 change of h2 is far below the 1e-6
  parity tolerance).
* Julia's `MersenneTwister`/`shuffle` stream cannot be reproduced; permutations are supplied by the
  caller (`perm_idx`) or drawn from NumPy's PCG64.  Exact permutation parity is unpinned.

All matrices are float64.  Citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
import warnings
from typing import NamedTuple, Optional, Sequence

import numpy as np

__all__ = [
    "rowMultiply", "colDivide", "colStandardize", "r2lod", "lod2r", "makeweights", "wls", "wls_multivar",
    "resid", "rss", "brent_optim", "gridbrent", "fitlmm", "calcKinship", "transform_rotation",
    "transform_reweight", "transform_permute", "computeR_LMM", "univar_liteqtl", "weighted_liteqtl",
    "find_optim_h2", "gridscan_by_bin", "bulkscan", "bulkscan_null", "bulkscan_null_grid",
    "bulkscan_alt_grid", "scan", "scan_null", "scan_alt", "scan_perms_lite", "lod2log10p", "read_he",
]


class BulkLMMError(Exception):
    """Stands in for Julia's ErrorException; `.msg` carries the reference's message string."""

    def __init__(self, msg: str):
        super().__init__(msg)
        self.msg = msg


def _f64(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float64)


def _mat(a) -> np.ndarray:
    a = _f64(a)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


# ----------------------------------------------------------------------------------------------
# src/util.jl
# ----------------------------------------------------------------------------------------------

def rowMultiply(A, x):
    """src/util.jl:139-156 -- B[j,i] = A[j,i]*x[j]."""
    A = _mat(A)
    x = _f64(x).ravel()
    if x.shape[0] != A.shape[0]:
        raise BulkLMMError("Matrix and vector size do not match.")
    return A * x[:, None]


def checkZeros(x) -> bool:
    """src/util.jl:47-56 -- isapprox(i, 0; atol=eps, rtol=0)."""
    return bool(np.any(np.abs(_f64(x)) <= np.finfo(np.float64).eps))


def colDivide(A, x):
    """src/util.jl:58-78 (out-of-place form of colDivide!)."""
    A = _mat(A)
    x = _f64(x).ravel()
    if x.shape[0] != A.shape[1]:
        raise BulkLMMError("Matrix and vector size do not match.")
    if checkZeros(x):
        raise BulkLMMError("Dividing by zeros: the input vector can not contain any zeros!")
    return A / x[None, :]


def colStandardize(A):
    """src/util.jl:88-96 -- centre, divide by the sample std (ddof=1)."""
    A = _mat(A)
    sA = A - A.mean(axis=0, keepdims=True)
    s = sA.std(axis=0, ddof=1)
    return colDivide(sA, s)


def lod2log10p(lod, df: int = 1):
    """src/util.jl:199-206 -- -log10 of the chi-square(df) survival function at 2*ln(10)*LOD."""
    from scipy.stats import chi2

    lrs = _f64(lod) * 2.0 * math.log(10.0)
    return -chi2.logsf(lrs, df) / math.log(10.0)


def read_he(path: str) -> np.ndarray:
    """Helium (.he) matrix as used by test/kinship_test.jl:5 -- 56-byte header
    (int64 nrow, int64 ncol, ...), then column-major float64."""
    raw = open(path, "rb").read()
    nrow, ncol = np.frombuffer(raw[:16], dtype="<i8")
    body = np.frombuffer(raw[56:56 + 8 * nrow * ncol], dtype="<f8")
    return body.reshape((ncol, nrow)).T.copy()


# ----------------------------------------------------------------------------------------------
# src/bulkscan_helpers.jl:22-24 and its test helper
# ----------------------------------------------------------------------------------------------

def r2lod(r, n: int):
    """src/bulkscan_helpers.jl:22-24 -- -(n/2)*log10(1.0 - r^2)."""
    r = _f64(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        return -(n / 2.0) * np.log10(1.0 - r * r)


def lod2r(lod: float, n: int) -> float:
    """test/bulkscan_test.jl:9-13."""
    return math.sqrt(1.0 - 10.0 ** (-2.0 / n * lod))


# ----------------------------------------------------------------------------------------------
# src/wls.jl
# ----------------------------------------------------------------------------------------------

class LSEstimates(NamedTuple):
    b: np.ndarray
    sigma2: float
    ell: float


class LSEstimatesMultivar(NamedTuple):
    B: np.ndarray
    Sigma2: np.ndarray
    Ell: np.ndarray


def _qr_solve(X, Y):
    """Julia `qr(X)\\Y` (Householder least squares) and 2*logabsdet(R)."""
    Q, R = np.linalg.qr(X, mode="reduced")
    coef = np.linalg.solve(R, Q.T @ Y) if R.shape[0] == R.shape[1] else np.linalg.lstsq(X, Y, rcond=None)[0]
    logdet = 2.0 * float(np.sum(np.log(np.abs(np.diag(R)))))
    return coef, logdet


def wls(y, X, w, prior, reml: bool = False, loglik: bool = True, method: str = "qr") -> LSEstimates:
    """src/wls.jl:27-97."""
    y = _mat(y)
    X = _mat(X)
    w = _f64(w).ravel()
    (n, p) = X.shape
    n = y.shape[0]
    if np.any(w <= 0.0):
        warnings.warn("Some weights are not positive.")
    with np.errstate(invalid="ignore"):
        sqrtw = np.sqrt(w)
    yy = rowMultiply(y, sqrtw)
    XX = rowMultiply(X, sqrtw)
    if method == "cholesky":
        G = XX.T @ XX
        coef = np.linalg.solve(G, XX.T @ yy)
        logdetXXtXX = float(np.linalg.slogdet(G)[1])
    else:
        coef, logdetXXtXX = _qr_solve(XX, yy)
    yyhat = XX @ coef
    rss0 = float(np.linalg.norm(yy - yyhat) ** 2)
    prior_df = prior[1] + 2 if prior[1] > 0.0 else prior[1]
    if reml:
        sigma2_e = (rss0 + prior[0] * prior[1]) / ((n - p) + prior_df)
    else:
        sigma2_e = (rss0 + prior[0] * prior[1]) / (n + prior_df)
    if loglik:
        with np.errstate(divide="ignore", invalid="ignore"):
            ll = -0.5 * ((n + prior[1]) * np.log(sigma2_e) - np.sum(np.log(w)) + (rss0 + prior[0] * prior[1]) / sigma2_e)
            if reml:
                ll = ll + 0.5 * (p * np.log(sigma2_e) - logdetXXtXX)
        ll = float(ll)
    else:
        ll = float("nan")
    return LSEstimates(coef, float(sigma2_e), ll)


def wls_multivar(Y, X, w, prior, reml: bool = False, loglik: bool = True, method: str = "qr") -> LSEstimatesMultivar:
    """src/wls.jl:103-176 -- column-wise `wls`; Sigma2 and Ell are 1 x m."""
    Y = _mat(Y)
    X = _mat(X)
    w = _f64(w).ravel()
    (n, p) = X.shape
    n = Y.shape[0]
    if np.any(w <= 0.0):
        warnings.warn("Some weights are not positive.")
    with np.errstate(invalid="ignore"):
        sqrtw = np.sqrt(w)
    YY = rowMultiply(Y, sqrtw)
    XX = rowMultiply(X, sqrtw)
    if method == "cholesky":
        G = XX.T @ XX
        coef = np.linalg.solve(G, XX.T @ YY)
        logdetXXtXX = float(np.linalg.slogdet(G)[1])
    else:
        coef, logdetXXtXX = _qr_solve(XX, YY)
    YYhat = XX @ coef
    rss0 = (np.linalg.norm(YY - YYhat, axis=0) ** 2).reshape(1, -1)
    prior_df = prior[1] + 2 if prior[1] > 0.0 else prior[1]
    if reml:
        sigma2_e = (rss0 + prior[0] * prior[1]) / ((n - p) + prior_df)
    else:
        sigma2_e = (rss0 + prior[0] * prior[1]) / (n + prior_df)
    if loglik:
        with np.errstate(divide="ignore", invalid="ignore"):
            ll = -0.5 * ((n + prior[1]) * np.log(sigma2_e) - np.sum(np.log(w)) + (rss0 + prior[0] * prior[1]) / sigma2_e)
            if reml:
                ll = ll + 0.5 * (p * np.log(sigma2_e) - logdetXXtXX)
    else:
        ll = np.full_like(rss0, np.nan)
    return LSEstimatesMultivar(coef, sigma2_e, ll)


def resid(y, X, method: str = "qr"):
    """src/wls.jl:221-263."""
    y = _mat(y)
    X = _mat(X)
    if method == "cholesky":
        b = np.linalg.solve(X.T @ X, X.T @ y)
    else:
        b, _ = _qr_solve(X, y)
    return y - X @ b


def rss(y, X, method: str = "qr"):
    """src/wls.jl:191-207 -- column RSS as a 1 x k row."""
    r = resid(y, X, method=method)
    return np.sum(r * r, axis=0, keepdims=True)


# ----------------------------------------------------------------------------------------------
# src/lmm.jl, src/gridbrent.jl (+ Optim.jl Brent, third-party)
# ----------------------------------------------------------------------------------------------

def makeweights(h2: float, lam) -> np.ndarray:
    """src/lmm.jl:15-33."""
    lam = _f64(lam).ravel()
    h2 = float(h2)
    with np.errstate(divide="ignore"):
        delta = np.float64(h2) / np.float64(1.0 - h2)
    if np.isinf(delta):
        raise BulkLMMError("Heritability of 1 is not allowed.")
    return 1.0 / (delta * lam + 1.0)


class BrentResult(NamedTuple):
    minimizer: float
    minimum: float
    iterations: int
    f_calls: int
    converged: bool


_GOLDEN = 0.5 * (3.0 - math.sqrt(5.0))
_SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)
_EPS = float(np.finfo(np.float64).eps)


def brent_optim(f, x_lower: float, x_upper: float, rel_tol: float = _SQRT_EPS, abs_tol: float = _EPS,
                iterations: int = 1000) -> BrentResult:
    """Optim.jl `optimize(f, a, b, Brent())` restated from its published algorithm (third-party code,
    not in /root/reference; reached from src/gridbrent.jl:16).  See SURVEY.md Appendix A.3."""
    if x_lower > x_upper:
        raise BulkLMMError("x_lower must be less than x_upper")
    new_minimizer = x_lower + _GOLDEN * (x_upper - x_lower)
    new_minimum = f(new_minimizer)
    f_calls = 1
    step = 0.0
    old_step = 0.0
    old_minimizer = new_minimizer
    old_old_minimizer = new_minimizer
    old_minimum = new_minimum
    old_old_minimum = new_minimum
    iteration = 0
    converged = False
    while iteration < iterations:
        p = 0.0
        q = 0.0
        x_tol = rel_tol * abs(new_minimizer) + abs_tol
        x_midpoint = (x_upper + x_lower) / 2
        if abs(new_minimizer - x_midpoint) <= 2 * x_tol - (x_upper - x_lower) / 2:
            converged = True
            break
        iteration += 1
        if abs(old_step) > x_tol:
            r = (new_minimizer - old_minimizer) * (new_minimum - old_old_minimum)
            q = (new_minimizer - old_old_minimizer) * (new_minimum - old_minimum)
            p = (new_minimizer - old_old_minimizer) * q - (new_minimizer - old_minimizer) * r
            q = 2 * (q - r)
            if q > 0:
                p = -p
            else:
                q = -q
        if abs(p) < abs(q * old_step / 2) and p < q * (x_upper - new_minimizer) and p < q * (new_minimizer - x_lower):
            old_step = step
            step = p / q
            x_temp = new_minimizer + step
            if (x_temp - x_lower) < 2 * x_tol or (x_upper - x_temp) < 2 * x_tol:
                step = x_tol if new_minimizer < x_midpoint else -x_tol
        else:
            old_step = (x_upper - new_minimizer) if new_minimizer < x_midpoint else (x_lower - new_minimizer)
            step = _GOLDEN * old_step
        if abs(step) >= x_tol:
            new_x = new_minimizer + step
        else:
            new_x = new_minimizer + (x_tol if step > 0 else -x_tol)
        new_f = f(new_x)
        f_calls += 1
        if new_f < new_minimum:
            if new_x < new_minimizer:
                x_upper = new_minimizer
            else:
                x_lower = new_minimizer
            old_old_minimizer = old_minimizer
            old_old_minimum = old_minimum
            old_minimizer = new_minimizer
            old_minimum = new_minimum
            new_minimizer = new_x
            new_minimum = new_f
        else:
            if new_x < new_minimizer:
                x_lower = new_x
            else:
                x_upper = new_x
            if new_f <= old_minimum or old_minimizer == new_minimizer:
                old_old_minimizer = old_minimizer
                old_old_minimum = old_minimum
                old_minimizer = new_x
                old_minimum = new_f
            elif new_f <= old_old_minimum or old_old_minimizer == new_minimizer or old_old_minimizer == old_minimizer:
                old_old_minimizer = new_x
                old_old_minimum = new_f
    return BrentResult(new_minimizer, new_minimum, iteration, f_calls, converged)


class GridBrentResult(NamedTuple):
    minimum: float
    minimizer: float


def gridbrent(f, a: float, b: float, ninterval: int = 1) -> GridBrentResult:
    """src/gridbrent.jl:9-24 -- Brent on `ninterval` equal sub-intervals, first smallest minimum wins."""
    points = np.array([a + (b - a) * (i / ninterval) for i in range(ninterval + 1)])  # range(a, b, length=k+1)
    res = [brent_optim(f, float(points[i]), float(points[i + 1])) for i in range(ninterval)]
    minimumv = np.array([r.minimum for r in res])
    idx = int(np.argmin(minimumv))
    return GridBrentResult(res[idx].minimum, res[idx].minimizer)


class LMMEstimates(NamedTuple):
    b: np.ndarray
    sigma2: float
    h2: float
    ell: float


def fitlmm(y, X, lam, prior, reml: bool = False, loglik: bool = True, method: str = "qr",
           optim_interval: int = 1, h20: float = 0.5, d: float = 1.0) -> LMMEstimates:
    """src/lmm.jl:56-86."""
    y = _mat(y)
    X = _mat(X)

    def logLik0(h2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = wls(y, X, makeweights(h2, lam), prior, reml=reml, loglik=loglik, method=method)
        return -out.ell

    lb = max(h20 - d, 0.0)
    ub = min(h20 + d, 1.0)
    opt = gridbrent(logLik0, lb, ub, optim_interval)
    h2 = opt.minimizer
    est = wls(y, X, makeweights(h2, lam), prior, reml=reml, loglik=loglik, method=method)
    return LMMEstimates(est.b, est.sigma2, float(h2), est.ell)


# ----------------------------------------------------------------------------------------------
# src/kinship.jl, src/transform_helpers.jl
# ----------------------------------------------------------------------------------------------

def calcKinship(geno) -> np.ndarray:
    """src/kinship.jl:4-14."""
    geno = _mat(geno)
    X = geno - 0.5
    K = 2.0 * (X @ X.T) / X.shape[1] + 0.5
    np.fill_diagonal(K, 1.0)
    return K


def transform_rotation(y, g, K, addIntercept: bool = True, decomp_scheme: str = "eigen"):
    """src/transform_helpers.jl:1-54 -- returns (Ut*y, Ut*X, lambda)."""
    y = _mat(y)
    g = _mat(g)
    K = _mat(K)
    n = y.shape[0]
    if g.shape[0] != n or K.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.")
    X = np.hstack([np.ones((n, 1)), g]) if addIntercept else g
    if decomp_scheme == "eigen":
        vals, vecs = np.linalg.eigh(K)
        Ut = vecs.T
        if np.any(vals < -1e-7):
            warnings.warn("Negative eigenvalues exist. The kinship matrix supplied may not be SPD.")
        return Ut @ y, Ut @ X, vals
    elif decomp_scheme == "svd":
        _, S, Vt = np.linalg.svd(K)
        if np.any(S < -1e-7):
            warnings.warn("Negative eigenvalues exist. The kinship matrix supplied may not be SPD.")
        return Vt @ y, Vt @ X, S
    raise BulkLMMError("Please choose either `eigen` or `svd` for decomposition of the kinship matrix.")


def transform_reweight(y0, X0, lam, n_covars: int = 1, prior_a: float = 0.0, prior_b: float = 0.0,
                       method: str = "qr", optim_interval: int = 1, reml: bool = False, h2_override: Optional[float] = None):
    """src/transform_helpers.jl:57-92 -- (r0*sqrtw, resid(sqrtw*X0m, sqrtw*X0c), sigma2, h2).
    `h2_override`: test hook (not in the reference) -- the final `wls` of fitlmm at a supplied h2."""
    y0 = _mat(y0)
    X0 = _mat(X0)
    if h2_override is None:
        vc = fitlmm(y0, X0[:, :n_covars], lam, [prior_a, prior_b], reml=reml, method=method, optim_interval=optim_interval)
    else:
        est = wls(y0, X0[:, :n_covars], makeweights(h2_override, lam), [prior_a, prior_b], reml=reml, method=method)
        vc = LMMEstimates(est.b, est.sigma2, float(h2_override), est.ell)
    r0 = y0 - X0[:, :n_covars] @ vc.b
    sqrtw = np.sqrt(makeweights(vc.h2, lam))
    copy_r0 = rowMultiply(r0, sqrtw)
    copy_X0 = rowMultiply(X0, sqrtw)
    X00 = resid(copy_X0[:, n_covars:], copy_X0[:, :n_covars])
    return copy_r0, X00, vc.sigma2, vc.h2


def make_perm_idx(n: int, nperms: int, rndseed: int = 0) -> np.ndarray:
    """Permutation index matrix n x nperms (0-based).  Julia's MersenneTwister stream
    (src/transform_helpers.jl:98, src/util.jl:175) cannot be reproduced; NumPy PCG64 is used."""
    rng = np.random.Generator(np.random.PCG64(rndseed))
    return np.stack([rng.permutation(n) for _ in range(nperms)], axis=1).astype(np.int32) if nperms > 0 \
        else np.zeros((n, 0), dtype=np.int32)


def transform_permute(r0, nperms: int = 1024, rndseed: int = 0, original: bool = True,
                      perm_idx: Optional[np.ndarray] = None):
    """src/transform_helpers.jl:94-102 + src/util.jl:162-179 -- first column is the original."""
    r0 = _mat(r0)
    x = r0[:, 0]
    if perm_idx is None:
        perm_idx = make_perm_idx(x.shape[0], nperms, rndseed)
    cols = [x] if original else []
    for i in range(perm_idx.shape[1]):
        cols.append(x[perm_idx[:, i]])
    return np.stack(cols, axis=1)


# ----------------------------------------------------------------------------------------------
# src/bulkscan_helpers.jl
# ----------------------------------------------------------------------------------------------

def computeR_LMM(wY, wX, wIntercept):
    """src/bulkscan_helpers.jl:47-64."""
    Y00 = resid(wY, wIntercept)
    X00 = resid(wX, wIntercept)
    norm_Y = np.linalg.norm(Y00, axis=0)
    norm_X = np.linalg.norm(X00, axis=0)
    Y00 = colDivide(Y00, norm_Y)
    X00 = colDivide(X00, norm_X)
    return X00.T @ Y00


def univar_liteqtl(y0_j, X0_intercept, X0_covar, lambda0, prior_variance: float = 0.0, prior_sample_size: float = 0.0,
                   reml: bool = False, optim_interval: int = 1, h2_override: Optional[float] = None):
    """src/bulkscan_helpers.jl:127-150 -- returns (R = p x 1 LODs, h2).
    `h2_override` (test hook, not in the reference) skips fitlmm and evaluates lines 138-146 at the given h2,
    so the LOD arithmetic can be checked separately from the optimiser's sqrt(eps) noise in h2."""
    y0 = _mat(y0_j)
    n = y0.shape[0]
    if h2_override is None:
        vc = fitlmm(y0, X0_intercept, lambda0, [prior_variance, prior_sample_size], reml=reml, optim_interval=optim_interval)
    else:
        vc = LMMEstimates(np.zeros((1, 1)), float("nan"), float(h2_override), float("nan"))
    sqrtw = np.sqrt(np.abs(makeweights(vc.h2, lambda0)))
    wy0 = rowMultiply(y0, sqrtw)
    wX0_intercept = rowMultiply(X0_intercept, sqrtw)
    wX0_covar = rowMultiply(X0_covar, sqrtw)
    R = computeR_LMM(wy0, wX0_covar, wX0_intercept)
    return r2lod(R, n), vc.h2


def weighted_liteqtl(Y0, X0, lambda0, hsq: float, num_of_covar: int = 1):
    """src/bulkscan_helpers.jl:175-201."""
    Y0 = _mat(Y0)
    X0 = _mat(X0)
    n = Y0.shape[0]
    sqrtw = np.sqrt(np.abs(makeweights(hsq, lambda0)))
    wY0 = rowMultiply(Y0, sqrtw)
    wX0 = rowMultiply(X0, sqrtw)
    wX0_intercept = wX0[:, :num_of_covar]
    wX0_covar = wX0[:, num_of_covar:]
    return r2lod(computeR_LMM(wY0, wX0_covar, wX0_intercept), n)


def find_optim_h2(h2_list, results):
    """src/bulkscan_helpers.jl:204-211 -- first maximum wins."""
    idx = np.argmax(results, axis=0)
    return _f64(h2_list)[idx], idx


class ResultsByBin(NamedTuple):
    idxs_by_bin: list
    LODs_by_bin: list
    h2_taken: list
    ell: np.ndarray


def gridscan_by_bin(pheno, geno, covar, kinship, grid, addIntercept: bool = True, prior_variance: float = 1.0,
                    prior_sample_size: float = 0.0, reml: bool = False, decomp_scheme: str = "eigen") -> ResultsByBin:
    """src/bulkscan_helpers.jl:239-292."""
    pheno = _mat(pheno)
    covar = _mat(covar)
    Y0, X0, lambda0 = transform_rotation(pheno, np.hstack([covar, _mat(geno)]), kinship,
                                         addIntercept=addIntercept, decomp_scheme=decomp_scheme)
    prior = [prior_variance, prior_sample_size]
    num_of_covar = covar.shape[1] + 1 if addIntercept else covar.shape[1]
    X0_intercept = X0[:, :num_of_covar]
    ell_rows = []
    for h in grid:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ell_rows.append(wls_multivar(Y0, X0_intercept, makeweights(h, lambda0), prior, reml=reml).Ell)
    ell_results = np.vstack(ell_rows)
    optim_h2, _ = find_optim_h2(grid, ell_results)
    h2_taken = list(dict.fromkeys(optim_h2.tolist()))  # `unique`, order irrelevant to the result
    idxs, lods = [], []
    for h in h2_taken:
        mask = optim_h2 == h
        idxs.append(mask)
        lods.append(weighted_liteqtl(Y0[:, mask], X0, lambda0, h, num_of_covar=num_of_covar))
    return ResultsByBin(idxs, lods, h2_taken, ell_results)


# ----------------------------------------------------------------------------------------------
# src/bulkscan.jl
# ----------------------------------------------------------------------------------------------

def _apply_weights(Y, G, Covar, K, weights, addIntercept):
    """src/bulkscan.jl:231-250 (same block at :351-370, :457-476 and src/scan.jl:201-221)."""
    if weights is None:
        return Y, G, Covar, K, addIntercept
    w = _f64(weights).ravel()
    W = np.diag(w)
    Y_st = W @ Y
    G_st = W @ G
    if addIntercept:
        Covar_st = W @ np.hstack([np.ones((Y.shape[0], 1)), Covar])
    else:
        Covar_st = W @ Covar
    return Y_st, G_st, Covar_st, W @ K @ W, False


class BulkscanNullResult(NamedTuple):
    L: np.ndarray
    h2_null_list: np.ndarray


class BulkscanAltResult(NamedTuple):
    L: np.ndarray
    h2_panel: np.ndarray


def bulkscan_null(Y, G, K, Covar=None, addIntercept: bool = True, weights=None, prior_variance: float = 1.0,
                  prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1,
                  decomp_scheme: str = "eigen", nb: int = 1, nt_blas: int = 1, h2_override=None) -> BulkscanNullResult:
    """src/bulkscan.jl:188-210, 212-314 (null-exact).  `h2_override` (m values): test hook, see univar_liteqtl.  `nb`/`nt_blas` only change thread blocking in the
    reference; the per-trait arithmetic does not depend on them."""
    Y = _mat(Y)
    G = _mat(G)
    K = _mat(K)
    n = Y.shape[0]
    if Covar is None:
        Covar = np.ones((n, 1))
        addIntercept = False
    Covar = _mat(Covar)
    m = Y.shape[1]
    p = G.shape[1]
    num_of_covar = Covar.shape[1] + 1 if addIntercept else Covar.shape[1]
    Y_st, G_st, Covar_st, K_st, addIntercept = _apply_weights(Y, G, Covar, K, weights, addIntercept)
    Y0, X0, lambda0 = transform_rotation(Y_st, np.hstack([Covar_st, G_st]), K_st, addIntercept=addIntercept,
                                         decomp_scheme=decomp_scheme)
    X0_intercept = X0[:, :num_of_covar]
    X0_covar = X0[:, num_of_covar:]
    L = np.empty((p, m))
    h2 = np.zeros(m)
    for j in range(m):
        R, h = univar_liteqtl(Y0[:, j], X0_intercept, X0_covar, lambda0, prior_variance=prior_variance,
                              prior_sample_size=prior_sample_size, reml=reml, optim_interval=optim_interval,
                              h2_override=None if h2_override is None else float(h2_override[j]))
        L[:, j] = R[:, 0]
        h2[j] = h
    return BulkscanNullResult(L, h2)


def bulkscan_null_grid(Y, G, K, grid_list, Covar=None, addIntercept: bool = True, weights=None,
                       prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False,
                       decomp_scheme: str = "eigen") -> BulkscanNullResult:
    """src/bulkscan.jl:321-385 (null-grid) incl. reorder_results (src/bulkscan_helpers.jl:294-308) and
    get_h2_distribution (src/bulkscan.jl:387-397)."""
    Y = _mat(Y)
    G = _mat(G)
    K = _mat(K)
    n = Y.shape[0]
    if Covar is None:
        Covar = np.ones((n, 1))
        addIntercept = False
    Covar = _mat(Covar)
    m = Y.shape[1]
    p = G.shape[1]
    Y_st, G_st, Covar_st, K_st, addIntercept = _apply_weights(Y, G, Covar, K, weights, addIntercept)
    res = gridscan_by_bin(Y_st, G_st, Covar_st, K_st, list(grid_list), addIntercept=addIntercept,
                          prior_variance=prior_variance, prior_sample_size=prior_sample_size, reml=reml,
                          decomp_scheme=decomp_scheme)
    L = np.empty((p, m))
    h2 = np.zeros(m)
    for mask, lod, h in zip(res.idxs_by_bin, res.LODs_by_bin, res.h2_taken):
        L[:, mask] = lod
        h2[mask] = h
    return BulkscanNullResult(L, h2)


def bulkscan_alt_grid(Y, G, K, hsq_list, Covar=None, addIntercept: bool = True, weights=None,
                      prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False,
                      decomp_scheme: str = "eigen", compat_counter_quirk: bool = False, return_tables: bool = False):
    """src/bulkscan.jl:428-526 (alt-grid) with `tmax!` (src/bulkscan_helpers.jl:330-350).
    `return_tables` (test hook, not in the reference): also return the stack logL1[g, i, j] of every grid point, so that a
    test can tell a genuine arg-max disagreement from a tie decided at rounding level.

    Deviations, both flagged in SURVEY.md Appendix B: (B1) `num_of_covar` is passed for every grid point
    (the reference omits it at src/bulkscan.jl:510, which makes c>1 fail with a dimension error);
    (B2) `h2_panel` is the grid value at the first arg-max; `compat_counter_quirk=True` reproduces the
    reference's improvement-counter indexing (src/bulkscan_helpers.jl:342-343)."""
    Y = _mat(Y)
    G = _mat(G)
    K = _mat(K)
    n = Y.shape[0]
    if Covar is None:
        Covar = np.ones((n, 1))
        addIntercept = False
    Covar = _mat(Covar)
    p = G.shape[1]
    m = Y.shape[1]
    hsq_list = [float(h) for h in hsq_list]
    num_of_covar = Covar.shape[1] + 1 if addIntercept else Covar.shape[1]
    Y_st, G_st, Covar_st, K_st, addIntercept = _apply_weights(Y, G, Covar, K, weights, addIntercept)
    Y0, X0, lambda0 = transform_rotation(Y_st, np.hstack([Covar_st, G_st]), K_st, addIntercept=addIntercept,
                                         decomp_scheme=decomp_scheme)
    X0_base = X0[:, :num_of_covar]
    prior = [prior_variance, prior_sample_size]
    ln10 = math.log(10.0)

    def one(h):
        logLR = weighted_liteqtl(Y0, X0, lambda0, h, num_of_covar=num_of_covar) * ln10
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            logL0 = wls_multivar(Y0, X0_base, makeweights(h, lambda0), prior, reml=reml).Ell
        return logLR + np.repeat(logL0, p, axis=0), logL0

    logL1, logL0 = one(hsq_list[0])
    logL0_all = np.zeros((len(hsq_list), m))
    logL0_all[0, :] = logL0
    h2_panel = np.ones((p, m)) * hsq_list[0]
    counter = np.ones((p, m), dtype=np.int64)
    tables = [logL1.copy()] if return_tables else None
    for k, h in enumerate(hsq_list[1:], start=1):
        logL1_k, logL0_k = one(h)
        if return_tables:
            tables.append(logL1_k)
        logL0_all[k, :] = logL0_k
        better = logL1 < logL1_k
        logL1 = np.where(better, logL1_k, logL1)
        if compat_counter_quirk:
            counter = counter + better
            h2_panel = np.where(better, np.asarray(hsq_list)[np.minimum(counter, len(hsq_list)) - 1], h2_panel)
        else:
            h2_panel = np.where(better, h, h2_panel)
    logL0_opt = np.max(logL0_all, axis=0, keepdims=True)
    L = (logL1 - logL0_opt) / ln10
    if return_tables:
        return BulkscanAltResult(L, h2_panel), np.stack(tables)
    return BulkscanAltResult(L, h2_panel)


def bulkscan(Y, G, K, Covar=None, method: str = "null-grid", h2_grid=None, addIntercept: bool = True, weights=None,
             prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1,
             decomp_scheme: str = "eigen", output_pvals: bool = False, chisq_df: int = 1, nb: int = 1, nt_blas: int = 1):
    """src/bulkscan.jl:81-162 -- dispatcher; returns a dict with the reference's NamedTuple field names."""
    if h2_grid is None:
        h2_grid = [i / 10.0 for i in range(10)]  # collect(0.0:0.1:0.9)
    if method == "null-exact":
        r = bulkscan_null(Y, G, K, Covar=Covar, addIntercept=addIntercept, weights=weights, prior_variance=prior_variance,
                          prior_sample_size=prior_sample_size, reml=reml, optim_interval=optim_interval,
                          decomp_scheme=decomp_scheme)
        out = {"L": r.L, "h2_null_list": r.h2_null_list}
    elif method == "null-grid":
        r = bulkscan_null_grid(Y, G, K, h2_grid, Covar=Covar, addIntercept=addIntercept, weights=weights,
                               prior_variance=prior_variance, prior_sample_size=prior_sample_size, reml=reml,
                               decomp_scheme=decomp_scheme)
        out = {"L": r.L, "h2_null_list": r.h2_null_list}
    elif method == "alt-grid":
        r = bulkscan_alt_grid(Y, G, K, h2_grid, Covar=Covar, addIntercept=addIntercept, weights=weights,
                              prior_variance=prior_variance, prior_sample_size=prior_sample_size, reml=reml,
                              decomp_scheme=decomp_scheme)
        out = {"L": r.L, "h2_panel": r.h2_panel}
    else:
        # the reference falls through to an UndefVarError (src/bulkscan.jl:126-154)
        raise BulkLMMError("Unknown method `%s`; choose null-exact, null-grid or alt-grid." % method)
    if output_pvals:
        out["log10Pvals_mat"] = lod2log10p(out["L"], chisq_df)
        out["Chisq_df"] = chisq_df
    return out


# ----------------------------------------------------------------------------------------------
# src/scan.jl (single trait: the independent RSS-form cross-check, and the permutation GEMM)
# ----------------------------------------------------------------------------------------------

def scan_null(y, g, covar, K, prior, addIntercept: bool, reml: bool = False, method: str = "qr", optim_interval: int = 1,
              decomp_scheme: str = "eigen"):
    """src/scan.jl:310-360 -- per-marker RSS ratio; lod[i] = (-n/2)*(log10 rss1 - log10 rss0)."""
    y = _mat(y)
    g = _mat(g)
    covar = _mat(covar)
    (n, p) = g.shape
    num_of_covar = covar.shape[1] + 1 if addIntercept else covar.shape[1]
    y0, X0, lambda0 = transform_rotation(y, np.hstack([covar, g]), K, addIntercept=addIntercept, decomp_scheme=decomp_scheme)
    X0_covar = X0[:, :num_of_covar]
    out00 = fitlmm(y0, X0_covar, lambda0, prior, reml=reml, method=method, optim_interval=optim_interval)
    sqrtw = np.sqrt(makeweights(out00.h2, lambda0))
    y0 = rowMultiply(y0, sqrtw)
    X0 = rowMultiply(X0, sqrtw)
    X0_covar = X0[:, :num_of_covar]
    rss0 = rss(y0, X0_covar, method=method)[0, 0]
    lod = np.zeros(p)
    X = X0[:, :num_of_covar + 1].copy()
    for i in range(p):
        X[:, num_of_covar] = X0[:, num_of_covar + i]
        rss1 = rss(y0, X, method=method)[0, 0]
        lod[i] = (-n / 2) * (np.log10(rss1) - np.log10(rss0))
    return {"sigma2_e": out00.sigma2, "h2_null": out00.h2, "lod": lod}


def scan_alt(y, g, covar, K, prior, addIntercept: bool, reml: bool = False, method: str = "qr", optim_interval: int = 1,
             decomp_scheme: str = "eigen", true_weights: bool = False, h2_each_override: Optional[np.ndarray] = None,
             h2_null_override: Optional[float] = None):
    """src/scan.jl:397-453 -- variance components re-estimated per marker.
    The closing `wls(y0, X, sqrtw_alt, prior)` / `wls(y0, X0_covar, sqrtw_null, prior)` (:434-435) are handed the SQUARE ROOTS
    of the weights as `w` and carry no `reml` keyword; that is restated as written.  `true_weights` (not in the reference)
    evaluates both at makeweights(h2).  `h2_each_override` / `h2_null_override`: test hooks (skip the Brent searches)."""
    y = _mat(y)
    g = _mat(g)
    covar = _mat(covar)
    (n, p) = g.shape
    num_of_covar = covar.shape[1] + 1 if addIntercept else covar.shape[1]
    y0, X0, lambda0 = transform_rotation(y, np.hstack([covar, g]), K, addIntercept=addIntercept, decomp_scheme=decomp_scheme)
    X0_covar = X0[:, :num_of_covar]
    pve_list = np.empty(p)
    out00 = fitlmm(y0, X0_covar, lambda0, prior, reml=reml, method=method, optim_interval=optim_interval)
    h2_null = out00.h2 if h2_null_override is None else float(h2_null_override)
    lod = np.zeros(p)
    X = X0[:, :num_of_covar + 1].copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(p):
            X[:, num_of_covar] = X0[:, num_of_covar + i]
            if h2_each_override is None:
                h2_alt = fitlmm(y0, X, lambda0, prior, reml=reml, method=method, optim_interval=optim_interval).h2
            else:
                h2_alt = float(h2_each_override[i])
            w_null = makeweights(h2_null, lambda0)
            w_alt = makeweights(h2_alt, lambda0)
            if not true_weights:
                w_null = np.sqrt(w_null)
                w_alt = np.sqrt(w_alt)
            wls_alt = wls(y0, X, w_alt, prior)
            wls_null = wls(y0, X0_covar, w_null, prior)
            lod[i] = (wls_alt.ell - wls_null.ell) / np.log(10)
            pve_list[i] = h2_alt
    return {"sigma2_e": out00.sigma2, "h2_null": out00.h2, "h2_each_marker": pve_list, "lod": lod}


def scan_perms_lite(y, g, covar, K, prior_variance: float = 1.0, prior_sample_size: float = 0.0, addIntercept: bool = True,
                    method: str = "qr", optim_interval: int = 1, nperms: int = 1024, rndseed: int = 0, reml: bool = False,
                    decomp_scheme: str = "eigen", perm_idx: Optional[np.ndarray] = None, h2_override: Optional[float] = None,
                    rotation_override=None):
    """src/scan.jl:485-557.
    `rotation_override = (y0, X0, lambda0)`: test hook (not in the reference).  Permuting the rotated residuals
    depends on the order and SIGN of the eigenvectors, which LAPACK leaves unspecified; to compare two
    implementations element-wise they must share the rotation."""
    y = _mat(y)
    g = _mat(g)
    covar = _mat(covar)
    if y.shape[1] != 1:
        raise BulkLMMError("Can only handle one trait.")
    n = g.shape[0]
    if rotation_override is None:
        y0, X0, lambda0 = transform_rotation(y, np.hstack([covar, g]), K, addIntercept=addIntercept, decomp_scheme=decomp_scheme)
    else:
        y0, X0, lambda0 = (np.array(a, dtype=np.float64) for a in rotation_override)
    n_covars = covar.shape[1] + 1 if addIntercept else covar.shape[1]
    r0, X00, sigma2_e, h2_null = transform_reweight(y0, X0, lambda0, n_covars=n_covars, prior_a=prior_variance,
                                                    prior_b=prior_sample_size, reml=reml, method=method,
                                                    optim_interval=optim_interval, h2_override=h2_override)
    if nperms < 0:
        raise BulkLMMError("The required number of permutations must be a positive integer.")
    r0perm = transform_permute(r0, nperms=nperms, rndseed=rndseed, original=True, perm_idx=perm_idx)
    norm_y = np.linalg.norm(r0perm, axis=0)
    norm_X = np.linalg.norm(X00, axis=0)
    r0perm = colDivide(r0perm, norm_y)
    X00 = colDivide(X00, norm_X)
    L = r2lod(X00.T @ r0perm, n)
    return {"sigma2_e": sigma2_e, "h2_null": h2_null, "lod": L[:, 0].copy(), "L_perms": L[:, 1:].copy()}


def scan(y, g, K, covar=None, weights=None, prior_variance: float = 0.0, prior_sample_size: float = 0.0,
         addIntercept: bool = True, reml: bool = False, assumption: str = "null", method: str = "qr", optim_interval: int = 1,
         permutation_test: bool = False, nperms: int = 1024, rndseed: int = 0, decomp_scheme: str = "eigen",
         output_pvals: bool = False, chisq_df: int = 1, perm_idx: Optional[np.ndarray] = None,
         h2_override: Optional[float] = None, rotation_override=None, **alt_kw):
    """src/scan.jl:94-271 -- single-trait API."""
    y = _mat(y)
    g = _mat(g)
    K = _mat(K)
    n = y.shape[0]
    if covar is None:
        if not addIntercept:
            raise BulkLMMError("Intercept has to be added when no other covariate is given.")
        covar = np.ones((n, 1))
        addIntercept = False
    covar = _mat(covar)
    if weights is not None:
        y, g, covar, K, addIntercept = _apply_weights(y, g, covar, K, weights, addIntercept)
    if assumption == "null":
        if permutation_test:
            res = scan_perms_lite(y, g, covar, K, prior_variance=prior_variance, prior_sample_size=prior_sample_size,
                                  addIntercept=addIntercept, reml=reml, method=method, optim_interval=optim_interval,
                                  nperms=nperms, rndseed=rndseed, decomp_scheme=decomp_scheme, perm_idx=perm_idx,
                                  h2_override=h2_override, rotation_override=rotation_override)
        else:
            res = scan_null(y, g, covar, K, [prior_variance, prior_sample_size], addIntercept, reml=reml, method=method,
                            optim_interval=optim_interval, decomp_scheme=decomp_scheme)
    elif assumption == "alt":
        if permutation_test:
            raise BulkLMMError("Permutation test option currently is not supported for the alternative assumption.")
        res = scan_alt(y, g, covar, K, [prior_variance, prior_sample_size], addIntercept, reml=reml, method=method,
                       optim_interval=optim_interval, decomp_scheme=decomp_scheme, **alt_kw)
    else:
        raise BulkLMMError("Assumption keyword is not supported. Please enter null or alt.")
    if output_pvals:
        res["log10pvals"] = lod2log10p(res["lod"], chisq_df)
    return res
