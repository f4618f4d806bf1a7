"""Shared helpers for the test-suite: seeded synthetic data and the parity criterion."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# parity criterion of SURVEY.md §8(c): |L_gpu - L_ref| <= 1e-6*|L_ref| + 1e-10 (fp64)
RTOL = 1e-6
ATOL = 1e-10


def bxd_kinship() -> np.ndarray:
    """The real BXD kinship (79 x 79) held by the reference's tests: test/ref_data_for_tests/kinship_ref.he
    (Helium: 56-byte header, column-major float64), rounded to 12 digits as test/kinship_test.jl:5 does."""
    raw = open(os.path.join(GOLDEN, "bxd_kinship_ref.he"), "rb").read()
    nrow, ncol = np.frombuffer(raw[:16], dtype="<i8")
    K = np.frombuffer(raw[56:56 + 8 * nrow * ncol], dtype="<f8").reshape((ncol, nrow)).T.copy()
    return np.round(K, 12)


def make_geno(n: int, p: int, rng) -> np.ndarray:
    """RIL-like genotype probabilities: two-state Markov chain along markers (switch prob 0.01),
    2 % of entries replaced by U(0,1)."""
    sw = rng.random((n, p)) < 0.01
    sw[:, 0] = rng.random(n) < 0.5
    G = (np.cumsum(sw, axis=1) % 2).astype(np.float64)
    unc = rng.random((n, p)) < 0.02
    G[unc] = rng.random(int(unc.sum()))
    return G


def kinship_of(G: np.ndarray) -> np.ndarray:
    X = G - 0.5
    K = 2.0 * (X @ X.T) / X.shape[1] + 0.5
    np.fill_diagonal(K, 1.0)
    return np.round(K, 12)


def make_data(n=79, p=300, m=40, seed=20240, bxd=True, ncov=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    G = make_geno(n, p, rng)
    K = bxd_kinship() if (bxd and n == 79) else kinship_of(G)
    lam, U = np.linalg.eigh(K)
    lam = np.maximum(lam, 0)
    h2 = rng.uniform(0.0, 0.9, size=m)
    g = (U * np.sqrt(lam)) @ rng.standard_normal((n, m)) * np.sqrt(h2 / (1 - h2))
    e = rng.standard_normal((n, m))
    Y = g + e
    causal = rng.random(m) < 0.3
    q = rng.integers(0, p, size=m)
    beta = rng.standard_normal(m) * 1.5
    Y[:, causal] += G[:, q[causal]] * beta[causal]
    Y += 10.0  # a non-zero mean (the intercept matters)
    Covar = rng.standard_normal((n, ncov)) if ncov else None
    if ncov:
        Y += Covar @ rng.standard_normal((ncov, m))
    return Y, G, K, Covar


def assert_lod_close(got, ref, rtol=RTOL, atol=ATOL, what="LOD"):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = rtol * np.abs(ref) + atol
    bad = ~(err <= bound)
    if bad.any():
        i = np.unravel_index(np.argmax(np.where(bad, err, 0)), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} outside |d| <= {rtol}*|ref| + {atol}; "
                             f"worst at {i}: got {got[i]!r} ref {ref[i]!r} |d| {err[i]:.3e}; max rel "
                             f"{np.nanmax(err / np.maximum(np.abs(ref), 1e-300)):.3e}")


def assert_h2_panel_ties_only(got_panel, ref_panel, logL1, grid, quirk=False, rel=1e-12, what="h2_panel"):
    """alt-grid: wherever the arg-max grid value differs from the oracle's, the two candidates must be TIED in the oracle's
    own logL1 table (logL1[g, i, j], O.bulkscan_alt_grid(..., return_tables=True)) to `rel` relative -- tmax!'s strict `<`
    (src/bulkscan_helpers.jl:330-350) then decides at rounding level, on either side.  With the improvement-counter quirk
    (B2) the panel value depends on every comparison of the running maximum: a mismatch needs a near-tie somewhere along
    the grid.  Returns the number of (tie-explained) mismatches."""
    grid = np.asarray(grid, dtype=np.float64)
    bad = np.argwhere(got_panel != ref_panel)
    for i, j in bad:
        col = logL1[:, i, j]
        scale = max(1.0, float(np.abs(col).max()))
        if quirk:
            run = np.maximum.accumulate(col)
            gaps = np.abs(col[1:] - run[:-1])
            assert gaps.min() <= rel * scale, f"{what}[{i},{j}]: counter differs without a near-tie (min gap {gaps.min():.3e})"
        else:
            gg = int(np.flatnonzero(grid == got_panel[i, j])[0])
            gr = int(np.flatnonzero(grid == ref_panel[i, j])[0])
            assert abs(col[gg] - col[gr]) <= rel * scale, \
                f"{what}[{i},{j}]: {got_panel[i, j]} vs {ref_panel[i, j]}: logL1 {col[gg]!r} vs {col[gr]!r} is not a tie"
    return len(bad)


class DevBuf:
    """A device buffer through the HIP runtime itself (ctypes on libamdhip64.so), for tests that drive the *_dev entry points of the
    C ABI in-process: torch cannot be initialised in a process whose HIP runtime the library brought up first on these boxes (the
    torch-based checks run as their own programs: tests/helpers/)."""
    _hip = None

    def __init__(self, arr=None, nbytes=None):
        import ctypes as C
        if DevBuf._hip is None:
            DevBuf._hip = C.CDLL("libamdhip64.so")
        self.nbytes = int(arr.nbytes if arr is not None else nbytes)
        p = C.c_void_p()
        assert DevBuf._hip.hipMalloc(C.byref(p), C.c_size_t(max(self.nbytes, 8))) == 0
        self.ptr = p.value
        if arr is not None:
            a = np.ascontiguousarray(arr)
            assert DevBuf._hip.hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0

    def fill(self, arr):
        import ctypes as C
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes and DevBuf._hip.hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0

    def get(self, shape, dtype=np.float64):
        import ctypes as C
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes and DevBuf._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        import ctypes as C
        if self.ptr:
            DevBuf._hip.hipFree(C.c_void_p(self.ptr))
            self.ptr = None
