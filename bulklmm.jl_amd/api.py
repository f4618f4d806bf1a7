"""Host-side mirror of BulkLMM.jl's API for the bulkscan hot path, over the C ABI of libbulklmm_hip.so.

Same names, argument meaning, defaults, return field names and error strings as the reference
(src/bulkscan.jl:81-162,188-314,321-397,428-526; src/scan.jl:94-271,485-557; src/kinship.jl:4-14;
src/transform_helpers.jl:1-54; src/bulkscan_helpers.jl:175-201), so the parity tests read like the
reference's own tests.  All arithmetic happens on the GPU; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import warnings
from typing import NamedTuple, Optional

import numpy as np

from . import _lib as L


class BulkLMMError(Exception):
    """Julia ErrorException stand-in; `.msg` is the reference's message, `.code` the blmm_err."""

    def __init__(self, msg: str, code: int = -1):
        super().__init__(msg)
        self.msg = msg
        self.code = code


def _stream_arg(stream: Optional[int]):
    """None -> NULL (private stream); 0 -> BLMM_STREAM_NULL (adopt the legacy default stream); else the handle."""
    if stream is None:
        return None
    if int(stream) == 0:
        return C.c_void_p(-1)  # BLMM_STREAM_NULL, include/bulklmm_hip.h
    return C.c_void_p(int(stream))


class Context:
    """One GPU.  Not thread-safe (include/bulklmm_hip.h)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        """stream=None: a private non-blocking stream of the library (no ordering against the caller's streams);
        an integer hipStream_t handle: every *_dev call enqueues there.  The handle 0 -- torch.cuda's default stream,
        the legacy null stream -- is adopted as such (BLMM_STREAM_NULL), NOT replaced by a private stream."""
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.blmm_create(int(device), _stream_arg(stream), C.byref(h))
        if rc != 0:
            raise BulkLMMError(self.lib.blmm_err_string(rc).decode(), rc)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.blmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            msg = self.lib.blmm_last_error(self.h).decode() or self.lib.blmm_err_string(rc).decode()
            raise BulkLMMError(msg, rc)

    def set_timing(self, on: bool):
        self.check(self.lib.blmm_set_timing(self.h, 1 if on else 0))

    def read_timings(self):
        """Sum of per-phase device times (ms) over the calls since the last read, and their count."""
        sums = (C.c_double * 6)()
        cnt = C.c_int64(0)
        self.check(self.lib.blmm_read_timings(self.h, sums, C.byref(cnt)))
        names = ("eigen", "rotate", "h2", "prep", "scan", "total")
        return {k: float(v) for k, v in zip(names, sums)}, int(cnt.value)

    def lowrank_profile(self):
        """blmm_lowrank_profile: (traits of the shared-weights class, [(traits, rank) per segment of the heritability axis]) of the
        last null-exact call -- what its low-rank weights form executed."""
        out = (C.c_int64 * 18)()
        self.check(self.lib.blmm_lowrank_profile(self.h, out))
        return int(out[1]), [(int(out[2 + 2 * s]), int(out[3 + 2 * s])) for s in range(int(out[0]))]

    def lowrank_columns(self, m: int):
        """blmm_lowrank_columns: (panel column of each of the m traits, region width, [shared, others] per region) of the last
        null-exact call's low-rank form."""
        col = np.empty(m, dtype=np.int32)
        w = C.c_int64(0)
        cnt = (C.c_int64 * 4)()
        self.check(self.lib.blmm_lowrank_columns(self.h, int(m), col.ctypes.data_as(C.c_void_p), C.byref(w), cnt))
        return col, int(w.value), [int(x) for x in cnt]

    def set_tuning(self, key: str, value: float):
        """blmm_set_tuning: the switches that select another arithmetic path (include/bulklmm_hip.h); key "defaults" resets."""
        self.check(self.lib.blmm_set_tuning(self.h, key.encode(), float(value)))

    def get_tuning(self, key: str) -> float:
        v = C.c_double(0.0)
        if self.lib.blmm_get_tuning(self.h, key.encode(), C.byref(v)) != 0:
            raise BulkLMMError("get_tuning: unknown key " + key)
        return float(v.value)

    def set_stream(self, stream: Optional[int]):
        self.check(self.lib.blmm_set_stream(self.h, _stream_arg(stream)))

    def synchronize(self):
        self.check(self.lib.blmm_synchronize(self.h))


class MultiContext:
    """Several GPUs of one node behind ONE call (blmm_create_multi): one host worker thread and one blmm_ctx per device.
    `devices=None`: every visible device; an id may be repeated (several shards on one GPU)."""

    def __init__(self, devices=None):
        self.lib = L.load()
        h = C.c_void_p()
        if devices is None:
            rc = self.lib.blmm_create_multi(None, 0, C.byref(h))
        else:
            ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = self.lib.blmm_create_multi(ids, len(devices), C.byref(h))
        if rc != 0:
            raise BulkLMMError(self.lib.blmm_err_string(rc).decode(), rc)
        self.h = h
        self.ndev = int(self.lib.blmm_multi_ndev(h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.blmm_destroy_multi(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            msg = self.lib.blmm_multi_last_error(self.h).decode() or self.lib.blmm_err_string(rc).decode()
            raise BulkLMMError(msg, rc)

    def shard(self, m: int, rank: int):
        lo, hi = C.c_int64(0), C.c_int64(0)
        self.lib.blmm_multi_shard(int(m), int(rank), self.ndev, C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    def device_result(self, rank: int):
        """(device pointer of L, ld, col_lo, col_hi, device pointer of h2) after a gather='none' / 'allgather' call."""
        dL, dH = C.c_void_p(), C.c_void_p()
        ld, lo, hi = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.check(self.lib.blmm_multi_device_result(self.h, int(rank), C.byref(dL), C.byref(ld), C.byref(lo), C.byref(hi), C.byref(dH)))
        return dL.value, int(ld.value), int(lo.value), int(hi.value), dH.value


    def last_colmax(self):
        """Per-trait maximum LOD and its marker (0-based) of the last bulkscan_multi call, reduced on the devices."""
        m = self._last_m
        mx = np.empty(m); arg = np.empty(m, dtype=np.int64)
        self.check(self.lib.blmm_multi_last_colmax(self.h, _p(mx), arg.ctypes.data_as(C.c_void_p)))
        return mx, arg

    def last_lod_threshold(self, thr: float, cap: int = 1 << 16):
        """(marker, trait, LOD) of every LOD > thr of the last bulkscan_multi call, sorted by (trait, marker)."""
        while True:
            ii = np.empty(cap, dtype=np.int32); jj = np.empty(cap, dtype=np.int32); ll = np.empty(cap)
            cnt = C.c_int64(0)
            self.check(self.lib.blmm_multi_last_lod_threshold(self.h, float(thr), cap, ii.ctypes.data_as(C.c_void_p),
                                                              jj.ctypes.data_as(C.c_void_p), _p(ll), C.byref(cnt)))
            if cnt.value <= cap:
                break
            cap = int(cnt.value)
        k = int(cnt.value)
        order = np.lexsort((ii[:k], jj[:k]))
        return ii[:k][order], jj[:k][order], ll[:k][order]


_GATHER = {"none": L.BLMM_GATHER_NONE, "host_shards": L.BLMM_GATHER_HOST_SHARDS, "allgather": L.BLMM_GATHER_ALLGATHER}

_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


MAX_INDIVIDUALS = 2048   # blmm_api.hip: prepare_eigen -- the device eigensolver (tridiagonalisation + divide and conquer) stops there


def _check_n(n: int):
    """The one documented capability gap against the reference (whose LAPACK eigen has no size limit, src/transform_helpers.jl:21-34):
    refused HERE, before anything is uploaded, with the library's own message and code."""
    if n > MAX_INDIVIDUALS:
        raise BulkLMMError("more than 2048 individuals: the device eigensolver (tridiagonalisation + divide and conquer) stops at n = 2048", -10)


def _F(a, ndim=2) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    if ndim == 2 and a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.asfortranarray(a)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _opts(method=L.BLMM_NULL_GRID, reml=False, addIntercept=True, decomp_scheme="eigen", optim_interval=1,
          prior_variance=1.0, prior_sample_size=0.0, compat_flags=0) -> L.blmm_opts:
    if decomp_scheme == "eigen":
        d = L.BLMM_EIGEN
    elif decomp_scheme == "svd":
        d = L.BLMM_SVD
    else:
        d = 99  # the library raises the reference's message (src/transform_helpers.jl:51)
    return L.blmm_opts(int(method), int(bool(reml)), int(bool(addIntercept)), d, int(optim_interval), int(compat_flags),
                       float(prior_variance), float(prior_sample_size))


def _raise_status(st: L.blmm_status):
    """Re-issue the reference's warnings / errors from the device counters."""
    if st.n_neg_eig:
        warnings.warn("Negative eigenvalues exist. The kinship matrix supplied may not be SPD.")  # src/transform_helpers.jl:29
    if st.n_nonpos_weight:
        warnings.warn("Some weights are not positive.")  # src/wls.jl:36
    if st.n_zero_norm:
        raise BulkLMMError(L.ERR_ZERO_NORM_MSG, -8)  # src/util.jl:70


class BulkscanNullResult(NamedTuple):
    L: np.ndarray
    h2_null_list: np.ndarray


class BulkscanAltResult(NamedTuple):
    L: np.ndarray
    h2_panel: np.ndarray


_METHODS = {"null-exact": L.BLMM_NULL_EXACT, "null-grid": L.BLMM_NULL_GRID, "alt-grid": L.BLMM_ALT_GRID}


def calcKinship(geno, ctx: Optional[Context] = None, digits: Optional[int] = None) -> np.ndarray:
    """src/kinship.jl:4-14.  `digits=12` gives `round.(calcKinship(geno), digits = 12)`, the README's convention
    (README.md:176-181), rounded on the device."""
    ctx = ctx or default_context()
    G = _F(geno)
    n, p = G.shape
    K = np.empty((n, n), dtype=np.float64, order="F")
    if digits is None:
        ctx.check(ctx.lib.blmm_kinship(ctx.h, _p(G), n, p, _p(K)))
    else:
        ctx.check(ctx.lib.blmm_kinship_rounded(ctx.h, _p(G), n, p, int(digits), _p(K)))
    return K


def _read_table(kind: str, path: str, *args) -> np.ndarray:
    lib = L.load()
    h = C.c_void_p()
    rc = lib.blmm_read_he(path.encode(), C.byref(h)) if kind == "he" else lib.blmm_read_csv(path.encode(), *args, C.byref(h))
    if rc != 0:
        raise BulkLMMError("could not read %s (%s)" % (path, lib.blmm_err_string(rc).decode()), rc)
    try:
        out = np.empty((int(lib.blmm_table_rows(h)), int(lib.blmm_table_cols(h))), order="F")
        lib.blmm_table_copy(h, _p(out))
    finally:
        lib.blmm_table_free(h)
    return out


def readGenoProb(file: str) -> np.ndarray:
    """src/readData.jl:41-70 (getmarkernames = getids = true): header line and id column dropped."""
    return _read_table("csv", file, 1, 1, 1, 0)


def readGenoProb_ExcludeComplements(file: str) -> np.ndarray:
    """src/readData.jl:85-96: the odd (1-based) probability columns of readGenoProb."""
    return _read_table("csv", file, 1, 1, 2, 0)


def readBXDpheno(file: str) -> np.ndarray:
    """src/readData.jl:159-161: readdlm(file, ','; skipstart=1)[:, 2:end-1]."""
    return _read_table("csv", file, 1, 1, 1, 1)


def readBXDgeno(file: str, skipstart: int = 1) -> np.ndarray:
    """src/readData.jl:163-165: readdlm(file, ','; skipstart)[:, 2:2:end]."""
    return _read_table("csv", file, int(skipstart), 1, 2, 0)


def readhe(file: str) -> np.ndarray:
    """Helium .he matrix (test/kinship_test.jl:5)."""
    return _read_table("he", file)


class DeviceLOD:
    """The LOD matrix of a `keep_on_device=True` call: it stays in the context's HBM workspace (blmm_bulkscan with L_out == NULL)
    and is reduced there -- what README.md:246-255, 354-359 and get_thresholds do with L -- until the context's next call that
    produces a matrix.  `shape`, `colmax()`, `threshold(t)`, `get_thresholds(probs)`, `columns(idx)`, `log10p(df)`,
    `to_host()`."""

    def __init__(self, ctx: Context, p: int, m: int):
        self.ctx, self.shape = ctx, (p, m)

    def _alive(self):
        pp, mm = C.c_int64(0), C.c_int64(0)
        if self.ctx.lib.blmm_last_dims(self.ctx.h, C.byref(pp), C.byref(mm)) != 0 or (pp.value, mm.value) != self.shape:
            raise BulkLMMError("the device-resident LOD matrix has been replaced by a later call on its context")

    def colmax(self):
        self._alive()
        m = self.shape[1]
        mx = np.empty(m); arg = np.empty(m, dtype=np.int64)
        self.ctx.check(self.ctx.lib.blmm_last_lod_colmax(self.ctx.h, _p(mx), arg.ctypes.data_as(C.c_void_p)))
        return mx, arg

    def threshold(self, thr: float, cap: int = 1 << 16):
        """(marker, trait, LOD) triplets of every LOD > thr, 0-based, sorted by (trait, marker)."""
        self._alive()
        while True:
            ii = np.empty(cap, dtype=np.int32); jj = np.empty(cap, dtype=np.int32); ll = np.empty(cap)
            cnt = C.c_int64(0)
            self.ctx.check(self.ctx.lib.blmm_last_lod_threshold(self.ctx.h, float(thr), cap, ii.ctypes.data_as(C.c_void_p),
                                                                jj.ctypes.data_as(C.c_void_p), _p(ll), C.byref(cnt)))
            if cnt.value <= cap:
                break
            cap = int(cnt.value)
        k = int(cnt.value)
        order = np.lexsort((ii[:k], jj[:k]))
        return ii[:k][order], jj[:k][order], ll[:k][order]

    def get_thresholds(self, probs):
        self._alive()
        pr = np.ascontiguousarray(np.asarray(probs, dtype=np.float64).ravel())
        out = np.empty(pr.shape[0])
        self.ctx.check(self.ctx.lib.blmm_last_get_thresholds(self.ctx.h, _p(pr), pr.shape[0], _p(out)))
        return out

    def columns(self, idx):
        """L[:, idx] (p x len(idx)) -- the LOD profiles of a few traits, the only part of L that crosses PCIe."""
        self._alive()
        ix = np.ascontiguousarray(np.asarray(idx, dtype=np.int64).ravel())
        out = np.empty((self.shape[0], ix.shape[0]), order="F")
        self.ctx.check(self.ctx.lib.blmm_last_lod_columns(self.ctx.h, ix.ctypes.data_as(C.c_void_p), ix.shape[0], _p(out)))
        return out

    def log10p(self, df: int = 1):
        self._alive()
        return _last_log10p(self.ctx, self.shape, df)

    def to_host(self):
        return self.columns(np.arange(self.shape[1]))


def _bulkscan_call(method, Y, G, K, Covar, h2_grid, addIntercept, weights, prior_variance, prior_sample_size, reml,
                   optim_interval, decomp_scheme, compat_flags, ctx, return_status=False, keep_on_device=False, pvals_df=None):
    Y = _F(Y)
    G = _F(G)
    K = _F(K)
    n, m = Y.shape
    p = G.shape[1]
    if G.shape[0] != n or K.shape[0] != n or K.shape[1] != n:
        raise BulkLMMError("Dimension mismatch.", -2)  # src/transform_helpers.jl:9-11
    _check_n(n)
    cov = None
    ncov = 0
    if Covar is not None:
        cov = _F(Covar)
        if cov.shape[0] != n:
            raise BulkLMMError("Dimension mismatch.", -2)
        ncov = cov.shape[1]
    else:
        addIntercept = True  # bulkscan(Y, G, K): ones(n,1) as the only covariate (src/bulkscan.jl:97-104)
    w = None if weights is None else np.ascontiguousarray(np.asarray(weights, dtype=np.float64).ravel())
    if w is not None and w.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    grid = None
    ngrid = 0
    if method != L.BLMM_NULL_EXACT:
        grid = np.ascontiguousarray(np.asarray(h2_grid, dtype=np.float64).ravel())
        ngrid = grid.shape[0]
    o = _opts(method, reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size, compat_flags)
    ctx = ctx or default_context()  # after the argument checks: those must not need a GPU
    Lout = None if keep_on_device else np.empty((p, m), dtype=np.float64, order="F")
    h2 = np.empty((p, m) if method == L.BLMM_ALT_GRID else (m,), dtype=np.float64, order="F")
    st = L.blmm_status()
    # `output_pvals` (src/bulkscan.jl:154-157) is asked for right in front of the call -- after every check above, so that no
    # request is left armed by an exception -- and the library consumes it first thing in the call whatever happens next
    if pvals_df is not None:
        ctx.check(ctx.lib.blmm_set_log10p_output(ctx.h, None, 0, int(pvals_df)))
    try:
        ctx.check(ctx.lib.blmm_bulkscan(ctx.h, C.byref(o), _p(Y), n, m, _p(G), p, _p(cov), ncov, _p(K), _p(w), _p(grid), ngrid,
                                        _p(Lout), _p(h2), C.byref(st)))
    finally:
        if pvals_df is not None:
            ctx.lib.blmm_set_log10p_output(ctx.h, None, 0, 0)
    _raise_status(st)
    if keep_on_device:
        Lout = DeviceLOD(ctx, p, m)
    if return_status:
        return Lout, h2, st
    return Lout, h2


def host_register(a: np.ndarray):
    """Pin the pages of an existing array (blmm_host_register): as an output it is then filled at PCIe link rate."""
    lib = L.load()
    if lib.blmm_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes) != 0:
        raise BulkLMMError("hipHostRegister failed", -12)


def host_unregister(a: np.ndarray):
    L.load().blmm_host_unregister(a.ctypes.data_as(C.c_void_p))


def bulkscan_into(ctx: Context, method: int, Y, G, K, L_out: np.ndarray, h2_out: Optional[np.ndarray] = None, *, h2_grid=None,
                  prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1):
    """blmm_bulkscan (host pointers) writing into caller-provided Fortran-ordered outputs -- what a Julia caller that
    reuses its result Array does; used by bench.py for the end-to-end time."""
    Y = _F(Y); G = _F(G); K = _F(K)
    n, m = Y.shape
    p = G.shape[1]
    assert L_out.flags.f_contiguous and L_out.shape == (p, m)
    if h2_out is None:
        h2_out = np.empty((p, m) if method == L.BLMM_ALT_GRID else (m,), order="F")
    grid, ngrid = None, 0
    if method != L.BLMM_NULL_EXACT:
        grid = np.ascontiguousarray(np.asarray(h2_grid if h2_grid is not None else [i / 10.0 for i in range(10)], dtype=np.float64))
        ngrid = grid.shape[0]
    o = _opts(method, reml, True, "eigen", optim_interval, prior_variance, prior_sample_size)
    ctx.check(ctx.lib.blmm_bulkscan(ctx.h, C.byref(o), _p(Y), n, m, _p(G), p, None, 0, _p(K), None, _p(grid), ngrid,
                                    _p(L_out), _p(h2_out), None))
    return L_out, h2_out


def bulkscan_null(Y, G, K, Covar=None, *, nb: int = 1, nt_blas: int = 1, addIntercept: bool = True, weights=None,
                  prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1,
                  decomp_scheme: str = "eigen", ctx: Optional[Context] = None, keep_on_device: bool = False,
                  _pvals_df=None) -> BulkscanNullResult:
    """src/bulkscan.jl:188-314.  `nb` / `nt_blas` are accepted and ignored (CPU thread blocking).  keep_on_device: `L` is a
    DeviceLOD (the matrix stays in HBM)."""
    Lo, h2 = _bulkscan_call(L.BLMM_NULL_EXACT, Y, G, K, Covar, None, addIntercept, weights, prior_variance, prior_sample_size,
                            reml, optim_interval, decomp_scheme, 0, ctx, keep_on_device=keep_on_device, pvals_df=_pvals_df)
    return BulkscanNullResult(Lo, h2)


def bulkscan_null_grid(Y, G, K, grid_list, Covar=None, *, weights=None, addIntercept: bool = True, prior_variance: float = 1.0,
                       prior_sample_size: float = 0.0, reml: bool = False, decomp_scheme: str = "eigen",
                       ctx: Optional[Context] = None, keep_on_device: bool = False, _pvals_df=None) -> BulkscanNullResult:
    """src/bulkscan.jl:321-385."""
    Lo, h2 = _bulkscan_call(L.BLMM_NULL_GRID, Y, G, K, Covar, grid_list, addIntercept, weights, prior_variance,
                            prior_sample_size, reml, 1, decomp_scheme, 0, ctx, keep_on_device=keep_on_device, pvals_df=_pvals_df)
    return BulkscanNullResult(Lo, h2)


def bulkscan_alt_grid(Y, G, K, hsq_list, Covar=None, *, reml: bool = False, prior_variance: float = 1.0,
                      prior_sample_size: float = 0.0, weights=None, addIntercept: bool = True, decomp_scheme: str = "eigen",
                      compat_counter_quirk: bool = False, ctx: Optional[Context] = None, keep_on_device: bool = False,
                      _pvals_df=None) -> BulkscanAltResult:
    """src/bulkscan.jl:428-526 (h2_panel = grid value at the arg-max)."""
    Lo, h2 = _bulkscan_call(L.BLMM_ALT_GRID, Y, G, K, Covar, hsq_list, addIntercept, weights, prior_variance,
                            prior_sample_size, reml, 1, decomp_scheme,
                            L.BLMM_COMPAT_ALT_COUNTER if compat_counter_quirk else 0, ctx, keep_on_device=keep_on_device,
                            pvals_df=_pvals_df)
    return BulkscanAltResult(Lo, h2)


def bulkscan_alt_exact(Y, G, K, Covar=None, *, reml: bool = False, prior_variance: float = 0.0, prior_sample_size: float = 0.0,
                       weights=None, addIntercept: bool = True, optim_interval: int = 1, decomp_scheme: str = "eigen",
                       alt_true_weights: bool = False, ctx: Optional[Context] = None) -> dict:
    """The bulk form of `scan(...; assumption = "alt")` (scan_alt, src/scan.jl:397-453; SURVEY.md N3 -- the reference itself only
    has the single-trait function and the grid approximation bulkscan_alt_grid): for every (trait, marker) the exact
    heritability under the alternative, one Brent search per test on the device.  Returns L (p x m), h2_panel (p x m),
    h2_null_list (m), sigma2_e (m); column j equals scan(Y[:, j], ...; assumption = "alt") bit for bit.  Defaults are
    scan's (prior 0 / 0), not bulkscan's."""
    Y = _F(Y); G = _F(G); K = _F(K)
    n, m = Y.shape
    p = G.shape[1]
    if G.shape[0] != n or K.shape[0] != n or K.shape[1] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    _check_n(n)
    if Covar is None and not addIntercept:
        raise BulkLMMError("Intercept has to be added when no other covariate is given.", -7)
    cov, ncov = None, 0
    if Covar is not None:
        cov = _F(Covar)
        if cov.shape[0] != n:
            raise BulkLMMError("Dimension mismatch.", -2)
        ncov = cov.shape[1]
    else:
        addIntercept = True
    w = None if weights is None else np.ascontiguousarray(np.asarray(weights, dtype=np.float64).ravel())
    if w is not None and w.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    o = _opts(L.BLMM_NULL_EXACT, reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    if alt_true_weights:
        o.compat_flags |= L.BLMM_COMPAT_ALT_TRUE_WEIGHTS
    st = L.blmm_status()
    ctx = ctx or default_context()
    Lo = np.empty((p, m), order="F"); H = np.empty((p, m), order="F"); h2 = np.empty(m); s2 = np.empty(m)
    ctx.check(ctx.lib.blmm_bulkscan_alt_exact(ctx.h, C.byref(o), _p(Y), n, m, _p(G), p, _p(cov), ncov, _p(K), _p(w), _p(Lo), _p(H),
                                              _p(h2), _p(s2), C.byref(st)))
    _raise_status(st)
    return {"L": Lo, "h2_panel": H, "h2_null_list": h2, "sigma2_e": s2}


def bulkscan_multi(mctx: MultiContext, Y, G, K, Covar=None, *, method: str = "null-grid", h2_grid=None, gather: str = "host_shards",
                   addIntercept: bool = True, weights=None, prior_variance: float = 1.0, prior_sample_size: float = 0.0,
                   reml: bool = False, optim_interval: int = 1, decomp_scheme: str = "eigen", return_status: bool = False,
                   keep_on_device: bool = False) -> dict:
    """bulkscan over every GPU of `mctx` in ONE call (blmm_bulkscan_multi): the trait blocks the reference deals to its
    threads (src/bulkscan.jl:263-309) go to the devices.  Same result fields as `bulkscan`; `gather` = "host_shards"
    (default), "none" or "allgather" (include/bulklmm_hip.h)."""
    if method not in _METHODS:
        raise BulkLMMError("Unknown method `%s`; choose null-exact, null-grid or alt-grid." % method, -5)
    if gather not in _GATHER:
        raise BulkLMMError("gather must be one of none, host_shards, allgather")
    Y = _F(Y); G = _F(G); K = _F(K)
    n, m = Y.shape
    p = G.shape[1]
    if G.shape[0] != n or K.shape[0] != n or K.shape[1] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    _check_n(n)
    cov, ncov = None, 0
    if Covar is not None:
        cov = _F(Covar)
        if cov.shape[0] != n:
            raise BulkLMMError("Dimension mismatch.", -2)
        ncov = cov.shape[1]
    else:
        addIntercept = True
    w = None if weights is None else np.ascontiguousarray(np.asarray(weights, dtype=np.float64).ravel())
    if w is not None and w.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    meth = _METHODS[method]
    grid, ngrid = None, 0
    if meth != L.BLMM_NULL_EXACT:
        grid = np.ascontiguousarray(np.asarray(h2_grid if h2_grid is not None else [i / 10.0 for i in range(10)], dtype=np.float64))
        ngrid = grid.shape[0]
    o = _opts(meth, reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    mo = L.blmm_multi_opts(_GATHER[gather], 0)
    # keep_on_device: no L_out -- every device keeps its block in HBM; mctx.last_colmax() / last_lod_threshold() reduce them there
    Lout = None if keep_on_device else np.empty((p, m), dtype=np.float64, order="F")
    h2 = np.empty((p, m) if meth == L.BLMM_ALT_GRID else (m,), dtype=np.float64, order="F")
    mctx._last_m = m
    sts = (L.blmm_status * mctx.ndev)()
    mctx.check(mctx.lib.blmm_bulkscan_multi(mctx.h, C.byref(o), C.byref(mo), _p(Y), n, m, _p(G), p, _p(cov), ncov, _p(K), _p(w),
                                            _p(grid), ngrid, _p(Lout), _p(h2), sts))
    for st in sts:
        _raise_status(st)
    out = {"L": Lout, ("h2_panel" if meth == L.BLMM_ALT_GRID else "h2_null_list"): h2}
    if return_status:
        out["status"] = list(sts)
    return out


def lod2log10p(lod, df: int = 1, ctx: Optional[Context] = None):
    """lod2log10p.(lod, df), src/util.jl:199-206, on the GPU (kernels_post.hip: erfc / erfcx for df = 1, ln Q(df/2, .) in
    log space otherwise)."""
    ctx = ctx or default_context()
    a = np.asarray(lod, dtype=np.float64)
    flat = np.asfortranarray(a.reshape(-1, 1) if a.ndim != 2 else a)
    out = np.empty(flat.shape, order="F")
    ctx.check(ctx.lib.blmm_lod2log10p(ctx.h, _p(flat), flat.shape[0], flat.shape[1], int(df), _p(out)))
    return out.reshape(a.shape) if a.ndim != 2 else out


def _last_log10p(ctx: Context, shape, df: int) -> np.ndarray:
    """-log10 p of the LOD matrix the context's last host-pointer call produced (still in HBM: no re-upload)."""
    out = np.empty(shape, order="F")
    ctx.check(ctx.lib.blmm_last_log10p(ctx.h, int(df), _p(out)))
    return out


def lod_threshold(L_mat, thr: float, ctx: Optional[Context] = None, cap: Optional[int] = None):
    """Sparse triplets (marker, trait, LOD) of every LOD > thr, 0-based, sorted by (trait, marker) -- the filter behind
    plot_eQTL(...; threshold) (README.md:354-359), run on the GPU with a device-side count."""
    ctx = ctx or default_context()
    Lm = _F(L_mat)
    p, m = Lm.shape
    cap = int(cap) if cap is not None else max(1024, Lm.size // 64)
    while True:
        ii = np.empty(cap, dtype=np.int32); jj = np.empty(cap, dtype=np.int32); ll = np.empty(cap)
        cnt = C.c_int64(0)
        ctx.check(ctx.lib.blmm_lod_threshold(ctx.h, _p(Lm), p, m, float(thr), cap, ii.ctypes.data_as(C.c_void_p),
                                             jj.ctypes.data_as(C.c_void_p), _p(ll), C.byref(cnt)))
        if cnt.value <= cap:
            break
        cap = int(cnt.value)
    k = int(cnt.value)
    order = np.lexsort((ii[:k], jj[:k]))
    return ii[:k][order], jj[:k][order], ll[:k][order]


def bulkscan(Y, G, K, Covar=None, *, method: str = "null-grid", h2_grid=None, nb: int = 1, nt_blas: int = 1,
             addIntercept: bool = True, weights=None, prior_variance: float = 1.0, prior_sample_size: float = 0.0,
             reml: bool = False, optim_interval: int = 1, decomp_scheme: str = "eigen", output_pvals: bool = False,
             chisq_df: int = 1, ctx: Optional[Context] = None, keep_on_device: bool = False) -> dict:
    """src/bulkscan.jl:81-162.  Returns a dict with the reference NamedTuple's field names.  keep_on_device=True (not in the
    reference): `L` is a DeviceLOD handle -- the p x m matrix stays in HBM and is reduced there (colmax, threshold triplets,
    permutation quantiles, single columns); the call then costs ~2 ms at BXD size instead of ~39 ms, 36 of which are L's trip
    over PCIe."""
    if h2_grid is None:
        h2_grid = [i / 10.0 for i in range(10)]  # collect(0.0:0.1:0.9)
    if method not in _METHODS:
        raise BulkLMMError("Unknown method `%s`; choose null-exact, null-grid or alt-grid." % method, -5)
    # lod2log10p.(L, chisq_df) merged into the result (src/bulkscan.jl:154-157): asked for with the scan, so that the scan
    # kernels write it from their epilogues (chisq_df = 1, null-* methods) into a buffer of the context
    pv = int(chisq_df) if output_pvals else None
    if method == "null-exact":
        r = bulkscan_null(Y, G, K, Covar, addIntercept=addIntercept, weights=weights, prior_variance=prior_variance,
                          prior_sample_size=prior_sample_size, reml=reml, optim_interval=optim_interval,
                          decomp_scheme=decomp_scheme, ctx=ctx, keep_on_device=keep_on_device, _pvals_df=pv)
        out = {"L": r.L, "h2_null_list": r.h2_null_list}
    elif method == "null-grid":
        r = bulkscan_null_grid(Y, G, K, h2_grid, Covar, weights=weights, addIntercept=addIntercept,
                               prior_variance=prior_variance, prior_sample_size=prior_sample_size, reml=reml,
                               decomp_scheme=decomp_scheme, ctx=ctx, keep_on_device=keep_on_device, _pvals_df=pv)
        out = {"L": r.L, "h2_null_list": r.h2_null_list}
    else:
        r = bulkscan_alt_grid(Y, G, K, h2_grid, Covar, reml=reml, prior_variance=prior_variance,
                              prior_sample_size=prior_sample_size, weights=weights, addIntercept=addIntercept,
                              decomp_scheme=decomp_scheme, ctx=ctx, keep_on_device=keep_on_device, _pvals_df=pv)
        out = {"L": r.L, "h2_panel": r.h2_panel}
    if output_pvals:
        out["log10Pvals_mat"] = _last_log10p(ctx or default_context(), out["L"].shape, chisq_df)
        out["Chisq_df"] = chisq_df
    return out


def bulkscan_reduced(Y, G, K, Covar=None, *, method: str = "null-grid", h2_grid=None, threshold: Optional[float] = None,
                     cap: int = 1 << 20, addIntercept: bool = True, weights=None, prior_variance: float = 1.0,
                     prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1, decomp_scheme: str = "eigen",
                     ctx: Optional[Context] = None, return_status: bool = False) -> dict:
    """bulkscan WITHOUT the LOD matrix (blmm_bulkscan_reduced; not in the reference, whose users reduce L on the CPU:
    README.md:246-255, 354-359): per trait the peak LOD and its marker, and -- `threshold` given -- every (marker, trait, LOD) with
    LOD > threshold, computed in the scan kernels' epilogues; L is never written.  Returns {"max_lod": m, "argmax": m (0-based),
    "h2_null_list": m [, "triplets": (i, j, lod) sorted by (trait, marker)], "route": 1 fused | 2 through a resident matrix}.
    `cap`: room for the triplets (16 bytes each, untouched pages cost nothing); more hits than that and the WHOLE call runs again
    with the count it reported -- at the BXD shape, LOD > 5 gives 1e5 triplets, hence the default of 2^20."""
    if h2_grid is None:
        h2_grid = [i / 10.0 for i in range(10)]
    if method not in _METHODS:
        raise BulkLMMError("Unknown method `%s`; choose null-exact, null-grid or alt-grid." % method, -5)
    meth = _METHODS[method]
    Y = _F(Y); G = _F(G); K = _F(K)
    n, m = Y.shape
    p = G.shape[1]
    if G.shape[0] != n or K.shape[0] != n or K.shape[1] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    _check_n(n)
    cov, ncov = None, 0
    if Covar is not None:
        cov = _F(Covar)
        if cov.shape[0] != n:
            raise BulkLMMError("Dimension mismatch.", -2)
        ncov = cov.shape[1]
    else:
        addIntercept = True
    w = None if weights is None else np.ascontiguousarray(np.asarray(weights, dtype=np.float64).ravel())
    if w is not None and w.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    grid, ngrid = None, 0
    if meth != L.BLMM_NULL_EXACT:
        grid = np.ascontiguousarray(np.asarray(h2_grid, dtype=np.float64).ravel())
        ngrid = grid.shape[0]
    o = _opts(meth, reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    ctx = ctx or default_context()
    mx = np.empty(m); arg = np.empty(m, dtype=np.int64); h2 = np.empty(m)
    st = L.blmm_status()
    want = threshold is not None
    while True:
        cnt = C.c_int64(0)
        ii = np.empty(max(cap, 1), dtype=np.int32); jj = np.empty(max(cap, 1), dtype=np.int32); ll = np.empty(max(cap, 1))
        r = L.blmm_reduced(mx.ctypes.data, arg.ctypes.data, 1 if want else 0, float(threshold) if want else 0.0, int(cap) if want else 0,
                           ii.ctypes.data, jj.ctypes.data, ll.ctypes.data, C.addressof(cnt))
        ctx.check(ctx.lib.blmm_bulkscan_reduced(ctx.h, C.byref(o), _p(Y), n, m, _p(G), p, _p(cov), ncov, _p(K), _p(w), _p(grid), ngrid,
                                                C.byref(r), _p(h2), C.byref(st)))
        if not want or cnt.value <= cap:
            break
        cap = int(cnt.value)
    _raise_status(st)
    out = {"max_lod": mx, "argmax": arg, "route": int(ctx.lib.blmm_last_reduced_route(ctx.h))}
    if meth != L.BLMM_ALT_GRID:
        out["h2_null_list"] = h2
    if want:
        k = int(cnt.value)
        order = np.lexsort((ii[:k], jj[:k]))
        out["triplets"] = (ii[:k][order], jj[:k][order], ll[:k][order])
    if return_status:
        out["status"] = st
    return out


def scan(y, g, K, covar=None, *, weights=None, prior_variance: float = 0.0, prior_sample_size: float = 0.0,
         addIntercept: bool = True, reml: bool = False, assumption: str = "null", method: str = "qr", optim_interval: int = 1,
         permutation_test: bool = False, nperms: int = 1024, rndseed: int = 0, profileLL: bool = False, markerID: int = 0,
         h2_grid=None, decomp_scheme: str = "eigen",
         output_pvals: bool = False, chisq_df: int = 1, perm_idx=None, perm_precision: str = "f64",
         alt_true_weights: bool = False, ctx: Optional[Context] = None) -> dict:
    """src/scan.jl:94-271: the single-trait scan routed through the same GPU kernels (Brent + exact-weights LOD kernel with
    m = 1), the permutation test (src/scan.jl:485-557), and assumption == "alt" (scan_alt, src/scan.jl:397-453: one Brent
    search per marker on the device; adds `h2_each_marker`; `alt_true_weights` see BLMM_COMPAT_ALT_TRUE_WEIGHTS).
    `perm_idx` (n x nperms, 0-based) supplies the permutations; otherwise the library draws them from
    `rndseed` with its own generator (Julia's MersenneTwister stream is not reproducible).
    `perm_precision="f32"` (not in the reference; BASELINE.json configs[4]) computes L_perms on the fp32 matrix cores
    and returns it as float32; the null model and `lod` stay fp64."""
    y = _F(y)
    if covar is None and not addIntercept:
        raise BulkLMMError("Intercept has to be added when no other covariate is given.", -7)  # src/scan.jl:167-169
    if profileLL:   # src/scan.jl:252-267 (profile_LL of src/analysis_helpers): not part of the GPU path
        raise NotImplementedError("profileLL = true (profile_LL) is outside the GPU hot path; `markerID` / `h2_grid` only matter there")
    # `method` ("qr" / "cholesky") picks a CPU factorisation in the reference; the GPU path has one (closed-form WLS)
    if assumption == "alt" and permutation_test:
        raise BulkLMMError("Permutation test option currently is not supported for the alternative assumption.")
    if assumption not in ("null", "alt"):
        raise BulkLMMError("Assumption keyword is not supported. Please enter null or alt.")
    if y.shape[1] != 1:
        raise BulkLMMError("Can only handle one trait.", -6)  # src/scan.jl:496-498
    G = _F(g)
    K = _F(K)
    n = y.shape[0]
    p = G.shape[1]
    if G.shape[0] != n or K.shape[0] != n or K.shape[1] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    _check_n(n)
    cov = None
    ncov = 0
    if covar is not None:
        cov = _F(covar)
        if cov.shape[0] != n:
            raise BulkLMMError("Dimension mismatch.", -2)
        ncov = cov.shape[1]
    else:
        addIntercept = True
    w = None if weights is None else np.ascontiguousarray(np.asarray(weights, dtype=np.float64).ravel())
    if w is not None and w.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    o = _opts(L.BLMM_NULL_EXACT, reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    st = L.blmm_status()
    if not permutation_test:
        nperms = 0
    if nperms < 0:
        raise BulkLMMError("The required number of permutations must be a positive integer.", -9)
    pidx = None
    if perm_idx is not None and nperms > 0:
        pidx = np.asfortranarray(np.asarray(perm_idx, dtype=np.int32))
        if pidx.shape != (n, nperms):
            raise BulkLMMError("Dimension mismatch.", -2)
    if perm_precision not in ("f64", "f32"):
        raise BulkLMMError("perm_precision must be \"f64\" or \"f32\".")
    ctx = ctx or default_context()
    scal = np.zeros(2)
    lod = np.empty(p)
    if assumption == "alt":
        if alt_true_weights:
            o.compat_flags |= L.BLMM_COMPAT_ALT_TRUE_WEIGHTS
        h2e = np.empty(p)
        ctx.check(ctx.lib.blmm_scan_alt(ctx.h, C.byref(o), _p(y), n, _p(G), p, _p(cov), ncov, _p(K), _p(w), _p(scal), _p(lod),
                                        _p(h2e), C.byref(st)))
        _raise_status(st)
        out = {"sigma2_e": float(scal[0]), "h2_null": float(scal[1]), "h2_each_marker": h2e, "lod": lod}
        if output_pvals:
            out["log10pvals"] = lod2log10p(lod, chisq_df, ctx=ctx)
        return out
    f32 = perm_precision == "f32"
    Lp = np.empty((p, max(nperms, 1)), order="F", dtype=np.float32 if f32 else np.float64)
    fn = ctx.lib.blmm_scan_perms_f32 if f32 else ctx.lib.blmm_scan_perms
    ctx.check(fn(ctx.h, C.byref(o), _p(y), n, _p(G), p, _p(cov), ncov, _p(K), _p(w), nperms,
                                      C.c_uint64(int(rndseed)), _p(pidx), _p(scal), _p(lod), _p(Lp), C.byref(st)))
    _raise_status(st)
    out = {"sigma2_e": float(scal[0]), "h2_null": float(scal[1]), "lod": lod}
    if permutation_test:
        out["L_perms"] = Lp[:, :nperms]
    if output_pvals:
        out["log10pvals"] = lod2log10p(lod, chisq_df, ctx=ctx)
        if permutation_test and not f32 and nperms > 0:
            out["log10Pvals_perms"] = _last_log10p(ctx, (p, nperms), chisq_df)  # the reference's UndefVarError fixed (B3)
    return out


def lod_colmax(L_mat, ctx: Optional[Context] = None):
    """Per-column maximum of an LOD matrix and the (0-based) marker where it sits, computed on the GPU."""
    ctx = ctx or default_context()
    Lm = _F(L_mat)
    p, m = Lm.shape
    mx = np.empty(m)
    arg = np.empty(m, dtype=np.int64)
    ctx.check(ctx.lib.blmm_lod_colmax(ctx.h, _p(Lm), p, m, _p(mx), arg.ctypes.data_as(C.c_void_p)))
    return mx, arg


def get_thresholds(L_perms, signif_level, ctx: Optional[Context] = None):
    """src/analysis_helpers/single_trait_analysis.jl:13-23: quantiles of the per-permutation peak LODs; column maxima,
    sort and Julia's default (linear interpolation) quantile all on the GPU (blmm_get_thresholds)."""
    ctx = ctx or default_context()
    Lm = _F(L_perms)
    p, nperms = Lm.shape
    thr_probs = np.ascontiguousarray(1.0 - np.atleast_1d(np.asarray(signif_level, dtype=np.float64)))
    thrs = np.empty(thr_probs.shape[0])
    ctx.check(ctx.lib.blmm_get_thresholds(ctx.h, _p(Lm), p, nperms, _p(thr_probs), thr_probs.shape[0], _p(thrs)))
    return {"probs": thr_probs, "thrs": thrs}


# ---- lower-level seams ------------------------------------------------------------------------------

def transform_rotation(y, g, K, *, addIntercept: bool = True, decomp_scheme: str = "eigen", ctx: Optional[Context] = None):
    """src/transform_helpers.jl:1-54: (Ut*y, Ut*[1 g], lambda).  Eigenvector signs/order within equal
    eigenvalues are arbitrary, exactly as with LAPACK."""
    y = _F(y)
    g = _F(g)
    K = _F(K)
    n, m = y.shape
    if g.shape[0] != n or K.shape[0] != n:
        raise BulkLMMError("Dimension mismatch.", -2)
    _check_n(n)
    ctx = ctx or default_context()  # after the argument checks: those must not need a GPU
    o = _opts(decomp_scheme=decomp_scheme, addIntercept=addIntercept)
    if addIntercept:
        cov, ncov, G, c = None, 0, g, 1
    else:
        # the first column of g plays the covariate role only for the layout of X0; rotation is column-wise
        cov, ncov, G, c = np.asfortranarray(g[:, :1]), 1, np.asfortranarray(g[:, 1:]), 1
        if G.shape[1] == 0:
            raise BulkLMMError("Dimension mismatch.", -2)
    p = G.shape[1]
    Y0 = np.empty((n, m), order="F")
    X0 = np.empty((n, c + p), order="F")
    lam = np.empty(n)
    st = L.blmm_status()
    ctx.check(ctx.lib.blmm_rotate(ctx.h, C.byref(o), _p(y), n, m, _p(G), p, _p(cov), ncov, _p(K), _p(Y0), _p(X0), _p(lam), C.byref(st)))
    _raise_status(st)
    return Y0, X0, lam


def fitlmm_bulk(Y0, Z0, lambda0, prior=(0.0, 0.0), *, reml: bool = False, optim_interval: int = 1, ctx: Optional[Context] = None):
    """fitlmm (src/lmm.jl:56-86) for every column of Y0: returns (h2, sigma2, ell), m each."""
    ctx = ctx or default_context()
    Y0 = _F(Y0)
    Z0 = _F(Z0)
    lam = np.ascontiguousarray(np.asarray(lambda0, dtype=np.float64))
    n, m = Y0.shape
    o = _opts(reml=reml, optim_interval=optim_interval, prior_variance=prior[0], prior_sample_size=prior[1])
    h2, s2, ell = np.empty(m), np.empty(m), np.empty(m)
    st = L.blmm_status()
    ctx.check(ctx.lib.blmm_null_h2_brent(ctx.h, C.byref(o), _p(Y0), n, m, _p(Z0), Z0.shape[1], _p(lam), _p(h2), _p(s2), _p(ell), C.byref(st)))
    return h2, s2, ell


def null_loglik_grid(Y0, Z0, lambda0, h2_grid, prior=(1.0, 0.0), *, reml: bool = False, ctx: Optional[Context] = None):
    """wls_multivar(...).Ell over a grid (src/bulkscan_helpers.jl:267-269): ngrid x m."""
    ctx = ctx or default_context()
    Y0 = _F(Y0)
    Z0 = _F(Z0)
    lam = np.ascontiguousarray(np.asarray(lambda0, dtype=np.float64))
    grid = np.ascontiguousarray(np.asarray(h2_grid, dtype=np.float64))
    n, m = Y0.shape
    o = _opts(reml=reml, prior_variance=prior[0], prior_sample_size=prior[1])
    Ell = np.empty((grid.shape[0], m), order="F")
    st = L.blmm_status()
    ctx.check(ctx.lib.blmm_null_loglik_grid(ctx.h, C.byref(o), _p(Y0), n, m, _p(Z0), Z0.shape[1], _p(lam), _p(grid), grid.shape[0], _p(Ell), C.byref(st)))
    return Ell


def weighted_liteqtl(Y0, X0, lambda0, hsq: float, *, num_of_covar: int = 1, ctx: Optional[Context] = None):
    """src/bulkscan_helpers.jl:175-201."""
    ctx = ctx or default_context()
    Y0 = _F(Y0)
    X0 = _F(X0)
    lam = np.ascontiguousarray(np.asarray(lambda0, dtype=np.float64))
    n, m = Y0.shape
    p = X0.shape[1] - num_of_covar
    out = np.empty((p, m), order="F")
    st = L.blmm_status()
    ctx.check(ctx.lib.blmm_weighted_liteqtl(ctx.h, _p(Y0), n, m, _p(X0), num_of_covar, p, _p(lam), float(hsq), _p(out), C.byref(st)))
    _raise_status(st)
    return out


def liteqtl_given_h2(Y0, X0, lambda0, h2, *, num_of_covar: int = 1, ctx: Optional[Context] = None):
    """univar_liteqtl's scan part (src/bulkscan_helpers.jl:138-146) for every column of Y0 with per-trait h2."""
    ctx = ctx or default_context()
    Y0 = _F(Y0)
    X0 = _F(X0)
    lam = np.ascontiguousarray(np.asarray(lambda0, dtype=np.float64))
    h2 = np.ascontiguousarray(np.asarray(h2, dtype=np.float64))
    n, m = Y0.shape
    p = X0.shape[1] - num_of_covar
    out = np.empty((p, m), order="F")
    st = L.blmm_status()
    ctx.check(ctx.lib.blmm_liteqtl_given_h2(ctx.h, _p(Y0), n, m, _p(X0), num_of_covar, p, _p(lam), _p(h2), _p(out), C.byref(st)))
    _raise_status(st)
    return out


# ---- device-resident entry points (torch tensors; used by bench.py and the multi-GPU path) -------------

def bulkscan_dev(ctx: Context, Y, G, K, L_out, h2_out, *, method: str = "null-exact", h2_grid=None, Covar=None, weights=None,
                 addIntercept: bool = True, prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False,
                 optim_interval: int = 1, decomp_scheme: str = "eigen", status: bool = False, log10p_out=None, chisq_df: int = 1):
    """blmm_bulkscan_dev on torch CUDA tensors laid out column-major: pass Y as a (m, n) contiguous tensor
    (= n x m column-major), G as (p, n), K as (n, n), L_out as (m, p) (= p x m column-major; rows may be padded: the
    leading dimension passed on is L_out.stride(0)).  `log10p_out` (same layout as L_out): `output_pvals` inside the scan
    (blmm_set_log10p_output).
    Enqueues on the context's stream and does not synchronise unless `status` is requested."""
    m, n = Y.shape
    p = G.shape[0]
    grid = None
    ngrid = 0
    if method != "null-exact":
        grid = np.ascontiguousarray(np.asarray(h2_grid if h2_grid is not None else [i / 10.0 for i in range(10)], dtype=np.float64))
        ngrid = grid.shape[0]
    ncov = 0 if Covar is None else Covar.shape[0]
    if Covar is None:
        addIntercept = True
    o = _opts(_METHODS[method], reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    st = L.blmm_status() if status else None
    # armed right in front of the call (after everything above that can raise); the library consumes the request first thing
    if log10p_out is not None:
        ctx.check(ctx.lib.blmm_set_log10p_output(ctx.h, log10p_out.data_ptr(), _ld(log10p_out, p), int(chisq_df)))
    try:
        ctx.check(ctx.lib.blmm_bulkscan_dev(ctx.h, C.byref(o), Y.data_ptr(), n, m, G.data_ptr(), p,
                                            None if Covar is None else Covar.data_ptr(), ncov, K.data_ptr(),
                                            None if weights is None else weights.data_ptr(), _p(grid), ngrid,
                                            L_out.data_ptr(), _ld(L_out, p), h2_out.data_ptr(), C.byref(st) if status else None))
    finally:
        if log10p_out is not None:
            ctx.lib.blmm_set_log10p_output(ctx.h, None, 0, 0)
    return st


def bulkscan_reduced_dev(ctx: Context, Y, G, K, max_out, argmax_out, h2_out, *, method: str = "null-exact", h2_grid=None, Covar=None,
                         weights=None, addIntercept: bool = True, prior_variance: float = 1.0, prior_sample_size: float = 0.0,
                         reml: bool = False, optim_interval: int = 1, decomp_scheme: str = "eigen", threshold: Optional[float] = None,
                         trip_i=None, trip_j=None, trip_lod=None, trip_count=None, status: bool = False):
    """blmm_bulkscan_reduced_dev on torch CUDA tensors (layouts as bulkscan_dev): max_out (m, float64), argmax_out (m, int64);
    threshold given: trip_i / trip_j (int32, cap), trip_lod (float64, cap), trip_count (int64, 1).  Synchronises the stream."""
    m, n = Y.shape
    p = G.shape[0]
    grid, ngrid = None, 0
    if method != "null-exact":
        grid = np.ascontiguousarray(np.asarray(h2_grid if h2_grid is not None else [i / 10.0 for i in range(10)], dtype=np.float64))
        ngrid = grid.shape[0]
    ncov = 0 if Covar is None else Covar.shape[0]
    if Covar is None:
        addIntercept = True
    o = _opts(_METHODS[method], reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    st = L.blmm_status() if status else None
    want = threshold is not None
    r = L.blmm_reduced(None if max_out is None else max_out.data_ptr(), None if argmax_out is None else argmax_out.data_ptr(),
                       1 if want else 0, float(threshold) if want else 0.0, int(trip_i.numel()) if want else 0,
                       trip_i.data_ptr() if want else None, trip_j.data_ptr() if want else None,
                       trip_lod.data_ptr() if want else None, trip_count.data_ptr() if want else None)
    ctx.check(ctx.lib.blmm_bulkscan_reduced_dev(ctx.h, C.byref(o), Y.data_ptr(), n, m, G.data_ptr(), p,
                                                None if Covar is None else Covar.data_ptr(), ncov, K.data_ptr(),
                                                None if weights is None else weights.data_ptr(), _p(grid), ngrid, C.byref(r),
                                                None if h2_out is None else h2_out.data_ptr(), C.byref(st) if status else None))
    return st


def prepare_dev(ctx: Context, K, *, Covar=None, weights=None, addIntercept: bool = True, decomp_scheme: str = "eigen", status: bool = False):
    """blmm_prepare_dev on torch CUDA tensors (K (n, n); Covar (ncov, n) = n x ncov column-major): design, eigen-decomposition and
    rotation matrix; the context then serves rotate_block_dev / bulkscan_prerotated_dev (one process per GPU: the marker
    rotation is sharded over the ranks, include/bulklmm_hip.h)."""
    n = K.shape[0]
    ncov = 0 if Covar is None else Covar.shape[0]
    if Covar is None:
        addIntercept = True
    o = _opts(L.BLMM_NULL_EXACT, False, addIntercept, decomp_scheme)
    st = L.blmm_status() if status else None
    ctx.check(ctx.lib.blmm_prepare_dev(ctx.h, C.byref(o), n, None if Covar is None else Covar.data_ptr(), ncov, K.data_ptr(),
                                       None if weights is None else weights.data_ptr(), C.byref(st) if status else None))
    return st


def rotated_rows(ctx: Context) -> int:
    return int(ctx.lib.blmm_rotated_rows(ctx.h))


def rotate_block_dev(ctx: Context, G_block, Xt_block):
    """blmm_rotate_block_dev: G_block (pb, n) contiguous (= n x pb column-major) -> Xt_block (rows, ld) contiguous, k-major."""
    pb = G_block.shape[0]
    ctx.check(ctx.lib.blmm_rotate_block_dev(ctx.h, G_block.data_ptr(), pb, Xt_block.data_ptr(), Xt_block.stride(0)))


def bulkscan_prerotated_dev(ctx: Context, Y, Xt_blocks, p: int, block_cols: int, L_out, h2_out, *, method: str = "null-exact", h2_grid=None,
                            prior_variance: float = 1.0, prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1,
                            status: bool = False):
    """blmm_bulkscan_prerotated_dev: Y (m, n); Xt_blocks (nblocks, rows, block_ld) contiguous -- the all-gathered output of
    rotate_block_dev, block b = markers [b block_cols, min(p, (b+1) block_cols)); L_out (m, p) [ld = stride(0)]."""
    m = Y.shape[0]
    grid, ngrid = None, 0
    if method != "null-exact":
        grid = np.ascontiguousarray(np.asarray(h2_grid if h2_grid is not None else [i / 10.0 for i in range(10)], dtype=np.float64))
        ngrid = grid.shape[0]
    o = _opts(_METHODS[method], reml, True, "eigen", optim_interval, prior_variance, prior_sample_size)
    st = L.blmm_status() if status else None
    nb, rows, bld = Xt_blocks.shape
    assert Xt_blocks.is_contiguous() and rows == rotated_rows(ctx)
    ctx.check(ctx.lib.blmm_bulkscan_prerotated_dev(ctx.h, C.byref(o), Y.data_ptr(), m, int(p), Xt_blocks.data_ptr(), nb, int(block_cols), bld,
                                                   _p(grid), ngrid, L_out.data_ptr(), _ld(L_out, int(p)), h2_out.data_ptr(),
                                                   C.byref(st) if status else None))
    return st


def _ld(t, p):
    """Leading dimension of a (cols, p) tensor that holds a p x cols column-major matrix."""
    if t.dim() != 2 or t.shape[1] != p or (p > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < p):
        raise ValueError("L_out must be a (m, p) tensor with unit stride along p")
    return t.stride(0) if t.shape[0] > 1 else max(p, t.stride(0))


def scan_perms_prerotated_dev(ctx: Context, y, Xt_blocks, p: int, block_cols: int, scalars_out, lod_out, Lperms_out, *, nperms: int,
                              seed: int = 0, perm_idx=None, prior_variance: float = 0.0, prior_sample_size: float = 0.0,
                              reml: bool = False, optim_interval: int = 1, status: bool = False):
    """blmm_scan_perms_prerotated_dev (after prepare_dev on this context): the permutation test on the gathered rotated marker
    blocks Xt_blocks (nblocks, rows, block_ld); y (n,), scalars_out (2,), lod_out (p,), Lperms_out (nperms, p) float64 or float32
    (fp32 matrix cores), perm_idx (nperms, n) int32 or None (the library's generator with `seed`)."""
    import torch
    o = _opts(L.BLMM_NULL_EXACT, reml, True, "eigen", optim_interval, prior_variance, prior_sample_size)
    st = L.blmm_status() if status else None
    nb, rows, bld = Xt_blocks.shape
    assert Xt_blocks.is_contiguous() and rows == rotated_rows(ctx)
    f32 = Lperms_out is not None and Lperms_out.dtype == torch.float32
    ctx.check(ctx.lib.blmm_scan_perms_prerotated_dev(ctx.h, C.byref(o), y.data_ptr(), int(p), Xt_blocks.data_ptr(), nb, int(block_cols), bld,
                                                     int(nperms), C.c_uint64(int(seed)), None if perm_idx is None else perm_idx.data_ptr(),
                                                     scalars_out.data_ptr(), lod_out.data_ptr(),
                                                     None if (f32 or Lperms_out is None) else Lperms_out.data_ptr(),
                                                     Lperms_out.data_ptr() if f32 else None, C.byref(st) if status else None))
    return st


def scan_perms_dev(ctx: Context, y, G, K, scalars_out, lod_out, Lperms_out, *, nperms: int, seed: int = 0, perm_idx=None,
                   Covar=None, weights=None, addIntercept: bool = True, prior_variance: float = 0.0,
                   prior_sample_size: float = 0.0, reml: bool = False, optim_interval: int = 1,
                   decomp_scheme: str = "eigen", status: bool = False):
    """blmm_scan_perms_dev on torch CUDA tensors: y (n,), G (p, n) [= n x p column-major], K (n, n),
    scalars_out (2,), lod_out (p,), Lperms_out (nperms, p) [= p x nperms column-major], perm_idx (nperms, n) int32.
    A float32 Lperms_out selects the fp32 permutation kernel (blmm_scan_perms_f32_dev)."""
    n = y.shape[0]
    p = G.shape[0]
    ncov = 0 if Covar is None else Covar.shape[0]
    if Covar is None:
        addIntercept = True
    o = _opts(L.BLMM_NULL_EXACT, reml, addIntercept, decomp_scheme, optim_interval, prior_variance, prior_sample_size)
    st = L.blmm_status() if status else None
    import torch
    f32 = Lperms_out is not None and Lperms_out.dtype == torch.float32
    fn = ctx.lib.blmm_scan_perms_f32_dev if f32 else ctx.lib.blmm_scan_perms_dev
    ctx.check(fn(ctx.h, C.byref(o), y.data_ptr(), n, G.data_ptr(), p,
                                          None if Covar is None else Covar.data_ptr(), ncov, K.data_ptr(),
                                          None if weights is None else weights.data_ptr(), int(nperms), C.c_uint64(int(seed)),
                                          None if perm_idx is None else perm_idx.data_ptr(), scalars_out.data_ptr(),
                                          lod_out.data_ptr(), None if Lperms_out is None else Lperms_out.data_ptr(),
                                          C.byref(st) if status else None))
    return st
