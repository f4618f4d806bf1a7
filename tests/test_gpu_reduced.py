"""The drop-in call without the 2 GB trip over PCIe ("the matrix never has to leave HBM"):

  * `bulkscan(...; keep_on_device=True)` = blmm_bulkscan with L_out == NULL: L stays in the context's workspace and the blmm_last_*
    consumers -- what README.md:246-255, 354-359 and get_thresholds (src/analysis_helpers/single_trait_analysis.jl:13-23) do with
    L -- serve it;
  * `bulkscan_reduced` = blmm_bulkscan_reduced: the scan kernels reduce in their epilogues and L is never WRITTEN; per-trait
    (max, argmax) and LOD > t triplets must be bit-identical to blmm_lod_colmax / blmm_lod_threshold on the stored matrix;
  * the one-shot / stale-state fixes of the round-3 advisor (blmm_prepare_dev state, blmm_set_log10p_output request) and the
    tuning keys that replaced the numerics-changing BLMM_* environment variables."""
import ctypes as C

import numpy as np
import pytest

from common import DevBuf, make_data

pytestmark = pytest.mark.gpu


def colmax_ref(L):
    """k_colmax's rule on the host: strictly larger wins, so the lowest marker among equal maxima; NaN never."""
    Lm = np.where(np.isnan(L), -np.inf, L)
    arg = np.argmax(Lm, axis=0)                 # first occurrence of the maximum
    return Lm[arg, np.arange(L.shape[1])], arg


def triplets_ref(L, thr):
    i, j = np.nonzero(L > thr)
    order = np.lexsort((i, j))
    return i[order].astype(np.int32), j[order].astype(np.int32), L[i[order], j[order]]


@pytest.mark.parametrize("method,ncov,m", [("null-exact", 0, 700), ("null-exact", 2, 300), ("null-grid", 1, 500), ("alt-grid", 0, 130),
                                            ("null-exact", 5, 90)])
def test_reduced_scan_equals_the_reductions_of_the_stored_matrix(blmm, method, ncov, m):
    """Fused routes (null-grid; null-exact with c <= 3: k_scan<table,perm> + k_scan_lr3 / k_scan_lr) and the routes through a
    resident matrix (alt-grid, c = 6): maxima, arg-maxima, triplets and h2 against the stored matrix of the ordinary call --
    bit for bit.  Ragged sizes: p and m are not multiples of the 128 x 64 tile."""
    Y, G, K, Cov = make_data(p=1013, m=m, seed=3300 + ncov + m, ncov=ncov)
    G = G.copy(); G[:, 700] = G[:, 3]                       # a duplicated marker: equal LODs -> the arg-max tie rule matters
    ctx = blmm.Context(0)
    full = blmm.bulkscan(Y, G, K, Cov, method=method, ctx=ctx)
    L = full["L"]
    mx, arg = colmax_ref(L)
    thr = float(np.quantile(L, 0.999))
    red = blmm.bulkscan_reduced(Y, G, K, Cov, method=method, threshold=thr, cap=64, ctx=ctx)     # cap too small: the wrapper retries
    assert red["route"] == (1 if (method == "null-grid" or (method == "null-exact" and ncov <= 2)) else 2)
    assert np.array_equal(red["max_lod"], mx) and np.array_equal(red["argmax"], arg)
    ti, tj, tl = triplets_ref(L, thr)
    gi, gj, gl = red["triplets"]
    assert gi.size == ti.size > 64 and np.array_equal(gi, ti) and np.array_equal(gj, tj) and np.array_equal(gl, tl)
    if method != "alt-grid":
        assert np.array_equal(red["h2_null_list"], full["h2_null_list"])
    # no triplets wanted: maxima only
    red2 = blmm.bulkscan_reduced(Y, G, K, Cov, method=method, ctx=ctx)
    assert "triplets" not in red2 and np.array_equal(red2["max_lod"], mx) and np.array_equal(red2["argmax"], arg)
    ctx.close()


def test_reduced_scan_reruns_through_a_resident_matrix_when_a_trait_needs_a_rescan(blmm):
    """The fused route speculates that no trait is flagged by the guards; with lr_tol = 0 every trait is (k_scan_fix would patch L,
    which does not exist): the call must notice and deliver the re-scanned values -- those of the ordinary call under the same
    tuning."""
    Y, G, K, _ = make_data(p=500, m=200, seed=3401)
    ctx = blmm.Context(0)
    ctx.set_tuning("lr_tol", 0.0)
    full = blmm.bulkscan(Y, G, K, method="null-exact", ctx=ctx)
    red = blmm.bulkscan_reduced(Y, G, K, method="null-exact", threshold=3.0, ctx=ctx, return_status=True)
    assert red["route"] == 2 and red["status"].lowrank_fallback == 200
    mx, arg = colmax_ref(full["L"])
    assert np.array_equal(red["max_lod"], mx) and np.array_equal(red["argmax"], arg)
    ti, tj, tl = triplets_ref(full["L"], 3.0)
    assert np.array_equal(red["triplets"][0], ti) and np.array_equal(red["triplets"][2], tl)
    ctx.set_tuning("defaults", 0)
    red1 = blmm.bulkscan_reduced(Y, G, K, method="null-exact", ctx=ctx)
    assert red1["route"] == 1
    ctx.close()


def test_reduced_scan_dev_entry_point_and_empty_shapes(blmm):
    L_ = blmm._lib
    Y, G, K, _ = make_data(p=300, m=150, seed=3402)
    n, m, p = 79, 150, 300
    ctx = blmm.Context(0)
    lib = ctx.lib
    o = blmm.api._opts(L_.BLMM_NULL_EXACT)
    dY, dG, dK = DevBuf(np.asfortranarray(Y).ravel("F")), DevBuf(np.asfortranarray(G).ravel("F")), DevBuf(np.asfortranarray(K).ravel("F"))
    dL, dh = DevBuf(nbytes=8 * p * m), DevBuf(nbytes=8 * m)
    ctx.check(lib.blmm_bulkscan_dev(ctx.h, C.byref(o), dY.ptr, n, m, dG.ptr, p, None, 0, dK.ptr, None, None, 0, dL.ptr, p, dh.ptr, None))
    ctx.synchronize()
    L = dL.get((m, p)).T
    dmx, dax, dh2 = DevBuf(nbytes=8 * m), DevBuf(nbytes=8 * m), DevBuf(nbytes=8 * m)
    cap = 4096
    dti, dtj, dtl, dtc = DevBuf(nbytes=4 * cap), DevBuf(nbytes=4 * cap), DevBuf(nbytes=8 * cap), DevBuf(np.zeros(1, dtype=np.int64))
    r = L_.blmm_reduced(dmx.ptr, dax.ptr, 1, 2.5, cap, dti.ptr, dtj.ptr, dtl.ptr, dtc.ptr)
    ctx.check(lib.blmm_bulkscan_reduced_dev(ctx.h, C.byref(o), dY.ptr, n, m, dG.ptr, p, None, 0, dK.ptr, None, None, 0, C.byref(r), dh2.ptr, None))
    assert lib.blmm_last_reduced_route(ctx.h) == 1
    rmx, rarg = colmax_ref(L)
    assert np.array_equal(dmx.get(m), rmx) and np.array_equal(dax.get(m, np.int64), rarg)
    assert np.array_equal(dh2.get(m), dh.get(m))
    k = int(dtc.get(1, np.int64)[0])
    ri, rj, rl = triplets_ref(L, 2.5)
    gi, gj, gl = dti.get(cap, np.int32)[:k], dtj.get(cap, np.int32)[:k], dtl.get(cap)[:k]
    order = np.lexsort((gi, gj))
    assert k == ri.size and np.array_equal(gi[order], ri) and np.array_equal(gj[order], rj) and np.array_equal(gl[order], rl)
    # cap smaller than the count: the count is still the total, only `cap` triplets are stored (all of them genuine)
    r2 = L_.blmm_reduced(None, None, 1, 2.5, 10, dti.ptr, dtj.ptr, dtl.ptr, dtc.ptr)
    dtl.fill(np.full(cap, -1.0))
    ctx.check(lib.blmm_bulkscan_reduced_dev(ctx.h, C.byref(o), dY.ptr, n, m, dG.ptr, p, None, 0, dK.ptr, None, None, 0, C.byref(r2), dh2.ptr, None))
    assert int(dtc.get(1, np.int64)[0]) == k and k > 10
    g10 = dtl.get(cap)
    assert np.all(g10[10:] == -1.0) and np.all(g10[:10] == L[dti.get(cap, np.int32)[:10], dtj.get(cap, np.int32)[:10]])
    for b in (dY, dG, dK, dL, dh, dmx, dax, dh2, dti, dtj, dtl, dtc):
        b.free()
    # no markers / no traits: -inf and -1, nothing crashes
    e = blmm.bulkscan_reduced(Y, G[:, :0], K, method="null-exact", ctx=ctx)
    assert np.all(np.isneginf(e["max_lod"])) and np.all(e["argmax"] == -1) and e["h2_null_list"].shape == (150,)
    e2 = blmm.bulkscan_reduced(Y[:, :0], G, K, method="null-grid", threshold=1.0, ctx=ctx)
    assert e2["max_lod"].shape == (0,) and e2["triplets"][0].size == 0
    ctx.close()


@pytest.mark.parametrize("method", ["null-exact", "null-grid", "alt-grid"])
def test_keep_on_device_serves_the_resident_matrix(blmm, method):
    Y, G, K, _ = make_data(p=777, m=333, seed=3500)
    ctx = blmm.Context(0)
    full = blmm.bulkscan(Y, G, K, method=method, ctx=ctx)
    kept = blmm.bulkscan(Y, G, K, method=method, ctx=ctx, keep_on_device=True)
    d = kept["L"]
    assert isinstance(d, blmm.DeviceLOD) and d.shape == (777, 333)
    hk = "h2_panel" if method == "alt-grid" else "h2_null_list"
    assert np.array_equal(kept[hk], full[hk])
    mx, arg = colmax_ref(full["L"])
    gmx, garg = d.colmax()
    assert np.array_equal(gmx, mx) and np.array_equal(garg, arg)
    ti, tj, tl = triplets_ref(full["L"], 3.0)
    gi, gj, gl = d.threshold(3.0, cap=16)
    assert np.array_equal(gi, ti) and np.array_equal(gj, tj) and np.array_equal(gl, tl)
    cols = [0, 332, 17, 17, 100]
    assert np.array_equal(d.columns(cols), full["L"][:, cols])
    assert np.array_equal(d.to_host(), full["L"])
    q = d.get_thresholds([0.5, 0.95])
    assert np.allclose(q, np.quantile(mx, [0.5, 0.95]), rtol=1e-14, atol=0)
    assert np.allclose(d.log10p(1), blmm.lod2log10p(full["L"], 1), rtol=1e-14, atol=1e-300)
    with pytest.raises(blmm.BulkLMMError):
        d.columns([333])
    blmm.bulkscan(Y[:, :10], G, K, method=method, ctx=ctx)           # another matrix takes its place: the handle says so
    with pytest.raises(blmm.BulkLMMError):
        d.colmax()
    ctx.close()


def test_multi_gpu_entry_point_keeps_the_blocks_on_the_devices(blmm):
    Y, G, K, _ = make_data(p=300, m=203, seed=3600)
    full = blmm.bulkscan(Y, G, K, method="null-exact")
    mx, arg = colmax_ref(full["L"])
    ti, tj, tl = triplets_ref(full["L"], 2.0)
    mc = blmm.MultiContext([0, 0, 0])
    for gather, keep in (("host_shards", True), ("none", False), ("allgather", False)):
        r = blmm.bulkscan_multi(mc, Y, G, K, method="null-exact", gather=gather, keep_on_device=keep)
        if keep:
            assert r["L"] is None
        assert np.array_equal(r["h2_null_list"], full["h2_null_list"])
        gmx, garg = mc.last_colmax()
        assert np.array_equal(gmx, mx) and np.array_equal(garg, arg), gather
        gi, gj, gl = mc.last_lod_threshold(2.0, cap=8)
        assert np.array_equal(gi, ti) and np.array_equal(gj, tj) and np.array_equal(gl, tl), gather
    mc.close()


# ---- round-3 advisor: stale three-call state, one-shot p-value request ------------------------------------------------------------
def test_prepare_state_is_invalidated_by_any_other_call_that_redoes_the_eigen_front(blmm):
    """blmm_prepare_dev leaves U / lambda / Z0 / the rotation matrix in the context for blmm_rotate_block_dev and the
    *_prerotated calls.  Any other entry point overwrites (and at a larger n reallocates) them: the three-call state must then be
    refused, not used."""
    ctx = blmm.Context(0)
    lib = ctx.lib
    Ya, Ga, Ka, _ = make_data(n=150, p=64, m=8, seed=3700, bxd=False)
    Yb, Gb, Kb, _ = make_data(n=300, p=64, m=8, seed=3701, bxd=False)
    o = blmm.api._opts(blmm._lib.BLMM_NULL_EXACT)
    dKa, dGa, dYa = DevBuf(np.asfortranarray(Ka).ravel("F")), DevBuf(np.asfortranarray(Ga).ravel("F")), DevBuf(np.asfortranarray(Ya).ravel("F"))
    ctx.check(lib.blmm_prepare_dev(ctx.h, C.byref(o), 150, None, 0, dKa.ptr, None, None))
    rows = int(lib.blmm_rotated_rows(ctx.h))
    assert rows == 152
    dX = DevBuf(nbytes=8 * rows * 64)
    ctx.check(lib.blmm_rotate_block_dev(ctx.h, dGa.ptr, 64, dX.ptr, 64))            # fine: prepared
    ctx.synchronize()
    first = dX.get((rows, 64))
    blmm.bulkscan(Yb, Gb, Kb, method="null-grid", ctx=ctx)  # n = 300 on the same context: the eigen front runs again
    assert int(lib.blmm_rotated_rows(ctx.h)) == 0
    assert lib.blmm_rotate_block_dev(ctx.h, dGa.ptr, 64, dX.ptr, 64) != 0 and b"blmm_prepare_dev has not run" in lib.blmm_last_error(ctx.h)
    dL, dh = DevBuf(nbytes=8 * 64 * 8), DevBuf(nbytes=8 * 8)
    rc = lib.blmm_bulkscan_prerotated_dev(ctx.h, C.byref(o), dYa.ptr, 8, 64, dX.ptr, 1, 64, 64, None, 0, dL.ptr, 64, dh.ptr, None)
    assert rc != 0 and b"blmm_prepare_dev has not run" in lib.blmm_last_error(ctx.h)
    # the lower-level seams overwrite Z0 / lambda as well
    ctx.check(lib.blmm_prepare_dev(ctx.h, C.byref(o), 150, None, 0, dKa.ptr, None, None))
    assert int(lib.blmm_rotated_rows(ctx.h)) == 152
    Y0, X0, lam = blmm.transform_rotation(Ya, Ga, Ka, ctx=ctx)
    assert int(lib.blmm_rotated_rows(ctx.h)) == 0
    blmm.fitlmm_bulk(Y0, X0[:, :1], lam, ctx=ctx)
    assert int(lib.blmm_rotated_rows(ctx.h)) == 0
    # and it comes back with a new prepare, giving the same rotated block as before
    ctx.check(lib.blmm_prepare_dev(ctx.h, C.byref(o), 150, None, 0, dKa.ptr, None, None))
    ctx.check(lib.blmm_rotate_block_dev(ctx.h, dGa.ptr, 64, dX.ptr, 64))
    ctx.check(lib.blmm_bulkscan_prerotated_dev(ctx.h, C.byref(o), dYa.ptr, 8, 64, dX.ptr, 1, 64, 64, None, 0, dL.ptr, 64, dh.ptr, None))
    ctx.synchronize()
    assert np.array_equal(dX.get((rows, 64)), first)
    whole = blmm.bulkscan(Ya, Ga, Ka, method="null-exact", ctx=ctx)
    assert np.array_equal(dL.get((8, 64)).T, whole["L"]) and np.array_equal(dh.get(8), whole["h2_null_list"])
    for b in (dKa, dGa, dYa, dX, dL, dh):
        b.free()
    ctx.close()


def test_pvalue_request_does_not_survive_a_failed_call(blmm):
    """blmm_set_log10p_output arms ONE call.  If that call fails its checks the request must be gone: the next, unrelated scan
    must neither compute p-values nor write to the pointer of the failed call."""
    ctx = blmm.Context(0)
    lib = ctx.lib
    Y, G, K, _ = make_data(p=96, m=24, seed=3800)
    n = Y.shape[0]
    dY, dG, dK = DevBuf(np.asfortranarray(Y).ravel("F")), DevBuf(np.asfortranarray(G).ravel("F")), DevBuf(np.asfortranarray(K).ravel("F"))
    dL, dh, dP = DevBuf(nbytes=8 * 96 * 24), DevBuf(nbytes=8 * 24), DevBuf(np.full(96 * 24, -7.0))
    dCov = DevBuf(np.zeros(n * n))
    o = blmm.api._opts(blmm._lib.BLMM_NULL_EXACT)
    bad = blmm.api._opts(blmm._lib.BLMM_NULL_EXACT); bad.decomp_scheme = 99
    calls = {
        "ldL": lambda: lib.blmm_bulkscan_dev(ctx.h, C.byref(o), dY.ptr, n, 24, dG.ptr, 96, None, 0, dK.ptr, None, None, 0, dL.ptr, 95, dh.ptr, None),
        "decomp": lambda: lib.blmm_bulkscan_dev(ctx.h, C.byref(bad), dY.ptr, n, 24, dG.ptr, 96, None, 0, dK.ptr, None, None, 0, dL.ptr, 96, dh.ptr, None),
        "c>=n": lambda: lib.blmm_bulkscan_dev(ctx.h, C.byref(o), dY.ptr, n, 24, dG.ptr, 96, dCov.ptr, n, dK.ptr, None, None, 0, dL.ptr, 96, dh.ptr, None),
        "null": lambda: lib.blmm_bulkscan_dev(ctx.h, C.byref(o), None, n, 24, dG.ptr, 96, None, 0, dK.ptr, None, None, 0, dL.ptr, 96, dh.ptr, None),
    }
    for kind, failing in calls.items():
        assert lib.blmm_set_log10p_output(ctx.h, dP.ptr, 96, 1) == 0
        assert failing() != 0, kind
        ctx.check(lib.blmm_bulkscan_dev(ctx.h, C.byref(o), dY.ptr, n, 24, dG.ptr, 96, None, 0, dK.ptr, None, None, 0, dL.ptr, 96, dh.ptr, None))
        ctx.synchronize()
        assert np.all(dP.get(96 * 24) == -7.0), kind + ": the next call wrote p-values nobody asked for"
    # a request that is honoured does write (the check above is not vacuous)
    assert lib.blmm_set_log10p_output(ctx.h, dP.ptr, 96, 1) == 0
    ctx.check(lib.blmm_bulkscan_dev(ctx.h, C.byref(o), dY.ptr, n, 24, dG.ptr, 96, None, 0, dK.ptr, None, None, 0, dL.ptr, 96, dh.ptr, None))
    ctx.synchronize()
    assert np.all(dP.get(96 * 24) >= 0.0)
    # the host-pointer form: a request that dies with a failed call leaves no matrix behind for blmm_last_log10p to hand out
    assert lib.blmm_set_log10p_output(ctx.h, None, 0, 1) == 0
    Lh = np.empty((96, 24), order="F"); hh = np.empty(24)
    Gf, Kf = np.asfortranarray(G), np.asfortranarray(K)
    rc = lib.blmm_bulkscan(ctx.h, C.byref(o), None, n, 24, blmm.api._p(Gf), 96, None, 0, blmm.api._p(Kf), None, None, 0, blmm.api._p(Lh), blmm.api._p(hh), None)
    assert rc != 0
    r = blmm.bulkscan(Y, G, K, method="null-exact", ctx=ctx)
    Pm = np.empty((96, 24), order="F")
    assert "log10Pvals_mat" not in r
    # and the Python mirror arms only after its own argument checks
    with pytest.raises(blmm.BulkLMMError):
        blmm.bulkscan(Y, G[:-1], K, method="null-exact", output_pvals=True, ctx=ctx)
    r2 = blmm.bulkscan(Y, G, K, method="null-exact", ctx=ctx)
    assert np.array_equal(r2["L"], r["L"])
    for b in (dY, dG, dK, dL, dh, dP, dCov):
        b.free()
    ctx.close()


def test_tuning_keys_replace_the_environment_and_the_environment_is_ignored(blmm, monkeypatch):
    """The switches that change the arithmetic are context properties; a BLMM_* variable in the caller's environment changes
    nothing unless BLMM_DEV_ENV=1 says it is a developer's run."""
    Y, G, K, _ = make_data(p=200, m=60, seed=3900)
    ctx = blmm.Context(0)
    base = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1, "eigen", 0, ctx, return_status=True)
    assert base[2].lowrank_fallback == 0 and base[2].lowrank_rank > 0
    monkeypatch.delenv("BLMM_DEV_ENV", raising=False)
    monkeypatch.setenv("BLMM_LR_TOL", "0")                 # a stray variable: ignored
    monkeypatch.setenv("BLMM_EXACT", "full")
    again = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1, "eigen", 0, ctx, return_status=True)
    assert again[2].lowrank_fallback == 0 and again[2].lowrank_rank == base[2].lowrank_rank and np.array_equal(again[0], base[0])
    monkeypatch.setenv("BLMM_DEV_ENV", "1")                # now it is a developer's run
    dev = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1, "eigen", 0, ctx, return_status=True)
    assert dev[2].lowrank_rank == 0                        # BLMM_EXACT=full: the full-rank kernel, no weight basis
    monkeypatch.delenv("BLMM_DEV_ENV"); monkeypatch.delenv("BLMM_LR_TOL"); monkeypatch.delenv("BLMM_EXACT")
    assert ctx.get_tuning("lr_tol") == 1e-13 and ctx.get_tuning("pval_fused") == 1 and ctx.get_tuning("lr_split") == -1
    ctx.set_tuning("exact_full_rank", 1)
    full = blmm.api._bulkscan_call(blmm._lib.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1, "eigen", 0, ctx, return_status=True)
    assert full[2].lowrank_rank == 0 and np.array_equal(full[1], base[1])
    assert np.abs(full[0] - base[0]).max() <= 1e-9 * max(1.0, np.abs(base[0]).max())
    for bad in (("lr_tol", -1.0), ("lr_segments", 9), ("eigen_solver", 0.5), ("no_such_key", 1)):
        with pytest.raises(blmm.BulkLMMError):
            ctx.set_tuning(*bad)
    ctx.set_tuning("defaults", 0)
    assert ctx.get_tuning("exact_full_rank") == 0
    ctx.close()
