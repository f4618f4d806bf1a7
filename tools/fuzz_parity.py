"""Randomised parity sweep (developer tool, run on the GPU box): random shapes / options through the host mirror against
the oracle with the criteria of tests/test_gpu_parity.py.  python tools/fuzz_parity.py [ncases] [seed]"""
import sys, os, time, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as blmm
import oracle.bulklmm_oracle as O
from common import make_data, assert_lod_close

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
fails = 0
t0 = time.time()


def check_reduced(method, L, h2, Y, G, K, Cov, w, oi, kw, grid):
    """blmm_bulkscan_reduced (the scan kernels reduce in their epilogues / the routes through a resident matrix) against the
    reductions of the matrix the ordinary call returned: bit for bit."""
    thr = float(np.quantile(L, 0.98)) if L.size else 1.0
    red = blmm.bulkscan_reduced(Y, G, K, Cov, method=method, h2_grid=grid, threshold=thr, cap=16, weights=w, optim_interval=oi, **kw)
    Lm = np.where(np.isnan(L), -np.inf, L)
    if L.shape[0] > 0:
        arg = np.argmax(Lm, axis=0)
        assert np.array_equal(red["max_lod"], Lm[arg, np.arange(L.shape[1])]) and np.array_equal(red["argmax"], arg), "reduced: maxima"
    i, j = np.nonzero(L > thr)
    o = np.lexsort((i, j))
    ti, tj, tl = red["triplets"]
    assert np.array_equal(ti, i[o]) and np.array_equal(tj, j[o]) and np.array_equal(tl, L[i[o], j[o]]), "reduced: triplets"
    if method != "alt-grid":
        assert np.array_equal(red["h2_null_list"], h2), "reduced: h2"


for case in range(ncases):
    n = int(rng.choice([5, 8, 13, 31, 47, 64, 79, 80, 93, 100, 111, 124, 125, 140, 160, 200, 260]))
    kkind = str(rng.choice(["markers", "markers", "one-marker", "few-markers", "duplicated-individuals"]))
    p = int(rng.choice([1, 2, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300]))
    m = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 70, 128, 129, 1030, 2100]))
    ncov = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 7, 8, 11, 19, 31]))      # beyond 7: the run-time-c kernels (kernels_dyn.hip)
    if ncov + 2 >= n: ncov = 0
    method = str(rng.choice(["null-exact", "null-exact", "null-grid", "alt-grid", "perms", "scan-alt"]))
    if ncov > 29 and method == "scan-alt": ncov = 19                    # scan_alt: its design [Z0 x] has c + 1 <= 32 columns (k_dyn_alt_brent beyond 8)
    if ncov > 7 and n < 3 * ncov: ncov = 7                              # keep the null design comfortably full rank
    oi = int(rng.choice([1, 1, 1, 2, 3]))
    reml = bool(rng.random() < 0.25)
    svd = bool(rng.random() < 0.15)
    use_w = bool(rng.random() < 0.2)
    prior = (1.0, 0.1) if rng.random() < 0.2 else (1.0, 0.0)
    if m > 100 and (n > 100 or method == "alt-grid"): m = 70          # keep the oracle quick
    if m > 1100 and n > 64: m = 1030
    if n > 140: m = min(m, 17); p = min(p, 129)
    if n < 10: ncov = 0
    desc = f"case {case}: n={n} p={p} m={m} ncov={ncov} K={kkind} {method} reml={reml} svd={svd} weights={use_w} prior={prior} optim_interval={oi}"
    try:
        Y, G, K, Cov = make_data(n=n, p=p, m=m, seed=1000 + case + 7919 * seed0, ncov=ncov, bxd=(n == 79))
        # a marker that is constant over the individuals (common at n = 5 .. 13) is collinear with the intercept: its projected
        # norm is rounding noise and r is garbage on both sides (the reference may throw a DomainError) -- not
        # a parity question, so such columns get random values instead
        const = np.ptp(G, axis=0) == 0
        if const.any():
            G = G.copy(); G[:, const] = np.random.default_rng(case).random((n, int(const.sum())))
        # degenerate kinships: the spectra that stress the eigensolvers (blocks of exactly equal / zero eigenvalues)
        if kkind != "markers" and n >= 13:
            gk = (np.random.default_rng(case + 5).random((n, 3)) < 0.5).astype(np.float64)
            if kkind == "one-marker":
                K = O.calcKinship(gk[:, :1])
            elif kkind == "few-markers":
                K = O.calcKinship(gk)
            else:
                Gd = G.copy(); Gd[n // 2:] = Gd[: n - n // 2]
                K = np.round(O.calcKinship(Gd), 12)
        w = rng.uniform(0.5, 2.0, size=n) if use_w else None
        kw = dict(reml=reml, decomp_scheme="svd" if svd else "eigen", prior_variance=prior[0], prior_sample_size=prior[1])
        grid = [i / 10.0 for i in range(10)]
        if method == "null-exact":
            got = blmm.bulkscan_null(Y, G, K, Cov, weights=w, optim_interval=oi, **kw)
            ref = O.bulkscan_null(Y, G, K, Covar=Cov, weights=w, optim_interval=oi, **kw)
            dh = np.abs(got.h2_null_list - ref.h2_null_list)
            # strict: every h2 within 1e-6 of the oracle's, EXCEPT a trait whose profile likelihood has two local maxima
            # and Brent (a local method) ended in the other one -- it must then be a local maximum of the oracle's own
            # likelihood (tests/test_gpu_guard.py::test_fullsize_h2_audit_all_traits measures the rate: 2 of 35,554)
            for j in np.flatnonzero(dh > 1e-6):
                Yw, Gw, Cw, Kw, ai = (Y, G, Cov, K, True) if w is None else O._apply_weights(Y, G, Cov if Cov is not None else np.ones((n, 0)), K, w, True)
                Zc = np.ones((n, 1)) if Cw is None else (np.hstack([np.ones((n, 1)), Cw]) if ai else Cw)
                y0, X0, lam0 = O.transform_rotation(Yw[:, j:j + 1], np.hstack([Zc, Gw[:, :1]]), Kw, addIntercept=False, decomp_scheme=kw["decomp_scheme"])
                ell = lambda h: O.wls(y0, X0[:, :Zc.shape[1]], O.makeweights(h, lam0), list(prior), reml=reml).ell
                hg = float(got.h2_null_list[j]); eg = ell(hg)
                assert eg >= ell(min(hg + 1e-3, 1 - 1e-9)) - 1e-9 * abs(eg) and eg >= ell(max(hg - 1e-3, 0.0)) - 1e-9 * abs(eg), \
                    f"h2[{j}]: {hg} vs {ref.h2_null_list[j]} is not a local maximum"
            okc = dh <= 1e-6
            assert np.sum((got.L[:, okc] - ref.L[:, okc]) ** 2, axis=0).max() <= 1e-7, "sum d^2"
            pin = O.bulkscan_null(Y, G, K, Covar=Cov, weights=w, h2_override=got.h2_null_list, **kw)
            # strict for every trait: the h2 -> 1 boundary traits with badly conditioned weighted covariates (case 237 of seed
            # 201: n = 13, 8 null covariates, cond 2e4) are re-scanned with an orthogonalised projection (kernels_dyn.hip)
            assert_lod_close(got.L, pin.L)
            if case % 2 == 1:
                check_reduced("null-exact", got.L, got.h2_null_list, Y, G, K, Cov, w, oi, kw, grid)
            if case % 3 == 0:       # `output_pvals` written by the scan itself (low-rank, exact, dyn and re-scan kernels alike)
                rp = blmm.bulkscan(Y, G, K, Cov, method="null-exact", weights=w, optim_interval=oi, output_pvals=True, **kw)
                refp = O.lod2log10p(rp["L"], 1)
                assert np.array_equal(rp["L"], got.L) and np.all(np.abs(rp["log10Pvals_mat"] - refp) <= 1e-10 * np.abs(refp) + 1e-14), "output_pvals"
        elif method == "perms":
            nperms = int(rng.choice([1, 5, 64, 130]))
            pidx = O.make_perm_idx(n, nperms, case)
            skw = dict(reml=reml, decomp_scheme="svd" if svd else "eigen", prior_variance=prior[0], prior_sample_size=prior[1])
            f32 = bool(rng.random() < 0.4)
            got = blmm.scan(Y[:, 0], G, K, Cov, permutation_test=True, nperms=nperms, perm_idx=pidx, weights=w,
                            perm_precision="f32" if f32 else "f64", **skw)
            cov1 = np.ones((n, 1)) if Cov is None else np.hstack([np.ones((n, 1)), Cov])
            if w is None:
                rot = blmm.transform_rotation(Y[:, :1], np.hstack([cov1, G]), K, addIntercept=False, decomp_scheme=skw["decomp_scheme"])
                pin = O.scan(Y[:, 0], G, K, covar=cov1, addIntercept=False, permutation_test=True, nperms=nperms, perm_idx=pidx,
                             h2_override=got["h2_null"], rotation_override=rot, **skw)
                assert_lod_close(got["lod"], pin["lod"])
                if f32:
                    e = np.abs(got["L_perms"].astype(np.float64) - pin["L_perms"])
                    assert np.all(e <= 1e-3 * np.abs(pin["L_perms"]) + 1e-4), "f32 perms"
                else:
                    assert_lod_close(got["L_perms"], pin["L_perms"])
            else:
                assert np.isfinite(got["L_perms"]).all()
        elif method == "scan-alt":
            pp = min(p, 65)                                        # the oracle runs one Python Brent search per marker
            true_w = bool(rng.random() < 0.3)
            skw = dict(reml=reml, decomp_scheme="svd" if svd else "eigen", prior_variance=prior[0], prior_sample_size=prior[1],
                       optim_interval=oi, weights=w)
            got = blmm.scan(Y[:, :1], G[:, :pp], K, Cov, assumption="alt", alt_true_weights=true_w, **skw)
            own = O.scan(Y[:, :1], G[:, :pp], K, covar=Cov, assumption="alt", true_weights=true_w, **skw)
            null_differs = abs(got["h2_null"] - own["h2_null"]) > 1e-6
            if null_differs:
                # the same rule as for null-exact above: a two-humped null profile (rank-deficient kinship, REML) on which the two
                # searches -- both local -- ended on different humps; the device's must be a local maximum of the ORACLE's likelihood
                # (case 202 of seed 411: K of rank 39 at n = 260, device h2 = 6e-16 with ell -154.885, oracle 0.129 with -155.437)
                Yw, Gw, Cw, Kw, ai = (Y[:, :1], G, Cov, K, True) if w is None else O._apply_weights(Y[:, :1], G, Cov if Cov is not None else np.ones((n, 0)), K, w, True)
                Zc = np.ones((n, 1)) if Cw is None else (np.hstack([np.ones((n, 1)), Cw]) if ai else Cw)
                y0, X0, lam0 = O.transform_rotation(Yw, np.hstack([Zc, Gw[:, :1]]), Kw, addIntercept=False, decomp_scheme=skw["decomp_scheme"])
                ell = lambda h: O.wls(y0, X0[:, :Zc.shape[1]], O.makeweights(h, lam0), list(prior), reml=reml).ell
                hg = float(got["h2_null"]); eg = ell(hg)
                assert eg >= ell(min(hg + 1e-3, 1 - 1e-9)) - 1e-9 * abs(eg) and eg >= ell(max(hg - 1e-3, 0.0)) - 1e-9 * abs(eg), \
                    f"h2_null {hg} vs {own['h2_null']} is not a local maximum"
            pin = O.scan(Y[:, :1], G[:, :pp], K, covar=Cov, assumption="alt", true_weights=true_w,
                         h2_each_override=got["h2_each_marker"], h2_null_override=got["h2_null"], **skw)
            assert_lod_close(got["lod"], pin["lod"], atol=1e-9)
            # each side's own per-marker search: h2 may differ where the profile is flat or two-humped, the LOD then barely
            dl = np.abs(got["lod"] - own["lod"])
            if pp >= 10 and not null_differs:      # (a handful of markers cannot carry a quantile: n = 8 with REML has flat profiles)
                assert np.quantile(dl, 0.9) <= 1e-6 * max(1.0, np.abs(own["lod"]).max()) + 1e-7, "scan_alt LOD"
        elif method == "null-grid":
            got = blmm.bulkscan_null_grid(Y, G, K, grid, Cov, weights=w, **kw)
            ref = O.bulkscan_null_grid(Y, G, K, grid, Covar=Cov, weights=w, **kw)
            same = got.h2_null_list == ref.h2_null_list      # Ell ties between grid points may resolve differently
            assert same.mean() >= 0.98, "grid choice"
            assert_lod_close(got.L[:, same], ref.L[:, same])
            if case % 2 == 1:
                check_reduced("null-grid", got.L, got.h2_null_list, Y, G, K, Cov, w, 1, kw, grid)
        else:
            if ncov > 0 and Cov is not None and Cov.shape[1] + 1 > 1:
                pass
            got = blmm.bulkscan_alt_grid(Y, G, K, grid, Cov, weights=w, **kw)
            ref = O.bulkscan_alt_grid(Y, G, K, grid, Covar=Cov, weights=w, **kw)
            assert_lod_close(got.L, ref.L, atol=1e-9)
            if case % 4 == 1:
                check_reduced("alt-grid", got.L, None, Y, G, K, Cov, w, 1, kw, grid)
        print("ok  ", desc, flush=True)
    except blmm.BulkLMMError as e:
        # an input the reference rejects (e.g. a zero-norm marker at n = 5): fine if the oracle raises the same message
        try:
            {"null-exact": lambda: O.bulkscan_null(Y, G, K, Covar=Cov, weights=w, optim_interval=oi, **kw),
             "null-grid": lambda: O.bulkscan_null_grid(Y, G, K, grid, Covar=Cov, weights=w, **kw),
             "alt-grid": lambda: O.bulkscan_alt_grid(Y, G, K, grid, Covar=Cov, weights=w, **kw),
             "perms": lambda: O.scan(Y[:, 0], G, K, covar=Cov, permutation_test=True, nperms=4, weights=w),
             "scan-alt": lambda: O.scan(Y[:, :1], G[:, :min(p, 65)], K, covar=Cov, assumption="alt", weights=w)}[method]()
            # a zero-norm marker that only one side detects: degenerate inputs (n = 5) put the projected norm at rounding level,
            # within a factor of a few of the reference's eps threshold on either side
            same = "Dividing by zeros" in e.msg and n <= 8
        except O.BulkLMMError as oe:
            same = str(oe) == e.msg
        if same:
            print("ok  ", desc, "(both sides raise:", e.msg + ")", flush=True)
        else:
            fails += 1
            print("FAIL", desc, "->", repr(e)[:300], flush=True)
    except O.BulkLMMError as oe:
        # the oracle rejects the input and the device did not: only the degenerate zero-norm case (n <= 8, projected marker
        # norm at rounding level, a factor of a few around the reference's eps threshold) is let through
        if "Dividing by zeros" in str(oe) and n <= 8:
            print("ok  ", desc, "(the oracle alone raises:", str(oe) + ")", flush=True)
        else:
            fails += 1
            print("FAIL", desc, "->", repr(oe)[:300], flush=True)
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("FAIL", desc, "->", repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
print(f"{ncases - fails}/{ncases} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
