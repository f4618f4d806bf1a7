"""Randomised check of the device eigensolvers alone (developer tool, GPU box): orthogonality of U, K = U diag U', eigenvalues
against LAPACK, on kinship-like and deliberately degenerate spectra.  python tools/fuzz_eig.py [ncases] [seed]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as blmm
import oracle.bulklmm_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = blmm.Context(0)
fails = 0
t0 = time.time()
for case in range(ncases):
    n = int(rng.choice([3, 16, 64, 79, 92, 124, 125, 126, 160, 200, 257, 333, 500, 640, 1000]))
    kind = str(rng.choice(["markers", "one-marker", "few-markers", "dup-rows", "clusters", "identity", "rank1+diag", "tiny", "huge", "negative"]))
    if kind in ("markers", "dup-rows"):
        p = int(rng.choice([n // 2 + 1, 2 * n, 5 * n]))
        G = (rng.random((n, p)) < 0.5).astype(np.float64)
        if kind == "dup-rows":
            G[n // 2:] = G[: n - n // 2]
        K = np.round(O.calcKinship(G), 12)
    elif kind in ("one-marker", "few-markers"):
        K = O.calcKinship((rng.random((n, 1 if kind == "one-marker" else 3)) < 0.5).astype(np.float64))
    else:
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        if kind == "clusters":
            lam = np.repeat(rng.uniform(0.1, 10.0, 4), -(-n // 4))[:n]
        elif kind == "identity":
            lam = np.full(n, 2.5)
        elif kind == "rank1+diag":
            lam = np.concatenate([[float(n)], np.full(n - 1, 0.5)])
        elif kind == "tiny":
            lam = rng.uniform(0.1, 10.0, n) * 1e-40
        elif kind == "huge":
            lam = rng.uniform(0.1, 10.0, n) * 1e40
        else:
            lam = rng.uniform(-1.0, 10.0, n)
        K = (Q * lam) @ Q.T
        K = (K + K.T) / 2
        if kind == "identity":
            K = 2.5 * np.eye(n)
    desc = f"case {case}: n={n} {kind}"
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Ut, _, lam_d = blmm.transform_rotation(np.eye(n), np.ones((n, 2)), K, addIntercept=False, ctx=ctx)
        U = np.asarray(Ut).T
        lam_d = np.asarray(lam_d)
        sc = max(np.abs(K).max(), 1e-300)
        orth = np.abs(U.T @ U - np.eye(n)).max()
        res = np.abs(K - (U * lam_d) @ U.T).max() / sc
        ev = np.abs(np.sort(lam_d) - np.linalg.eigvalsh(K)).max() / sc
        ok = orth <= 1e-11 and res <= 1e-11 * n and ev <= 1e-11
        print("ok  " if ok else "FAIL", desc, f"orth {orth:.1e} resid {res:.1e} eig {ev:.1e}", flush=True)
        fails += 0 if ok else 1
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("FAIL", desc, "->", repr(e)[:200], flush=True)
        ctx = blmm.Context(0)
print(f"{ncases - fails}/{ncases} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
