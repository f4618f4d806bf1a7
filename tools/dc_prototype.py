"""NumPy model of the device eigensolver for 124 < n (kernels_eig.hip): Householder tridiagonalisation, Cuppen's divide
and conquer on the tridiagonal matrix (deflation as LAPACK dlaed2, secular roots relative to the nearer pole, Gu-Eisenstat
re-derived z for orthogonal vectors), back-transformation.  Same data flow, index conventions and tolerances as the HIP
kernels, so that each kernel can be checked against one function here.  python tools/dc_prototype.py [n]"""
import sys
import numpy as np

EPS = np.finfo(np.float64).eps


def sytrd(A):
    """d, e, V (row k = reflector k: v[k+1] = 1 implicitly stored explicitly), tau; lower-variant as the kernel."""
    A = A.copy()
    n = A.shape[0]
    d = np.zeros(n); e = np.zeros(n - 1); tau = np.zeros(max(n - 2, 0)); V = np.zeros((n, n))
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        alpha = x[0]
        xn = np.linalg.norm(x[1:])
        if xn == 0.0:
            t = 0.0; beta = alpha; v = np.zeros_like(x); v[0] = 1.0
        else:
            beta = -np.copysign(np.hypot(alpha, xn), alpha)
            t = (beta - alpha) / beta
            v = x / (alpha - beta); v[0] = 1.0
        d[k] = A[k, k]; e[k] = beta; tau[k] = t; V[k, k + 1:] = v
        A22 = A[k + 1:, k + 1:]
        p = t * (A22 @ v)
        w = p - 0.5 * t * (p @ v) * v
        A22 -= np.outer(v, w) + np.outer(w, v)
    d[n - 2] = A[n - 2, n - 2]; d[n - 1] = A[n - 1, n - 1]; e[n - 2] = A[n - 1, n - 2]
    return d, e, V, tau


def tql(d, e):
    """implicit QL with eigenvectors (EISPACK tql2), ascending order."""
    n = len(d); d = d.copy(); e = np.append(e.copy(), 0.0); Z = np.eye(n)
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n - 1:
                dd = abs(d[m]) + abs(d[m + 1])
                if abs(e[m]) <= EPS * dd:
                    break
                m += 1
            if m == l:
                break
            it += 1
            assert it < 60
            g = (d[l + 1] - d[l]) / (2.0 * e[l]); r = np.hypot(g, 1.0)
            g = d[m] - d[l] + e[l] / (g + np.copysign(r, g))
            s = c = 1.0; p = 0.0
            i = m - 1
            broke = False
            while i >= l:
                f = s * e[i]; b = c * e[i]
                r = np.hypot(f, g); e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p; e[m] = 0.0; broke = True
                    break
                s = f / r; c = g / r; g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b; p = s * r; d[i + 1] = g + p; g = c * r - b
                f2 = Z[:, i + 1].copy(); Z[:, i + 1] = s * Z[:, i] + c * f2; Z[:, i] = c * Z[:, i] - s * f2
                i -= 1
            if broke:
                continue
            d[l] -= p; e[l] = g; e[m] = 0.0
    o = np.argsort(d, kind="stable")
    return d[o], Z[:, o]


def secular_root(i, dl, zl, rho):
    """root i of 1 + rho sum z_j^2 / (dl_j - lam): returns (origin, tau, delta[j] = dl_j - lam).  Origin = the nearer pole;
    Bunch-Nielsen-Sorensen iteration (psi / phi each matched by r + s / (pole - t)), bracket kept by the sign of f."""
    K = len(dl); z2 = zl ** 2
    if i < K - 1:
        lo = dl[i]; mid = 0.5 * (dl[i + 1] - lo); dj = dl - lo
        fmid = 1.0 + rho * np.sum(z2 / (dj - mid))
        if fmid > 0:
            org = i; a, b = 0.0, mid
        else:
            org = i + 1; a, b = -mid, 0.0
        if fmid == 0:
            org = i; a = b = mid
    else:
        org = K - 1; a, b = 0.0, rho * np.sum(z2)
    dorg = dl[org]; dj = dl - dorg
    poleL = dl[i] - dorg; poleR = dl[i + 1] - dorg if i < K - 1 else 0.0
    t = 0.5 * (a + b)
    for it in range(100):
        if not a < b:
            break
        q = 1.0 / (dj - t); zq = z2 * q
        ps = rho * np.sum(zq[:i + 1]); psp = rho * np.sum((zq * q)[:i + 1])
        ph = rho * np.sum(zq[i + 1:]); php = rho * np.sum((zq * q)[i + 1:])
        f = 1.0 + ps + ph
        if f == 0:
            break
        if f > 0:
            b = t
        else:
            a = t
        dL = poleL - t; sps = psp * dL * dL; rps = ps - psp * dL
        if i < K - 1:
            dR = poleR - t; sph = php * dR * dR; rph = ph - php * dR
            c = 1.0 + rps + rph
            a2 = c; a1 = -(c * (poleL + poleR) + sps + sph); a0 = c * poleL * poleR + sps * poleR + sph * poleL
            disc = a1 * a1 - 4.0 * a2 * a0; tn = t
            if disc >= 0:
                qq = -0.5 * (a1 + np.copysign(np.sqrt(disc), a1))
                r1 = qq / a2 if a2 != 0 else a; r2 = a0 / qq if qq != 0 else a
                tn = r1 if a < r1 < b else r2
        else:
            tn = poleL + sps / (1.0 + rps)
        if not (a < tn < b):
            tn = 0.5 * (a + b)
        if tn == t or b - a <= 2 * EPS * max(abs(a), abs(b)) or abs(tn - t) <= EPS * abs(tn):
            t = tn
            break
        t = tn
    return org, t, dj - t


def merge(d1, Q1, d2, Q2, beta):
    """eigen-decomposition of diag(T1', T2') + |beta| v v' from those of T1' and T2'."""
    n1, n2 = len(d1), len(d2); N = n1 + n2
    Q = np.zeros((N, N)); Q[:n1, :n1] = Q1; Q[n1:, n1:] = Q2
    d = np.concatenate([d1, d2])
    z = np.concatenate([Q1[-1, :], np.sign(beta) * Q2[0, :]]) / np.sqrt(2.0)
    rho = 2.0 * abs(beta)
    order = np.argsort(d, kind="stable")
    tol = 8.0 * EPS * max(np.abs(d).max(), np.abs(z).max())
    keep_d, keep_z, keep_col, defl_d, defl_col = [], [], [], [], []
    if rho * np.abs(z).max() <= tol:
        defl_d = list(d[order]); defl_col = list(order)
    else:
        pj = None
        for nj in order:
            if rho * abs(z[nj]) <= tol:
                defl_d.append(d[nj]); defl_col.append(nj)
                continue
            if pj is None:
                pj = nj
                continue
            s = z[pj]; c = z[nj]; tau = np.hypot(c, s); t = d[nj] - d[pj]; c /= tau; s = -s / tau
            if abs(t * c * s) <= tol:
                z[nj] = tau; z[pj] = 0.0
                qp = Q[:, pj].copy(); qn = Q[:, nj].copy()
                Q[:, pj] = c * qp + s * qn; Q[:, nj] = c * qn - s * qp
                tt = d[pj] * c * c + d[nj] * s * s
                d[nj] = d[pj] * s * s + d[nj] * c * c; d[pj] = tt
                defl_d.append(d[pj]); defl_col.append(pj)
                pj = nj
            else:
                keep_d.append(d[pj]); keep_z.append(z[pj]); keep_col.append(pj)
                pj = nj
        keep_d.append(d[pj]); keep_z.append(z[pj]); keep_col.append(pj)
    K = len(keep_d)
    lam = np.empty(K); Wt = np.zeros((K, K))
    if K:
        dl = np.array(keep_d); zl = np.array(keep_z)
        Dm = np.empty((K, K))
        for i in range(K):
            org, t, delta = secular_root(i, dl, zl, rho)
            lam[i] = dl[org] + t; Dm[i] = delta
        # Gu-Eisenstat: z-hat from the computed roots
        zh = np.empty(K)
        for j in range(K):
            prod = Dm[j, j]
            for i in range(K):
                if i != j:
                    prod *= Dm[i, j] / (dl[j] - dl[i])
            zh[j] = np.copysign(np.sqrt(abs(prod)), zl[j])
        for i in range(K):
            v = zh / Dm[i]
            Wt[i] = v / np.linalg.norm(v)
    allv = np.concatenate([lam, np.array(defl_d)]) if K or defl_d else np.array([])
    pos = np.argsort(np.argsort(allv, kind="stable"), kind="stable")
    Qo = np.zeros((N, N)); do = np.zeros(N)
    if K:
        Qn = Q[:, keep_col] @ Wt.T
        for i in range(K):
            Qo[:, pos[i]] = Qn[:, i]; do[pos[i]] = lam[i]
    for t, col in enumerate(defl_col):
        Qo[:, pos[K + t]] = Q[:, col]; do[pos[K + t]] = defl_d[t]
    return do, Qo, K


def tridiag_dc(d, e, leaf=32):
    n = len(d)
    nl = 1
    while (n + nl - 1) // nl > leaf:
        nl *= 2
    bounds = [round(i * n / nl) for i in range(nl + 1)]
    d = d.copy()
    for b in bounds[1:-1]:
        d[b - 1] -= abs(e[b - 1]); d[b] -= abs(e[b - 1])
    nodes = [(bounds[i], bounds[i + 1]) + tql(d[bounds[i]:bounds[i + 1]], e[bounds[i]:bounds[i + 1] - 1]) for i in range(nl)]
    kept = []
    while len(nodes) > 1:
        nxt = []
        for a, b in zip(nodes[0::2], nodes[1::2]):
            do, Qo, K = merge(a[2], a[3], b[2], b[3], e[a[1] - 1])
            kept.append((b[1] - a[0], K))
            nxt.append((a[0], b[1], do, Qo))
        nodes = nxt
    return nodes[0][2], nodes[0][3], kept


def back_transform(V, tau, Z):
    U = Z.copy()
    for k in range(len(tau) - 1, -1, -1):
        v = V[k, k + 1:]
        U[k + 1:, :] -= tau[k] * np.outer(v, v @ U[k + 1:, :])
    return U


def eigh_dc(A, leaf=32):
    d, e, V, tau = sytrd(A)
    lam, Z, kept = tridiag_dc(d, e, leaf)
    return lam, back_transform(V, tau, Z), kept


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(1)
    tests = {}
    G = (rng.random((n, 4 * n)) < 0.5).astype(float)
    X = G - 0.5
    K = 2 * X @ X.T / X.shape[1] + 0.5; np.fill_diagonal(K, 1.0)
    tests["kinship"] = np.round(K, 12)
    B = rng.standard_normal((n, n // 3)); tests["rank_deficient"] = B @ B.T / (n // 3)
    tests["identity_plus"] = np.eye(n) + 1e-9 * (lambda S: S + S.T)(rng.standard_normal((n, n)))
    Qr, _ = np.linalg.qr(rng.standard_normal((n, n)))
    tests["clusters"] = (Qr * np.repeat([0.5, 7.0, 7.0 + 1e-10, 100.0], [n // 4, n // 4, n // 4, n - 3 * (n // 4)])) @ Qr.T
    tests["wilkinson"] = np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    for name, A in tests.items():
        A = 0.5 * (A + A.T)
        lam, U, kept = eigh_dc(A)
        ref = np.linalg.eigvalsh(A)
        sc = max(np.abs(ref).max(), 1e-300)
        print(f"{name:16s} n={n} |lam-ref|/|A| {np.abs(np.sort(lam) - ref).max() / sc:.2e}  orth {np.abs(U.T @ U - np.eye(n)).max():.2e}  "
              f"resid {np.abs(A @ U - U * lam).max() / sc:.2e}  sorted {bool(np.all(np.diff(lam) >= 0))}  top merge kept {kept[-1]}")
