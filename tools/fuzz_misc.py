"""Randomised checks of what sits around the scan (developer tool, GPU box): the multi-GPU entry point against the single-context
call (shards on one device, every gather mode), the consumers of L against NumPy / SciPy, calcKinship and its rounding, the
readers.  python tools/fuzz_misc.py [ncases] [seed]"""
import sys, os, time, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import scipy.stats as ss
import bulklmm_jl_amd as blmm
import oracle.bulklmm_oracle as O
from common import make_data

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = blmm.Context(0)
fails = 0
t0 = time.time()
tmp = tempfile.mkdtemp()
for case in range(ncases):
    what = str(rng.choice(["multi", "multi", "pvals", "threshold", "quantiles", "kinship", "readers"]))
    desc = f"case {case}: {what}"
    try:
        if what == "multi":
            n = int(rng.choice([31, 79, 130])); p = int(rng.choice([5, 64, 200])); m = int(rng.choice([1, 3, 64, 130, 1100]))
            ndev = int(rng.choice([1, 2, 3, 5])); method = str(rng.choice(["null-exact", "null-grid", "alt-grid"]))
            gather = str(rng.choice(["host_shards", "none", "allgather"]))
            if method == "alt-grid": m = min(m, 130)
            Y, G, K, _ = make_data(n=n, p=p, m=m, seed=case, bxd=(n == 79))
            desc += f" n={n} p={p} m={m} ndev={ndev} {method} {gather}"
            one = blmm.bulkscan(Y, G, K, method=method, ctx=ctx)
            mc = blmm.MultiContext([0] * ndev)
            got = blmm.bulkscan_multi(mc, Y, G, K, method=method, gather=gather)
            mc.close()
            if gather == "host_shards":      # the host result is only defined for this mode; the others leave L on the devices
                key = "h2_panel" if method == "alt-grid" else "h2_null_list"
                assert np.array_equal(got["L"], one["L"]) and np.array_equal(got[key], one[key]), "multi != single"
        elif what == "pvals":
            p, m = int(rng.integers(1, 400)), int(rng.integers(1, 60)); df = int(rng.choice([1, 1, 2, 3, 7]))
            Lm = np.asfortranarray(rng.gamma(1.0, 2.0, (p, m)) * rng.choice([1.0, 10.0]))
            got = blmm.lod2log10p(Lm, df, ctx=ctx)
            ref = -ss.chi2.logsf(2 * np.log(10) * Lm, df) / np.log(10)
            fin = np.isfinite(ref)
            assert np.allclose(got[fin], ref[fin], rtol=1e-9, atol=1e-12), "log10p"
            desc += f" p={p} m={m} df={df}"
        elif what == "threshold":
            p, m = int(rng.integers(1, 700)), int(rng.integers(1, 90)); thr = float(rng.uniform(0.5, 6.0))
            Lm = np.asfortranarray(rng.gamma(1.0, 1.5, (p, m)))
            ii, jj, ll = blmm.lod_threshold(Lm, thr, ctx=ctx, cap=int(rng.choice([4, 1024])))
            ri, rj = np.nonzero(Lm > thr)
            order = np.lexsort((ri, rj))
            assert np.array_equal(ii, ri[order]) and np.array_equal(jj, rj[order]) and np.array_equal(ll, Lm[ri[order], rj[order]]), "triplets"
            desc += f" p={p} m={m} hits={ii.size}"
        elif what == "quantiles":
            p, nperm = int(rng.integers(1, 500)), int(rng.choice([1, 2, 17, 1000, 5000]))
            Lp = np.asfortranarray(rng.gamma(1.0, 1.5, (p, nperm)))
            sig = np.sort(rng.uniform(0.001, 0.5, int(rng.integers(1, 5))))
            got = blmm.get_thresholds(Lp, sig, ctx=ctx)
            ref = np.quantile(Lp.max(axis=0), 1.0 - sig)
            assert np.allclose(got["thrs"], ref, rtol=1e-13, atol=0), "quantiles"
            desc += f" p={p} nperm={nperm}"
        elif what == "kinship":
            n, p = int(rng.integers(2, 300)), int(rng.integers(1, 900))
            G = rng.random((n, p))
            K = blmm.calcKinship(G, ctx=ctx)
            assert np.abs(K - O.calcKinship(G)).max() <= 1e-13, "kinship"
            d = int(rng.choice([0, 4, 12]))
            Kr = blmm.calcKinship(G, ctx=ctx, digits=d)
            assert np.array_equal(Kr, np.round(K, d)), "rounded kinship"
            desc += f" n={n} p={p} digits={d}"
        else:
            n, pm = int(rng.integers(1, 40)), int(rng.integers(1, 30))
            prob = rng.random((n, pm)); geno = np.empty((n, 2 * pm)); geno[:, 0::2] = prob; geno[:, 1::2] = 1 - prob
            f = os.path.join(tmp, f"g{case}.csv")
            eol = str(rng.choice(["\n", "\r\n"]))
            with open(f, "w", newline="") as fh:
                fh.write(",".join(['"id"'] + [f'"m{j}"' for j in range(2 * pm)]) + eol)
                for i in range(n):
                    fh.write(",".join([f'"s,{i}"'] + [repr(float(x)) for x in geno[i]]) + eol)
            assert np.array_equal(blmm.readGenoProb(f), geno) and np.array_equal(blmm.readGenoProb_ExcludeComplements(f), prob), "csv"
            desc += f" n={n} pm={pm}"
        print("ok  ", desc, flush=True)
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("FAIL", desc, "->", repr(e)[:300], flush=True)
print(f"{ncases - fails}/{ncases} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
