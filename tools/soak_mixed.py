"""Stability soak with ALTERNATING shapes and methods on one context (developer tool, GPU box): workspace growth, the split h2
search's streams and events, and the two eigensolvers take turns; every result must be bit-identical to the first time its
configuration ran.  python tools/soak_mixed.py [rounds]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bulklmm_jl_amd as B
from common import make_data
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfgs = [dict(n=79, p=700, m=3000, method="null-exact"), dict(n=200, p=300, m=1500, method="null-exact"), dict(n=79, p=700, m=3000, method="null-grid"),
        dict(n=31, p=64, m=50, method="null-exact"), dict(n=130, p=129, m=1100, method="alt-grid"), dict(n=79, p=257, m=2100, method="null-exact")]
data = [make_data(n=c["n"], p=c["p"], m=c["m"], seed=10 + i, bxd=(c["n"] == 79)) for i, c in enumerate(cfgs)]
ctx = B.Context(0)
first = [None] * len(cfgs)
bad = 0
t0 = time.time()
for r in range(rounds):
    for i in np.random.default_rng(r).permutation(len(cfgs)):
        Y, G, K, _ = data[i]
        out = B.bulkscan(Y, G, K, method=cfgs[i]["method"], ctx=ctx)
        if first[i] is None:
            first[i] = out["L"].copy()
            assert np.isfinite(first[i]).all()
        elif not np.array_equal(out["L"], first[i]):
            bad += 1
    if r % 20 == 19:
        print(f"  round {r + 1}: {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"soak mixed: {rounds} rounds x {len(cfgs)} configurations, {bad} mismatching results, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
