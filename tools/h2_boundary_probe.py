"""Fraction of traits whose null h2 sits at the h2 = 0 boundary on the bench workload (diagnostic for the shared-weights fast path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bulklmm_jl_amd as B
import bench
Y, G, K = bench.synth(79, 7321, 35554, 20241)
ctx = B.Context(0)
r = B.bulkscan_null(Y, G, K, ctx=ctx)
h = r.h2_null_list
lam = np.linalg.eigvalsh(K)
print("lambda max", lam.max(), "min", lam.min())
for t in (1e-15, 1e-14, 1e-13, 1e-12, 1e-9, 1e-6):
    print("h2 <", t, (h < t).mean())
print("h2 > 0.999", (h > 0.999).mean(), "quantiles", np.quantile(h, [0.1, 0.25, 0.5, 0.75, 0.9]))
