"""scan(...; assumption = "alt") at BXD size: wall time of the host-pointer call, and the NumPy oracle on a marker sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bulklmm_jl_amd as B
from oracle import bulklmm_oracle as O
import bench
for (n, p) in ((79, 7321), (500, 50000)):
    Y, G, K = bench.synth(n, p, 8, 20241)
    ctx = B.Context(0)
    y = Y[:, [1]]
    B.scan(y, G, K, assumption="alt", ctx=ctx)
    t = time.time()
    for _ in range(5):
        r = B.scan(y, G, K, assumption="alt", ctx=ctx)
    dt = (time.time() - t) / 5
    print(f"n={n} p={p}: scan_alt {dt * 1e3:.2f} ms per call (host pointers), h2_null {r['h2_null']:.4f}, max lod {r['lod'].max():.3f}")
    if n == 79:
        t = time.time()
        ref = O.scan(y, G[:, :200], K, assumption="alt")
        dt_o = (time.time() - t) / 200
        print(f"   oracle (NumPy, 1 thread): {dt_o * 1e3:.2f} ms per marker -> {dt_o * p:.1f} s for {p} markers; max |dlod| on the sample {np.abs(ref['lod'] - r['lod'][:200]).max():.2e}")
