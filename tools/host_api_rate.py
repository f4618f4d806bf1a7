"""PCIe-inclusive rate of the host-pointer API (blmm_bulkscan): host Y/G/K in, host L out.  Run on the GPU box."""
import sys, time, importlib.util
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
spec = importlib.util.spec_from_file_location("bench", "bench.py"); Bn = importlib.util.module_from_spec(spec); sys.argv = ["x"]; spec.loader.exec_module(Bn)
import bulklmm_jl_amd as B
Y, G, K = Bn.synth(79, 7321, 35554, 20241)
for name, fn in (("null-exact", lambda: B.bulkscan_null(Y, G, K)), ("null-grid", lambda: B.bulkscan_null_grid(Y, G, K, [i / 16 for i in range(16)]))):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"{name}: {t * 1e3:.1f} ms host-to-host, {7321 * 35554 / t:.3e} tests/s, L = {r.L.nbytes / 1e9:.2f} GB -> {r.L.nbytes / t / 1e9:.1f} GB/s effective")
