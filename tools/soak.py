"""Stability soak (developer tool, GPU box): many back-to-back device-resident calls; every result must be bit-identical
to the first one (the work lists of the two-kernel Brent and the spin barrier of the multi-workgroup weight basis reorder
work, never results).  python tools/soak.py <n> <p> <m> <calls> [method]"""
import sys, time, importlib.util
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); Bn = importlib.util.module_from_spec(spec)
argv = sys.argv; sys.argv = ["x"]; spec.loader.exec_module(Bn); sys.argv = argv
import bulklmm_jl_amd as B
n, p, m, calls = (int(x) for x in sys.argv[1:5])
method = sys.argv[5] if len(sys.argv) > 5 else "null-exact"
Y, G, K = Bn.synth(n, p, m, 4242)
dev = torch.device("cuda", 0)
dY = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev); dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev)
dK = torch.from_numpy(np.ascontiguousarray(K.T)).to(dev)
dL = torch.empty((m, p), dtype=torch.float64, device=dev); dH = torch.empty((m,), dtype=torch.float64, device=dev)
ctx = B.Context(0, torch.cuda.current_stream().cuda_stream)
grid = [i / 16.0 for i in range(16)] if method != "null-exact" else None
B.bulkscan_dev(ctx, dY, dG, dK, dL, dH, method=method, h2_grid=grid)
torch.cuda.synchronize()
L0, H0 = dL.clone(), dH.clone()
assert torch.isfinite(L0).all()
t0 = time.time(); bad = 0
for i in range(calls):
    dL.zero_(); dH.zero_()
    B.bulkscan_dev(ctx, dY, dG, dK, dL, dH, method=method, h2_grid=grid)
    if i % 50 == 49 or i == calls - 1:
        torch.cuda.synchronize()
        if not (torch.equal(dL, L0) and torch.equal(dH, H0)):
            bad += 1
            print("MISMATCH at call", i, float((dL - L0).abs().max()), flush=True)
        if i % 500 == 499: print(f"  {i + 1} calls, {time.time() - t0:.0f} s", flush=True)
print(f"soak {method} n={n} p={p} m={m}: {calls} calls, {bad} mismatching checkpoints, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
