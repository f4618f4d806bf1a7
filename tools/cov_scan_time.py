"""Null-exact step at the BXD shape WITH null covariates (c = 2, 3 incl. the intercept), device-resident, ms per step and the scan phase:
python tools/cov_scan_time.py [ncov ...]   (BLMM_LR3=0: the two-wave kernel k_scan_lr<C, 1>; default: k_scan_lr3<C>)"""
import sys, time, importlib.util
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); Bn = importlib.util.module_from_spec(spec)
argv = sys.argv; sys.argv = ["x"]; spec.loader.exec_module(Bn); sys.argv = argv
import bulklmm_jl_amd as B
n, p, m = 79, 7321, 35554
Y, G, K = Bn.synth(n, p, m, 20241)
dev = torch.device("cuda", 0)
ctx = B.Context(0, torch.cuda.current_stream().cuda_stream)
dY = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev); dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev); dK = torch.from_numpy(K).to(dev)
dL = torch.empty((m, p), dtype=torch.float64, device=dev); dH = torch.empty((m,), dtype=torch.float64, device=dev)
for ncov in [int(x) for x in sys.argv[1:]] or [1, 2]:
    Cov = np.random.default_rng(5).standard_normal((n, ncov))
    dC = torch.from_numpy(np.ascontiguousarray(Cov.T)).to(dev)
    for _ in range(3): B.bulkscan_dev(ctx, dY, dG, dK, dL, dH, Covar=dC)
    torch.cuda.synchronize(); ctx.set_timing(True); ctx.read_timings()
    t0 = time.perf_counter()
    for _ in range(20): B.bulkscan_dev(ctx, dY, dG, dK, dL, dH, Covar=dC)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    ph, nc = ctx.read_timings(); ctx.set_timing(False)
    print(f"ncov={ncov} (c={ncov + 1}): {dt:.3f} ms per step, scan {ph['scan'] / nc:.3f} ms, finite {bool(torch.isfinite(dL[:64]).all())}")
