"""Does capturing ONE bulkscan step into a HIP graph shorten it?  (developer probe, GPU box: the review asked for a graph to remove the
~85 us of inter-kernel gaps.)  The step -- three streams forked and joined with events, ~25 kernels -- is captured with
torch.cuda.graph on the stream the context runs on and replayed; timed against the same number of plain calls.
python tools/graph_probe.py [steps]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); Bn = importlib.util.module_from_spec(spec)
argv = sys.argv; sys.argv = ["x"]; spec.loader.exec_module(Bn); sys.argv = argv
import bulklmm_jl_amd as B
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, p, m = 79, 7321, 35554
Y, G, K = Bn.synth(n, p, m, 4242)
dev = torch.device("cuda", 0)
dY = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev); dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev); dK = torch.from_numpy(np.ascontiguousarray(K.T)).to(dev)
L = torch.empty((m, p), dtype=torch.float64, device=dev); H = torch.empty((m,), dtype=torch.float64, device=dev)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    ctx = B.Context(0, s.cuda_stream)
    for _ in range(5):
        B.bulkscan_dev(ctx, dY, dG, dK, L, H, method="null-exact")
    s.synchronize()
    Lref = L.clone()
    def plain(k):
        t0 = time.perf_counter()
        for _ in range(k):
            B.bulkscan_dev(ctx, dY, dG, dK, L, H, method="null-exact")
        s.synchronize()
        return (time.perf_counter() - t0) / k * 1e3
    print("plain   %.4f ms per step" % plain(steps), flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            B.bulkscan_dev(ctx, dY, dG, dK, L, H, method="null-exact")
        L.zero_(); s.synchronize()
        g.replay(); s.synchronize()
        print("captured; replay reproduces the result:", bool(torch.equal(L, Lref)), flush=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        s.synchronize()
        print("graph   %.4f ms per step" % ((time.perf_counter() - t0) / steps * 1e3), flush=True)
        print("plain   %.4f ms per step (again)" % plain(steps), flush=True)
    except Exception as e:   # noqa: BLE001
        print("capture failed:", repr(e)[:400], flush=True)
