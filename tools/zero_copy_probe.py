"""Host-output experiments (run on the GPU box): (a) rate of one D2H copy of the BXD-sized L into pinned memory, (b) the same as two
/ four concurrent copies on separate streams, (c) the scan writing L STRAIGHT into pinned host memory (zero copy: the _dev entry point
with a pinned tensor as L_out) against scan + copy."""
import sys, time, importlib.util
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); Bn = importlib.util.module_from_spec(spec); sys.argv = ["x"]; spec.loader.exec_module(Bn)
import bulklmm_jl_amd as B
n, p, m = 79, 7321, 35554
Y, G, K = Bn.synth(n, p, m, 20241)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
ctx = B.Context(0, st.cuda_stream if st.cuda_stream else None)
dY = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev); dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev); dK = torch.from_numpy(K).to(dev)
L = torch.empty((m, p), dtype=torch.float64, device=dev); H = torch.empty(m, dtype=torch.float64, device=dev)
Lh = torch.empty((m, p), dtype=torch.float64, pin_memory=True)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
B.bulkscan_dev(ctx, dY, dG, dK, L, H); torch.cuda.synchronize()
gb = L.numel() * 8 / 1e9
a = t(lambda: Lh.copy_(L, non_blocking=True))
print(f"(a) one copy: {a * 1e3:.1f} ms = {gb / a:.1f} GB/s")
for parts in (2, 4):
    ss = [torch.cuda.Stream() for _ in range(parts)]
    ch = (m + parts - 1) // parts
    def multi():
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                Lh[i * ch:(i + 1) * ch].copy_(L[i * ch:(i + 1) * ch], non_blocking=True)
    b = t(multi)
    print(f"(b) {parts} concurrent copies: {b * 1e3:.1f} ms = {gb / b:.1f} GB/s")
c = t(lambda: (B.bulkscan_dev(ctx, dY, dG, dK, L, H), Lh.copy_(L, non_blocking=True)))
print(f"(c) scan + copy: {c * 1e3:.1f} ms")
d = t(lambda: B.bulkscan_dev(ctx, dY, dG, dK, Lh, H))
print(f"(c) scan writing into pinned host memory: {d * 1e3:.1f} ms")
ok = torch.equal(Lh, L.cpu())
print("zero-copy result equals the device result:", ok)
