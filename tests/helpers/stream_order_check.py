"""Child program of tests/test_gpu_parity.py::test_dev_entry_point_is_ordered_on_torchs_default_stream.
The output is zero-filled on torch's stream immediately before blmm_bulkscan_dev and read back by a torch op immediately
after it, without any explicit synchronisation in between; with a context that really runs on torch's stream the result is
the scan's, every time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bulklmm_jl_amd as blmm  # noqa: E402
from common import make_data  # noqa: E402

Y, G, K, _ = make_data(p=1500, m=3000, seed=3131)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
assert stream.cuda_stream == 0
ctx = blmm.Context(0, stream.cuda_stream)
dY = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev)
dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev)
dK = torch.from_numpy(np.ascontiguousarray(K.T)).to(dev)
dL = torch.full((3000, 1500), -1.0, dtype=torch.float64, device=dev)
dH = torch.empty(3000, dtype=torch.float64, device=dev)
ref = blmm.bulkscan_null(Y, G, K, ctx=blmm.Context(0))
for _ in range(5):
    dL.fill_(-7.0)                                    # torch's stream, right before
    blmm.bulkscan_dev(ctx, dY, dG, dK, dL, dH, method="null-exact")
    got = (dL + 0.0).cpu().numpy().T                  # torch's stream, right after (the .cpu() synchronises)
    assert np.array_equal(got, ref.L), "default stream: the scan was not ordered against torch's work"
s2 = torch.cuda.Stream(device=dev)                    # a non-default torch stream works the same way
ctx2 = blmm.Context(0, s2.cuda_stream)
with torch.cuda.stream(s2):
    dL.fill_(-7.0)
    blmm.bulkscan_dev(ctx2, dY, dG, dK, dL, dH, method="null-exact")
    got = (dL + 0.0).cpu().numpy().T
assert np.array_equal(got, ref.L), "side stream"
# a padded leading dimension (every column on its own 128-byte line; bench.py --ldl-align): same values, padding untouched,
# for every method that writes a p x m matrix
for method in ("null-exact", "null-grid"):
    dense = torch.empty((3000, 1500), dtype=torch.float64, device=dev)
    blmm.bulkscan_dev(ctx, dY, dG, dK, dense, dH, method=method)
    wide = torch.full((3000, 1504), -3.0, dtype=torch.float64, device=dev)
    blmm.bulkscan_dev(ctx, dY, dG, dK, wide[:, :1500], dH, method=method)
    assert torch.equal(wide[:, :1500], dense), method + ": padded ldL"
    assert bool((wide[:, 1500:] == -3.0).all()), method + ": padding written"
print("stream order ok")
