"""Child program of tests/test_gpu_configs.py::test_sharded_marker_rotation_equals_the_replicated_one (torch first, then the
library: the other order leaves torch without a HIP device in one process)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bulklmm_jl_amd as blmm  # noqa: E402
from common import make_data  # noqa: E402

n, p, m, R = 500, 20000, 384, 3
Y, G, K, _ = make_data(n=n, p=p, m=m, seed=20242, bxd=False)
dev = torch.device("cuda", 0)
dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev)
dK = torch.from_numpy(np.ascontiguousarray(K.T)).to(dev)
dYf = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev)
grid = [i / 16.0 for i in range(16)]
for method in ("null-exact", "null-grid", "alt-grid"):
    alt = method == "alt-grid"
    ctxs = [blmm.Context(0) for _ in range(R)]
    ref_ctx = blmm.Context(0)
    Lref = torch.empty((m, p), dtype=torch.float64, device=dev)
    Href = torch.empty((m, p) if alt else (m,), dtype=torch.float64, device=dev)
    blmm.bulkscan_dev(ref_ctx, dYf, dG, dK, Lref, Href, method=method, h2_grid=grid)
    ref_ctx.synchronize()
    bc = -(-p // R)
    bld = -(-bc // 128) * 128
    for c in ctxs:
        blmm.prepare_dev(c, dK)
    rows = blmm.rotated_rows(ctxs[0])
    assert rows == 504, rows
    gathered = torch.zeros((R, rows, bld), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()      # the contexts run on private non-blocking streams: torch's fill must have finished before they write
    for r, c in enumerate(ctxs):                         # rank r rotates its block straight into its slot of the gather buffer
        lo, hi = r * bc, min(p, (r + 1) * bc)
        blmm.rotate_block_dev(c, dG[lo:hi], gathered[r])
        c.synchronize()
    for r, c in enumerate(ctxs):
        lo, hi = blmm.trait_shard(m, r, R)
        Lr = torch.empty((hi - lo, p), dtype=torch.float64, device=dev)
        Hr = torch.empty((hi - lo, p) if alt else (hi - lo,), dtype=torch.float64, device=dev)
        Yr = dYf[lo:hi].contiguous()
        torch.cuda.synchronize()
        st = blmm.bulkscan_prerotated_dev(c, Yr, gathered, p, bc, Lr, Hr, method=method, h2_grid=grid, status=True)
        if not torch.equal(Lr, Lref[lo:hi]):
            d = (Lr - Lref[lo:hi])
            bad = (Lr != Lref[lo:hi]).nonzero()
            print("MISMATCH", method, r, "nan in mine", int(torch.isnan(Lr).sum()), "nan in ref", int(torch.isnan(Lref[lo:hi]).sum()), "n differing", bad.shape[0],
                  "first", bad[:5].tolist(), "cols(marker) range", int(bad[:, 1].min()), int(bad[:, 1].max()), "rows(trait) range", int(bad[:, 0].min()), int(bad[:, 0].max()),
                  "h2 equal", bool(torch.equal(Hr, Href[lo:hi])), flush=True)
            raise SystemExit(1)
        assert torch.equal(Hr, Href[lo:hi]), (method, r)
        assert st.n_nan_lod == 0
    for c in ctxs + [ref_ctx]:
        c.close()
    print("sharded rotation ok", method, flush=True)

# ---- the permutation test on the same gathered blocks: fp64 and fp32, the permutations sharded over the ranks --------------
nperms, R = 96, 3
y1 = dYf[0].contiguous()
pidx_all = torch.from_numpy(np.stack([np.random.default_rng(5 + b).permutation(n) for b in range(nperms)]).astype(np.int32)).to(dev)
for dt in (torch.float64, torch.float32):
    ref_ctx = blmm.Context(0)
    # (the gathered blocks are fp64-rotated: the one-call fp32 path is held to them with its own fp32 rotation switched off --
    #  tuning f32_rotation = 0 -- and, below, with it on to the fp32 contract)
    ref_ctx.set_tuning("f32_rotation", 0)
    sc_ref = torch.empty(2, dtype=torch.float64, device=dev); lod_ref = torch.empty(p, dtype=torch.float64, device=dev)
    Lp_ref = torch.empty((nperms, p), dtype=dt, device=dev)
    torch.cuda.synchronize()
    blmm.scan_perms_dev(ref_ctx, y1, dG, dK, sc_ref, lod_ref, Lp_ref, nperms=nperms, perm_idx=pidx_all)
    ref_ctx.synchronize()
    ctxs = [blmm.Context(0) for _ in range(R)]
    for c in ctxs:
        blmm.prepare_dev(c, dK)
    gathered = torch.zeros((R, rows, bld), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for r, c in enumerate(ctxs):
        lo, hi = r * bc, min(p, (r + 1) * bc)
        blmm.rotate_block_dev(c, dG[lo:hi], gathered[r])
        c.synchronize()
    for r, c in enumerate(ctxs):
        lo, hi = blmm.trait_shard(nperms, r, R)
        sc = torch.empty(2, dtype=torch.float64, device=dev); lod = torch.empty(p, dtype=torch.float64, device=dev)
        Lp = torch.empty((hi - lo, p), dtype=dt, device=dev)
        torch.cuda.synchronize()
        blmm.scan_perms_prerotated_dev(c, y1, gathered, p, bc, sc, lod, Lp, nperms=hi - lo, perm_idx=pidx_all[lo:hi].contiguous())
        c.synchronize()
        assert torch.equal(sc, sc_ref) and torch.equal(lod, lod_ref), (dt, r)
        assert torch.equal(Lp, Lp_ref[lo:hi]), (dt, r, float((Lp.double() - Lp_ref[lo:hi].double()).abs().max()))
    if dt == torch.float32:
        ref_ctx.set_tuning("f32_rotation", 1)
        sc2 = torch.empty(2, dtype=torch.float64, device=dev); lod2 = torch.empty(p, dtype=torch.float64, device=dev)
        Lp2 = torch.empty((nperms, p), dtype=dt, device=dev)
        torch.cuda.synchronize()
        blmm.scan_perms_dev(ref_ctx, y1, dG, dK, sc2, lod2, Lp2, nperms=nperms, perm_idx=pidx_all)
        ref_ctx.synchronize()
        assert torch.equal(sc2, sc_ref)
        assert bool(((lod2 - lod_ref).abs() <= 1e-6 * lod_ref.abs() + 1e-10).all()), float((lod2 - lod_ref).abs().max())
        d = (Lp2.double() - Lp_ref.double()).abs()
        assert bool((d <= 1e-3 * Lp_ref.double().abs() + 1e-4).all()), float(d.max())
    for c in ctxs + [ref_ctx]:
        c.close()
    print("sharded rotation ok perms", dt, flush=True)
