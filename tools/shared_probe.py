"""A/B of the shared-weights class on the bench workload (step time, class size, equality of the LODs)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, time
    import bulklmm_jl_amd as B
    from bulklmm_jl_amd import api as A, _lib as L
    import bench
    Y, G, K = bench.synth(79, 7321, 35554, 20241)
    ctx = B.Context(0)
    for it in range(3):
        t = time.time()
        Lo, h2, st = A._bulkscan_call(L.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1, "eigen", 0, ctx, return_status=True)
    np.save(sys.argv[1], Lo)
    print("shared", st.lowrank_shared, "fallback", st.lowrank_fallback, "rank", st.lowrank_rank, "resid", st.lowrank_resid, "nan", st.n_nan_lod)
else:
    for v in ("0", "1"):
        env = dict(os.environ, BLMM_DEV_ENV="1", BLMM_LR_SHARED=v)
        subprocess.run([sys.executable, __file__, f"/tmp/L{v}.npy"], env=env, check=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-host-api"], env=env, check=True,
                       stdout=open(f"/tmp/b{v}.json", "w"))
        d = json.loads(open(f"/tmp/b{v}.json").read().strip().splitlines()[-1])
        print("BLMM_LR_SHARED=" + v, "ms_per_step", d["ms_per_step"], "phases", d["phases_ms"])
    import numpy as np
    a, b = np.load("/tmp/L0.npy"), np.load("/tmp/L1.npy")
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-300)
    print("max |dLOD| / |LOD| between the two forms", np.nanmax(np.where(np.abs(a) > 1e-10, rel, 0)), "max abs", np.nanmax(np.abs(a - b)))
