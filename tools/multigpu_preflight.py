#!/usr/bin/env python3
"""Pre-flight for the first run on a multi-GPU node (no such node has been available to any round of this build: everything here has
only ever run with R = 1 or with ranks sharing one device).  Walks, for R = 2 .. N visible GPUs, what the N > 1 paths rely on and
prints one PASS / FAIL line per step, so that a failure can be attributed before bench.py is blamed:

  1. behind the C ABI, one process: blmm_bulkscan_multi on R DISTINCT devices -- host_shards (R PCIe links), none, allgather through
     RCCL (ncclCommInitAll + grouped ncclAllGather, in place) and through direct hipMemcpyPeerAsync copies -- each against the
     single-GPU result bit for bit, then the in-place consumers (blmm_multi_last_colmax / _lod_threshold);
  2. one process per GPU (bench.py's path): torch.distributed over RCCL -- all_gather_into_tensor of the LOD slots
     (`gather_verified` in the bench line) at n = 79, and the sharded marker rotation with its in-step all-gather at n = 300.

    python tools/multigpu_preflight.py [--max-gpus N]

Run it on the node itself; it needs nothing but this repository (built) and the image's torch.  BLMM_MULTI_LOG=1 makes the library
say on stderr which gather branch ran."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def step(name, ok, detail=""):
    print(("PASS " if ok else "FAIL ") + name + ((": " + detail) if detail else ""), flush=True)
    return ok


def abi_leg(R):
    """Runs in a child process (its own HIP state): blmm_bulkscan_multi on devices 0 .. R-1."""
    import numpy as np
    import bulklmm_jl_amd as B
    from common import make_data
    Y, G, K, _ = make_data(p=500, m=64 * R + 3, seed=77)          # ragged: the last shard is short
    one = B.bulkscan(Y, G, K, method="null-exact")
    arg = np.argmax(one["L"], axis=0)
    good = True
    for gather, env in (("host_shards", None), ("none", None), ("allgather", "rccl"), ("allgather", "peer")):
        if env:
            os.environ["BLMM_DEV_ENV"] = "1"; os.environ["BLMM_ALLGATHER"] = env
        mc = B.MultiContext(list(range(R)))
        try:
            r = B.bulkscan_multi(mc, Y, G, K, method="null-exact", gather=gather)
            same = bool(np.array_equal(r["L"], one["L"]) and np.array_equal(r["h2_null_list"], one["h2_null_list"]))
            mx, ax = mc.last_colmax()
            same = same and bool(np.array_equal(ax, arg) and np.array_equal(mx, one["L"][arg, np.arange(Y.shape[1])]))
            good = step(f"C ABI, {R} devices, gather {gather}" + (f" via {env}" if env else ""), same) and good
        except Exception as e:   # noqa: BLE001
            good = step(f"C ABI, {R} devices, gather {gather}" + (f" via {env}" if env else ""), False, repr(e)) and good
        finally:
            mc.close()
            os.environ.pop("BLMM_ALLGATHER", None)
    return good


def bench_leg(R, n, p, m, label):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(R), "--steps", "3", "--warmup", "1", "--n", str(n), "--p", str(p),
           "--m", str(m), "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    if run.returncode != 0:
        return step(label, False, (run.stderr or run.stdout)[-600:])
    line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    mg = j.get("multi_gpu") or {}
    ok = mg.get("ranks_seen") == list(range(R)) and mg.get("distinct_devices") == R and mg.get("gather_verified") is True and j["output_finite"]
    return step(label, ok, f"ms_per_step {j['ms_per_step']:.3f}, allgather_ms {j['allgather_ms']}, gather_verified {mg.get('gather_verified')}, "
                           f"devices {[d['pci_bus_id'] for d in mg.get('devices', [])]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-gpus", type=int, default=8)
    ap.add_argument("--abi-child", type=int, default=0)
    a = ap.parse_args()
    if a.abi_child:
        raise SystemExit(0 if abi_leg(a.abi_child) else 1)
    import torch          # device_count() does not initialise the GPU in this process
    ndev = min(torch.cuda.device_count(), a.max_gpus)
    step("visible GPUs", ndev >= 2, str(ndev))
    good = True
    for R in range(2, ndev + 1):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--abi-child", str(R)], timeout=1800)
        good = (r.returncode == 0) and good
        good = bench_leg(R, 79, 7321, 35554, f"bench.py --gpus {R}, BXD shape: RCCL all-gather of the LOD slots") and good
        good = bench_leg(R, 300, 4000, 64 * R, f"bench.py --gpus {R}, n = 300: sharded marker rotation + in-step all-gather") and good
    print("preflight " + ("OK" if good and ndev >= 2 else "NOT OK"))
    raise SystemExit(0 if good and ndev >= 2 else 1)


if __name__ == "__main__":
    main()
