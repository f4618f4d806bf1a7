#!/usr/bin/env python3
"""bench.py -- headline benchmark of the bulkscan hot path on MI355X.

Metric (BASELINE.json): (trait x marker) LOD tests per second.  One "step" = one complete bulkscan_null
(null-exact) pass -- device eigen-decomposition of K, rotation of Y and G, per-trait Brent h2, A-side
panels, the f64-MFMA LOD kernel -- over a BXD-shaped synthetic batch (n=79, p=7321, m=35554, fp64;
BASELINE.json configs[1]) whose inputs are already resident in HBM; the p x m LOD matrix stays in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Without WORLD_SIZE in the environment `--gpus N` starts the N ranks itself
(one child process per GPU with the torchrun environment, BEFORE anything touches the GPU in this one).  The headline for N > 1 is
STRONG scaling of the ONE BXD-shaped problem (BASELINE.json's metric: "full BXD-shaped bulkscan wall-time at 1/2/4/8
GPUs"): rank r scans the trait block sharding.trait_shard(m, r, N); no data-path collective in the timed region.  The
north_star's optional final step -- an RCCL all-gather of the LOD column blocks over xGMI -- is timed separately
(`allgather_ms`) and a second timed loop with the gather inside the step gives `gathered_ms_per_step`; `weak` holds the
weak-scaling figure (every rank scans a full BXD-shaped batch) measured in the same run.  `--scaling weak` makes that
one the headline instead.  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD MI355X datasheet, FP64 matrix (dense); measured sustained: see DESIGN.md
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 matrix, v_mfma_f32_32x32x2_f32 (exact fp32)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth(n, p, m, seed):
    from common import make_data
    Y, G, K, _ = make_data(n=n, p=p, m=m, seed=seed, bxd=(n == 79))
    return Y, G, K


def cpu_baseline(Y, G, K, budget_s=15.0):
    """The C/OpenMP restatement of the reference's null-exact loop (oracle/bulkscan_null_ref.c: per trait Brent over the
    QR-based wls, re-weighting and QR-residualising the whole rotated marker matrix, src/bulkscan.jl:268-286 and
    src/bulkscan_helpers.jl:127-150), traits spread over OpenMP threads the way the reference spreads trait blocks over
    Julia threads, on a bounded sample of the same workload.  It is the checker used as the timed CPU stand-in (the
    reference is pure Julia and cannot run here); it is never on the product path.
    `value` uses EVERY core the box grants this process (north_star: "the GPU box's own host cores, core count stated");
    `threads16` is the same sample on 16 threads, the count of the reference's own published run (README.md:322-332)."""
    from oracle import cref   # test infrastructure, used here only as the timed CPU baseline
    p = G.shape[1]
    # every core the box GRANTS this process: the scheduler affinity, capped by the cgroup's CPU quota (a one-GPU box of this pool
    # shows 256 CPUs in its affinity mask and a cpu.max of 16 CPUs: 256 OpenMP threads then time-share 16 cores and run at half the
    # 16-thread rate -- measured in round 4)
    granted = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota, cg = None, None
    try:
        cg = open("/sys/fs/cgroup/cpu.max").read().strip()
        q, per = cg.split()
        if q != "max":
            quota = max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cref.load().blmm_ref_max_threads(), granted, quota or granted))

    def run(threads, budget):
        probe = min(Y.shape[1], 4 * threads)
        cref.bulkscan_null(Y[:, :probe], G, K, nthreads=threads)          # warm-up (thread pool, page faults)
        t0 = time.perf_counter()
        cref.bulkscan_null(Y[:, :probe], G, K, nthreads=threads)
        per_trait = (time.perf_counter() - t0) / probe
        sample = int(min(Y.shape[1], max(probe, budget / max(per_trait, 1e-6))))
        t0 = time.perf_counter()
        cref.bulkscan_null(Y[:, :sample], G, K, nthreads=threads)
        dt = time.perf_counter() - t0
        return p * sample / dt, sample, dt

    v, sample, dt = run(cores, budget_s)
    out = {"value": v, "unit": "tests/s", "cores": cores, "cores_in_affinity_mask": granted, "cgroup_cpu_max": cg, "kind": "port",
           "sample": f"bulkscan_null (oracle/bulkscan_null_ref.c, C + OpenMP, eigen + rotation + per-trait Brent + scan) on the "
                     f"first {sample} of {Y.shape[1]} traits x {p} markers, {cores} threads, {dt:.1f} s"}
    if cores != 16 and granted >= 16:
        v16, s16, d16 = run(16, min(budget_s, 8.0))
        out["threads16"] = {"value": v16, "cores": 16, "sample": f"first {s16} traits, 16 threads, {d16:.1f} s (the reference's published run used 16 Julia threads)"}
    return out


def workload_name(a, n, p, m_total, m_local, f32, world):
    """Names the configuration actually run: method, shape, dtype and the BASELINE.json config it corresponds to (if any)."""
    dt = "fp32 permutation matrix (null model fp64)" if f32 else "fp64"
    shape = (n, p, m_total)
    if a.method == "null-exact" and shape == (79, 7321, 35554):
        which = "BASELINE.json configs[1]: bulkscan_null, BXD shape"
    elif a.method in ("null-grid", "alt-grid") and shape == (79, 7321, 35554):
        which = "BASELINE.json configs[3]: 16-point h2 grid, BXD shape (" + ("primary" if a.method == "null-grid" else "secondary") + " method)"
    elif a.method in ("null-exact", "null-grid") and (n, p) == (500, 50000):
        which = f"BASELINE.json configs[2]: synthetic large eQTL ({m_total} of its 20000 traits" + (", one of 8 shards" if m_total == 2500 else "") + ")"
    elif a.method == "perms" and (n, p) == (1000, 100000):
        which = f"BASELINE.json configs[4]: single-trait permutation test ({m_total} of its 10000 permutations" + (", one of 8 shards" if m_total == 1250 else "") + ")"
    else:
        which = "no BASELINE.json config has this method and shape"
    unit = "permutations" if a.method == "perms" else "traits"
    pv = " + output_pvals (-log10 p matrix written by the same step)" if getattr(a, "pvals", False) and a.method != "perms" else ""
    if getattr(a, "reduced", False):
        pv += " REDUCED OUTPUT (blmm_bulkscan_reduced_dev: per-trait peaks + LOD > 5 triplets, no L matrix; each call synchronises)"
    return (f"method={a.method} n={n} p={p} m={m_total} {dt}{pv}, {which}; {m_local} {unit} on rank 0 of {world}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=79)
    ap.add_argument("--p", type=int, default=7321)
    ap.add_argument("--m", type=int, default=35554)
    ap.add_argument("--method", default="null-exact", choices=["null-exact", "null-grid", "alt-grid", "perms"],
                    help="perms: scan(...; permutation_test=true) with --m permutations of ONE trait (BASELINE.json configs[4])")
    ap.add_argument("--perm-dtype", default="f64", choices=["f64", "f32"],
                    help="--method perms: precision of the permutation LOD matrix (f32 = BASELINE.json configs[4])")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue consecutive (independent) steps round-robin on this many contexts/streams; 1 = every step "
                         "is one complete bulkscan wall-time (the reported default), > 1 = pipelined throughput")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1 only; default strong (one BXD problem sharded over the ranks)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the end-to-end (host pointers in, host L out) timing")
    ap.add_argument("--gather", action="store_true", help="put the RCCL all-gather of the LOD shards inside the step")
    ap.add_argument("--ldl-align", type=int, default=1,
                    help="leading dimension of the device LOD matrix rounded up to this many doubles (1 = dense, ld = p, the "
                         "reference's layout and the default; 16 = every column starts on a 128-byte line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pvals", action="store_true",
                    help="`output_pvals = true`: the step also produces the -log10 p matrix (blmm_set_log10p_output; --pvals-pass runs it as "
                         "the column pass over the finished L instead of from the scan epilogues: tuning key pval_fused = 0)")
    ap.add_argument("--pvals-pass", action="store_true")
    ap.add_argument("--reduced", action="store_true",
                    help="the step is blmm_bulkscan_reduced_dev: per-trait peak LOD + marker and the LOD > 5 triplets out of the scan "
                         "kernels' epilogues, the p x m matrix is never written (NOT the reported default; null-exact / null-grid, N = 1)")
    ap.add_argument("--no-all-rank-form", action="store_true",
                    help="skip the extra timed loop with every trait in the rank-R form (profiling runs: its launches would be averaged into the kernel statistics)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    a = ap.parse_args()
    if a.reduced and (a.method not in ("null-exact", "null-grid") or a.gpus > 1 or a.streams > 1):
        raise SystemExit("--reduced: null-exact / null-grid on one GPU, one stream")

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # start the ranks ourselves, one child process per GPU with the torchrun environment (RANK / LOCAL_RANK /
        # WORLD_SIZE / MASTER_*); nothing in this process has touched the GPU (no torch import, no HIP call), and it only
        # waits for them.  (torch.distributed.run would do, but its argument parser claims bench.py's short options.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        while procs:          # a rank that dies (e.g. fewer GPUs than ranks) must not leave the others waiting at the rendezvous
            time.sleep(0.2)
            for pr in list(procs):
                r = pr.poll()
                if r is None:
                    continue
                procs.remove(pr)
                if r != 0:
                    rc = rc or r
                    for other in procs:
                        other.terminate()          # exactly the children started above
        raise SystemExit(rc)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if a.scaling is None:
        a.scaling = "strong" if world > 1 else "weak"   # N = 1: the two coincide

    n, p = a.n, a.p
    # ONE problem, generated identically on every rank (fixed seed); rank r owns the trait block trait_shard(m, r, N)
    Yf, G, K = synth(n, p, a.m, 20240 + 1)
    spec = __import__("importlib.util").util.spec_from_file_location("blmm_sharding", os.path.join(ROOT, "bulklmm.jl_amd", "sharding.py"))
    sharding = __import__("importlib.util").util.module_from_spec(spec)
    spec.loader.exec_module(sharding)          # pure Python: importing it does not touch the GPU
    lo, hi = sharding.trait_shard(a.m, rank, world)
    sizes = sharding.shard_sizes(a.m, world)
    mx = max(sizes)
    if a.scaling == "weak":
        Y = Yf; m_local = a.m; m_total = a.m * world
    else:
        Y = np.ascontiguousarray(Yf[:, lo:hi]); m_local = hi - lo; m_total = a.m

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.method == "null-exact":
        cpu = cpu_baseline(Y, G, K, a.cpu_budget)  # before the GPU is initialised in this process

    import torch
    import torch.distributed as dist
    import bulklmm_jl_amd as B

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the bulkscan path has no CPU fallback)")
    # BLMM_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share
    # devices, the collectives go through host tensors, the all-gather is skipped); the measured path is always nccl.
    backend = os.environ.get("BLMM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible (one rank per GPU; BLMM_BENCH_BACKEND=gloo rehearses "
                         "the control flow with ranks sharing devices)" % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    alt = a.method == "alt-grid"
    perms = a.method == "perms"
    f32 = perms and a.perm_dtype == "f32"
    ldt = torch.float32 if f32 else torch.float64
    # column-major device operands: a (cols, rows) contiguous tensor IS the rows x cols column-major matrix
    dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev)
    dK = torch.from_numpy(np.ascontiguousarray(K.T)).to(dev)
    grid = [i / 16.0 for i in range(16)] if a.method in ("null-grid", "alt-grid") else None
    stream = torch.cuda.current_stream()
    ctx = B.Context(dev_index, stream.cuda_stream)   # torch's current stream (handle 0 = the legacy default stream is adopted as such)
    if a.pvals_pass:
        ctx.set_tuning("pval_fused", 0)

    class Work:
        """One rank's operands and outputs for a trait block Ycols (n x mloc)."""
        def __init__(self, Ycols, slot_cols=None):
            self.m = Ycols.shape[1]
            self.dY = torch.from_numpy(np.ascontiguousarray(Ycols.T)).to(dev)
            # gathered layout: `world` slots of slot_cols columns each (the largest shard: ragged shards leave pad columns
            # at the end of their slot); this rank scans straight into its slot, the all-gather is in place
            self.slot = slot_cols
            if world > 1 and slot_cols is not None:
                self.full = torch.empty((world, slot_cols, p), dtype=ldt, device=dev)
                self.dL = self.full[rank][: self.m]
            else:
                self.full = None
                ldl = -(-p // a.ldl_align) * a.ldl_align
                self.dL = torch.empty((self.m, ldl), dtype=ldt, device=dev)[:, :p]
            self.dH = torch.empty((self.m, p) if alt else (max(self.m, 1),), dtype=torch.float64, device=dev)
            if a.reduced:
                self.rmax = torch.empty(max(self.m, 1), dtype=torch.float64, device=dev)
                self.rarg = torch.empty(max(self.m, 1), dtype=torch.int64, device=dev)
                self.ti = torch.empty(1 << 21, dtype=torch.int32, device=dev); self.tj = torch.empty(1 << 21, dtype=torch.int32, device=dev)
                self.tl = torch.empty(1 << 21, dtype=torch.float64, device=dev); self.tc = torch.zeros(1, dtype=torch.int64, device=dev)
            self.dP = torch.empty((self.m, p), dtype=torch.float64, device=dev) if a.pvals and not perms else None
            if perms:
                self.dy1 = self.dY[0].contiguous()
                self.dsc = torch.empty(2, dtype=torch.float64, device=dev)
                self.dlod = torch.empty(p, dtype=torch.float64, device=dev)
            # N > 1 and n >= 256: the marker rotation is SHARDED over the ranks (blmm_prepare_dev / blmm_rotate_block_dev, an
            # all-gather of the k-major blocks over xGMI, blmm_bulkscan_prerotated_dev) instead of repeated by every rank; at
            # n = 79 the whole rotation is 15 us and the replicated form is the faster one.  This all-gather is a data-path
            # collective INSIDE the step.
            self.shard_rot = shard_rotation
            if self.shard_rot:
                self.bc = -(-p // world)
                rows = -(-n // 8) * 8
                self.gx = torch.zeros((world, rows, -(-self.bc // 128) * 128), dtype=torch.float64, device=dev)
                self.gblk = dG[rank * self.bc: min(p, (rank + 1) * self.bc)]

        def scan(self, c=None, L=None, H=None):
            c = c or ctx; L = self.dL if L is None else L; H = self.dH if H is None else H
            if self.shard_rot:
                B.prepare_dev(c, dK)
                if self.gblk.shape[0] > 0:
                    B.rotate_block_dev(c, self.gblk, self.gx[rank])
                if backend == "nccl":
                    dist.all_gather_into_tensor(self.gx.view(-1), self.gx[rank].reshape(-1))
                else:
                    c.synchronize()
                    mine = self.gx[rank].cpu()
                    parts = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(parts, mine)
                    self.gx.copy_(torch.stack(parts))
                    torch.cuda.synchronize()
                if perms:
                    B.scan_perms_prerotated_dev(c, self.dy1, self.gx, p, self.bc, self.dsc, self.dlod, L, nperms=self.m, seed=1 + rank)
                else:
                    B.bulkscan_prerotated_dev(c, self.dY, self.gx, p, self.bc, L, H, method=a.method, h2_grid=grid)
            elif perms:
                B.scan_perms_dev(c, self.dy1, dG, dK, self.dsc, self.dlod, L, nperms=self.m, seed=1 + rank)
            elif a.reduced:
                B.bulkscan_reduced_dev(c, self.dY, dG, dK, self.rmax, self.rarg, H, method=a.method, h2_grid=grid, threshold=5.0,
                                       trip_i=self.ti, trip_j=self.tj, trip_lod=self.tl, trip_count=self.tc)
            else:
                B.bulkscan_dev(c, self.dY, dG, dK, L, H, method=a.method, h2_grid=grid, log10p_out=self.dP)

        def gather(self):
            if self.full is not None and backend == "nccl":
                dist.all_gather_into_tensor(self.full.view(-1), self.full[rank].reshape(-1))

    # (BLMM_BENCH_SHARD_ROT=1: the sharded form under gloo as well, the gather bounced through the host -- the one-GPU rehearsal
    # of this code path in tests/test_gpu_parity.py; over RCCL it is the default from n = 256)
    shard_rotation = world > 1 and n >= 256 and a.streams == 1 and (backend == "nccl" or os.environ.get("BLMM_BENCH_SHARD_ROT") == "1")
    work = Work(Y, mx if a.scaling == "strong" else a.m)

    # --streams S > 1: S independent contexts (own stream, own workspace, own outputs); step i runs on context i % S
    extra = []
    for _ in range(max(a.streams, 1) - 1):
        st_i = torch.cuda.Stream(device=dev)
        extra.append((B.Context(dev_index, st_i.cuda_stream), torch.empty_like(work.dL), torch.empty_like(work.dH)))
    step_no = [0]

    def step(w, gather):
        k = step_no[0] % (1 + len(extra))
        step_no[0] += 1
        if k == 0:
            w.scan()
        else:
            w.scan(*extra[k - 1])
        if gather:
            w.gather()

    def timed(w, gather, steps, warmup, marks=True):
        """`steps` steps bracketed by barrier + synchronize on both sides; the MAX over ranks of the wall time.  marks: the library's
        phase timings (HIP events between the phases, on its streams) are on -- what `phases_ms` and `roofline` are read from."""
        for _ in range(max(warmup, 1)):
            step(w, gather)
        torch.cuda.synchronize()
        ctx.set_timing(marks and not os.environ.get("BLMM_BENCH_NOTIMING"))
        ctx.read_timings()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(w, gather)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        ph, nc = ctx.read_timings()
        ctx.set_timing(False)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), ph, nc

    # The end-to-end section (its own context, ~0.5 s of complete host-to-host calls) runs BEFORE the timed loop: a device fresh out
    # of its idle power state scans ~4 % slower for its first ~50 steps (tools/warmup_sweep.sh: 1.65 ms per step behind 5 warm-up
    # steps, 1.61 behind 50 or 800; the eigen phase does not move), and `value` is meant to be the rate a long job sees.  The order
    # is reported in the line (`sections_before_timed_loop`).
    host_api = None
    if rank == 0 and world == 1 and not a.no_host_api and not perms and a.streams == 1 and not a.reduced:
        # SURVEY.md §8(d)'s third time: host inputs -> host L through the drop-in entry point (H2D of Y/G/K, the same
        # kernels, D2H of L into caller memory); once into a fresh pageable array, once into a pinned (registered) one
        Lh = np.empty((p, m_local), order="F")
        Lh[:] = 0.0                                    # touch the pages: the caller's allocation cost is not ours
        meth = {"null-exact": B._lib.BLMM_NULL_EXACT, "null-grid": B._lib.BLMM_NULL_GRID, "alt-grid": B._lib.BLMM_ALT_GRID}[a.method]
        hctx = B.Context(dev_index)
        # column-major inputs, as a Julia caller's Arrays are: the Python mirror would otherwise re-lay out the C-ordered synthetic Y
        # on every call (np.asfortranarray of 22 MB, ~4 ms of NumPy inside the timed call -- not part of the C ABI's time)
        Yf_, Gf_, Kf_ = np.asfortranarray(Y), np.asfortranarray(G), np.asfortranarray(K)
        def host_call(out):
            t0 = time.perf_counter()
            B.api.bulkscan_into(hctx, meth, Yf_, Gf_, Kf_, out, h2_grid=grid)
            return (time.perf_counter() - t0) * 1e3
        host_call(Lh)
        t_page = min(host_call(Lh) for _ in range(2))
        t_pin = None
        try:
            B.api.host_register(Lh)
            host_call(Lh)
            t_pin = min(host_call(Lh) for _ in range(2))
            B.api.host_unregister(Lh)
        except Exception as e:   # noqa: BLE001
            t_pin = None
        # ... and what the caller pays when L does not have to cross PCIe: (a) L_out == NULL -- the matrix stays in
        # HBM -- followed by the per-trait peaks and the LOD > 5 triplets from the resident matrix; (b) blmm_bulkscan_reduced -- the
        # same results out of the scan kernels' epilogues, L never written
        t_keep = t_red = red_route = None
        n_trip = None
        same = None
        if a.method in ("null-exact", "null-grid"):
            # the C ABI calls themselves, into buffers the caller already has (as for the L_out timings above: the Python mirror's
            # own allocations and its sort of the triplets are not the library's time)
            import ctypes as C
            cap = 1 << 21
            lib = hctx.lib
            o = B.api._opts(meth, False, True, "eigen", 1, 1.0, 0.0)
            gridv = None if grid is None else np.ascontiguousarray(np.asarray(grid, dtype=np.float64))
            ng = 0 if grid is None else len(grid)
            pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)   # noqa: E731
            mxk = np.empty(m_local); axk = np.empty(m_local, dtype=np.int64); h2k = np.empty(m_local)
            ik = np.empty(cap, dtype=np.int32); jk = np.empty(cap, dtype=np.int32); lk = np.empty(cap); ck = C.c_int64(0)
            mxr = np.empty(m_local); axr = np.empty(m_local, dtype=np.int64); h2r = np.empty(m_local)
            ir = np.empty(cap, dtype=np.int32); jr = np.empty(cap, dtype=np.int32); lr_ = np.empty(cap); cr = C.c_int64(0)
            for arr in (ik, jk, lk, ir, jr, lr_):
                arr[:] = 0                           # touch the pages
            def keep_call():
                t0 = time.perf_counter()
                hctx.check(lib.blmm_bulkscan(hctx.h, C.byref(o), pp(Yf_), n, m_local, pp(Gf_), p, None, 0, pp(Kf_), None, pp(gridv), ng, None, pp(h2k), None))
                hctx.check(lib.blmm_last_lod_colmax(hctx.h, pp(mxk), pp(axk)))
                hctx.check(lib.blmm_last_lod_threshold(hctx.h, 5.0, cap, pp(ik), pp(jk), pp(lk), C.byref(ck)))
                return (time.perf_counter() - t0) * 1e3
            red = B._lib.blmm_reduced(mxr.ctypes.data, axr.ctypes.data, 1, 5.0, cap, ir.ctypes.data, jr.ctypes.data, lr_.ctypes.data, C.addressof(cr))
            def red_call():
                t0 = time.perf_counter()
                hctx.check(lib.blmm_bulkscan_reduced(hctx.h, C.byref(o), pp(Yf_), n, m_local, pp(Gf_), p, None, 0, pp(Kf_), None, pp(gridv), ng, C.byref(red), pp(h2r), None))
                return (time.perf_counter() - t0) * 1e3
            keep_call()
            t_keep = min(keep_call() for _ in range(4))
            red_call()
            t_red = min(red_call() for _ in range(4))
            red_route = int(lib.blmm_last_reduced_route(hctx.h))
            n_trip = int(cr.value)
            kk, kr = int(ck.value), int(cr.value)
            ok_ = np.lexsort((ik[:kk], jk[:kk])); or_ = np.lexsort((ir[:kr], jr[:kr]))
            same = bool(kk == kr and np.array_equal(mxk, mxr) and np.array_equal(axk, axr) and np.array_equal(h2k, h2r)
                        and np.array_equal(ik[:kk][ok_], ir[:kr][or_]) and np.array_equal(jk[:kk][ok_], jr[:kr][or_])
                        and np.array_equal(lk[:kk][ok_], lr_[:kr][or_]))
        host_api = {"end_to_end_ms_pageable_out": t_page, "end_to_end_ms_pinned_out": t_pin,
                    "end_to_end_ms_keep_on_device": t_keep, "end_to_end_ms_reduced_out": t_red,
                    "reduced_route": red_route, "reduced_triplets_lod_gt_5": n_trip,
                    "reduced_equals_keep_on_device": same,
                    "tests_per_s_end_to_end": p * m_local / ((t_pin or t_page) * 1e-3),
                    "tests_per_s_end_to_end_reduced_out": (p * m_local / (t_red * 1e-3)) if t_red else None,
                    "note": "host Y/G/K in; *_out: host L out (2.08 GB over PCIe at BXD size); keep_on_device: L stays in HBM "
                            "(blmm_bulkscan, L_out == NULL) + per-trait peaks + LOD > 5 triplets from the resident matrix; reduced_out: the "
                            "same results from the scan epilogues, L never written (blmm_bulkscan_reduced); never `value`"}
        hctx.close()
    # Auxiliary loops FIRST, the contract's timed loop last (for the reason given above: by then the device has been scanning for
    # ~0.1 s; all of them are listed in `sections_before_timed_loop`).
    # The headline workload's h2 = 0 share (about half of the synthetic traits end at the boundary and take the cheaper
    # shared-weights class) is a property of the DATA: one more timed loop with the class switched off (every trait through the
    # rank-R form: tuning key lr_shared = 0) says what the step costs without it.
    all_rank = None
    if a.method == "null-exact" and world == 1 and a.streams == 1 and not a.no_all_rank_form and not a.reduced:
        ctx.set_tuning("lr_shared", 0)
        try:
            dt_r, ph_r, nc_r = timed(work, False, max(a.steps // 2, 3), 1)
            all_rank = {"ms_per_step": dt_r / max(a.steps // 2, 3) * 1e3, "scan_ms": ph_r["scan"] / max(nc_r, 1)}
        finally:
            ctx.set_tuning("lr_shared", 1)
        work.scan(); torch.cuda.synchronize()      # leave the default path's result in the outputs
    # ... and the same loop as a caller runs it by default -- no phase timings: every event record between two dependent kernels of
    # the library's main stream is a marker packet the next kernel waits for (profiles/r04_timeline_*timing*.txt)
    dt_nomarks = None
    if world == 1 and a.streams == 1:
        dt_nomarks, _, _ = timed(work, a.gather, a.steps, 1, marks=False)
    dt, phases, ncalls = timed(work, a.gather, a.steps, a.warmup)
    lr_rank = lr_resid = lr_fallback = lr_shared = lr_profile = None
    if a.method == "null-exact":   # one extra (untimed) call with a status read-back: the weight basis and its guard
        st = B.bulkscan_dev(ctx, work.dY, dG, dK, work.dL, work.dH, method=a.method, h2_grid=grid, status=True)
        lr_rank, lr_resid, lr_fallback = int(st.lowrank_rank), float(st.lowrank_resid), int(st.lowrank_fallback)
        lr_shared = int(st.lowrank_shared)
        lr_profile = ctx.lowrank_profile()          # (shared-weights traits, [(traits, rank) per segment of the heritability axis])
    # sanity: the output must be finite (a fast kernel with wrong results is not a result)
    chk = torch.isfinite(work.rmax).all().item() if a.reduced else torch.isfinite(work.dL[: min(64, work.m)]).all().item()

    ag_ms = gathered_ms = None
    weak = None
    multi = None
    if world > 1:
        # ---- who is here: every rank reports (rank, local device index, device name / UUID / PCI bus id, its trait count), so that
        #      the first run on a real node can be read from the JSON line alone
        pr = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
              "pci_bus_id": getattr(pr, "pci_bus_id", None), "m_local": int(work.m), "cols": [int(lo), int(hi)] if a.scaling == "strong" else [0, int(a.m)]}
        seen = [None] * world
        dist.all_gather_object(seen, me)
        multi = {"ranks_seen": sorted(x["rank"] for x in seen), "devices": seen, "backend": backend,
                 "rccl_world": dist.get_world_size(), "distinct_devices": len({(x["uuid"], x["pci_bus_id"], x["device_index"]) for x in seen}),
                 "rccl_version": None,
                 "marker_rotation_sharded": bool(shard_rotation)}
        try:
            multi["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:   # noqa: BLE001
            pass
        # ---- is the gathered matrix the single-GPU matrix?  After one all-gather every rank recomputes 8 columns that OTHER ranks
        #      own (a trait's column does not depend on which other traits are in the call: tests/test_gpu_fullsize.py's shard
        #      consistency) and compares them with its gathered copy bit for bit
        if backend == "nccl" and work.full is not None and not perms:
            work.scan(); work.gather(); torch.cuda.synchronize(); barrier()
            picks = []
            for k in range(8):
                s_ = (rank + 1 + k) % world
                if s_ == rank or sizes[s_] == 0:
                    continue
                j_ = (17 * k + 5 * rank) % (sizes[s_] if a.scaling == "strong" else a.m)
                picks.append((s_, j_, (sharding.trait_shard(a.m, s_, world)[0] + j_) if a.scaling == "strong" else j_))
            ok = True
            if picks:
                w8 = Work(np.ascontiguousarray(Yf[:, [g for _, _, g in picks]]), None)
                w8.scan(); torch.cuda.synchronize()
                for k, (s_, j_, _) in enumerate(picks):
                    ok = ok and bool(torch.equal(work.full[s_][j_], w8.dL[k]))
                del w8
            okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            multi["gather_verified"] = bool(okt.item())
            multi["gather_verified_columns_per_rank"] = len(picks)
        else:
            multi["gather_verified"] = None
        if backend == "nccl":
            work.gather(); torch.cuda.synchronize(); barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); work.gather(); e1.record()
            torch.cuda.synchronize()
            agt = torch.tensor([e0.elapsed_time(e1)], dtype=torch.float64, device=dev)
            dist.all_reduce(agt, op=dist.ReduceOp.MAX)
            ag_ms = float(agt.item())
            if not a.gather:
                dtg, _, _ = timed(work, True, a.steps, 1)
                gathered_ms = dtg / a.steps * 1e3
        # the other scaling mode in the same run, as an extra field
        other = "weak" if a.scaling == "strong" else "strong"
        w2 = Work(Yf if other == "weak" else np.ascontiguousarray(Yf[:, lo:hi]), None)
        dt2, _, _ = timed(w2, False, a.steps, 1)
        tests2 = p * (a.m * world if other == "weak" else a.m)
        weak = {"scaling": other, "value": tests2 / (dt2 / a.steps), "ms_per_step": dt2 / a.steps * 1e3,
                "m_per_gpu": a.m if other == "weak" else mx}
        del w2


    if rank == 0:
        ms_step = dt / a.steps * 1e3
        tests = p * m_total
        c = 1
        survey_flops = None
        if a.method == "null-exact":
            # executed contraction: num over the padded n, Sxx and s over the weight basis;
            # SURVEY.md §8(d)'s figure for the reference formulation is 2n(2+c) flops per test
            npad8 = -(-n // 8) * 8
            survey_flops = 2.0 * n * (2 + c) * p * m_local
            rank_form_flops = None
            if lr_rank:
                # traits of the shared-weights class (weights = 1 within the guard's tolerance) skip the rank-R phase
                kr4 = 4 * (-(-lr_rank // 4))
                rank_form_flops = 2.0 * (npad8 + (1 + c) * kr4) * p * m_local
                flops_launch = 2.0 * p * (npad8 * m_local + (1 + c) * kr4 * (m_local - (lr_shared or 0)))
                if lr_profile and lr_profile[1]:
                    # every segment of the heritability axis has its own basis: a trait's rank-R phase runs over ITS segment's rank
                    flops_launch = 2.0 * p * (npad8 * m_local + (1 + c) * sum(cnt * 4 * (-(-rk // 4)) for cnt, rk in lr_profile[1]))
            else:   # BLMM_EXACT=full: the (2+c) full-length contractions
                flops_launch = survey_flops
        elif a.method in ("null-grid", "perms"):
            flops_launch = 2.0 * n * p * m_local
        else:
            flops_launch = 2.0 * n * len(grid) * p * m_local
        scan_ms = phases["scan"] / max(ncalls, 1)
        roof = {"bound": "mfma", "achieved": flops_launch / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else None,
                "peak": FP32_MFMA_PEAK_TFLOPS if f32 else FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (flops_launch / (scan_ms * 1e-3) / 1e12 / (FP32_MFMA_PEAK_TFLOPS if f32 else FP64_MFMA_PEAK_TFLOPS)) if scan_ms > 0 else None,
                "traffic": None, "kernel": "k_scan_lr3 / k_scan_lr (rank-R class, weight basis per h2 segment) + k_scan<table, perm> (shared-weights class)" if a.method == "null-exact" else ("k_scan_f32" if f32 else "k_scan"),
                "kernel_ms": scan_ms, "alg_flops_per_launch": flops_launch,
                "alg_bytes_per_launch": (4.0 if f32 else 8.0) * p * m_local,
                "hbm_write_GBps": (4.0 if f32 else 8.0) * p * m_local / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else None,
                "hbm_peak_GBps": HBM_PEAK_GBS,
                "clock": "HIP events around the scan phase on the launching stream, averaged over the timed steps "
                         "(the rocprofv3 --kernel-trace average of the same kernel, which runs at the profiler's lower "
                         "clock, is in profiles/ and in frac_rocprof when profiles/traffic_latest.json matches)"}
        if survey_flops:
            roof["flops_per_test_executed"] = flops_launch / (p * m_local)
            roof["weight_basis_rank"] = lr_rank
            roof["weight_basis_resid"] = lr_resid
            roof["traits_rescanned_full_rank"] = lr_fallback
            roof["traits_shared_weights"] = lr_shared
            if lr_profile:
                roof["weight_basis_segments"] = [{"traits": cnt, "rank": rk} for cnt, rk in lr_profile[1]]
            if rank_form_flops and scan_ms > 0:   # the same launch priced as if every trait ran the rank-R phase (the r01 / r02a form)
                roof["frac_if_all_traits_rank_form"] = rank_form_flops / (scan_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
            roof["reference_formulation_flops_per_launch"] = survey_flops   # 2n(2+c) per test, SURVEY.md §8(d)
            roof["reference_formulation_equiv_TFLOPs"] = survey_flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else None
            roof["note"] = ("achieved/frac count the flops the kernel EXECUTES (low-rank weights form); the same launch "
                            "delivers the reference formulation's 2n(2+c) flops/test at reference_formulation_equiv_TFLOPs")
        if all_rank:
            roof["ms_per_step_all_rank_form"] = all_rank["ms_per_step"]
            roof["scan_ms_all_rank_form"] = all_rank["scan_ms"]
            if rank_form_flops and all_rank["scan_ms"] > 0:
                roof["frac_all_rank_form_measured"] = rank_form_flops / (all_rank["scan_ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
        # HBM traffic comes from PMC counters, which need their own rocprofv3 passes (tools/collect_profiles.py writes
        # profiles/traffic_latest.json): it is NOT measured by this run -- `traffic_measured_in_this_run` says so -- and is only
        # quoted when that profile was taken on the same method and shape
        roof["traffic_measured_in_this_run"] = False
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if tj.get("method") == a.method and tj.get("m") == m_local and tj.get("p") == p and tj.get("n", n) == n:
                    roof["traffic"] = tj.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = "from profiles/traffic_latest.json (" + str(tj.get("round", "?")) + "): " + str(tj.get("source"))
                    if tj.get("rocprof_kernel_avg_ms"):
                        roof["kernel_ms_rocprof"] = tj["rocprof_kernel_avg_ms"]
                        roof["frac_rocprof"] = flops_launch / (tj["rocprof_kernel_avg_ms"] * 1e-3) / 1e12 / roof["peak"]
            except Exception:
                pass
        out = {
            "metric": "trait x marker LOD tests/sec", "value": tests / (dt / a.steps), "unit": "tests/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "ms_per_step_no_phase_marks": (dt_nomarks / a.steps * 1e3) if dt_nomarks else None,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f32" if f32 else "f64", "data": "synthetic",
            "config": {"workload": workload_name(a, n, p, m_total, m_local, f32, world),
                       "n": n, "p": p, "m": m_total, "m_per_gpu": m_local, "method": a.method,
                       "parallelism": f"traits sharded over {world} GPU(s)" + ("; marker rotation sharded, rotated blocks all-gathered over xGMI inside the step" if shard_rotation else ""),
                       "gather_in_step": bool(a.gather),
                       "streams": max(a.streams, 1)},
            "phases_ms": {k: v / max(ncalls, 1) for k, v in phases.items()},
            "allgather_ms": ag_ms, "gathered_ms_per_step": gathered_ms, "other_scaling": weak, "output_finite": bool(chk),
            "multi_gpu": multi,
            "host_api": host_api, "sections_before_timed_loop": (["cpu_baseline (host only)"] if cpu else []) + (["host_api"] if host_api else [])
            + (["all_rank_form loop"] if all_rank else []) + (["no_phase_marks loop"] if dt_nomarks else []),
            "roofline": roof, "cpu_baseline": cpu,
            # the only number the reference publishes for this shape (default null-grid, 10-point grid, 16 Julia threads,
            # Xeon Silver 4214): 2.112 s -- different method and hardware, so it is context, not a vs_baseline
            "reference_published": {"tests_per_s": 260290834 / 2.112011, "seconds": 2.112011, "method": "null-grid (10-point grid)",
                                    "hardware": "Intel Xeon Silver 4214, 16 Julia threads", "source": "README.md:336-339"},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
