#!/usr/bin/env python3
"""bench.py -- headline benchmark of the bulkscan hot path on MI355X.

Metric (BASELINE.json): (trait x marker) LOD tests per second.  One "step" = one complete bulkscan_null
(null-exact) pass -- device eigen-decomposition of K, rotation of Y and G, per-trait Brent h2, A-side
panels, the f64-MFMA LOD kernel -- over a BXD-shaped synthetic batch (n=79, p=7321, m=35554, fp64;
BASELINE.json configs[1]) whose inputs are already resident in HBM; the p x m LOD matrix stays in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, traits sharded (weak scaling: every rank scans its own BXD-shaped shard of
m traits; no data-path collective in the timed region).  The north_star's optional final step -- an RCCL
all-gather of the LOD column shards over xGMI -- is timed once OUTSIDE the timed region and reported as
`allgather_ms` (add --gather to put it inside the step).
Rank 0 prints ONE JSON line."""
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


def _cpu_worker(args):
    Y, G, K = args
    from oracle import bulklmm_oracle as O  # the checker, used here only as the timed CPU baseline
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            r = O.bulkscan_null(Y, G, K)
    except ImportError:
        r = O.bulkscan_null(Y, G, K)
    return float(np.sum(r.L))


def cpu_baseline(Y, G, K, budget_s=15.0):
    """The NumPy oracle (a literal port of the reference's per-trait null-exact loop, src/bulkscan.jl:268-286)
    on a bounded sample of the same workload, trait blocks spread over worker processes the way the reference
    spreads them over threads."""
    import multiprocessing as mp
    p = G.shape[1]
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    t0 = time.perf_counter()
    _cpu_worker((Y[:, :2], G, K))
    per_trait = (time.perf_counter() - t0) / 2
    per_block = max(1, int(budget_s / max(per_trait, 1e-4)))
    per_block = min(per_block, max(1, Y.shape[1] // cores))
    sample = per_block * cores
    blocks = [(np.ascontiguousarray(Y[:, i * per_block:(i + 1) * per_block]), G, K) for i in range(cores)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(Y[:, :1], G[:, :8], K)] * cores)  # warm the workers (imports) outside the timing
        t0 = time.perf_counter()
        pool.map(_cpu_worker, blocks)
        dt = time.perf_counter() - t0
    return {"value": p * sample / dt, "unit": "tests/s", "cores": cores, "kind": "port",
            "sample": f"bulkscan_null (oracle, NumPy) on the first {sample} of {Y.shape[1]} traits x {p} markers, "
                      f"{cores} processes x {per_block} traits, {dt:.1f} s"}


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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--gather", action="store_true", help="put the RCCL all-gather of the LOD shards inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

    n, p = a.n, a.p
    m_local = a.m if a.scaling == "weak" else -(-a.m // world)
    m_total = m_local * world if a.scaling == "weak" else a.m
    # every rank draws its own shard (rank-dependent seed under weak scaling, a slice under strong scaling)
    if a.scaling == "weak":
        Y, G, K = synth(n, p, m_local, 20240 + 1 + 1000 * rank)
        if rank > 0:  # all ranks share the markers and the kinship of rank 0's draw
            _, G, K = synth(n, p, 1, 20240 + 1)
    else:
        Yf, G, K = synth(n, p, a.m, 20240 + 1)
        lo = min(rank * m_local, a.m)
        Y = np.ascontiguousarray(Yf[:, lo:min(lo + m_local, a.m)])
        m_local = Y.shape[1]

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

    # column-major device operands: a (cols, rows) contiguous tensor IS the rows x cols column-major matrix
    dY = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev)
    dG = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev)
    dK = torch.from_numpy(np.ascontiguousarray(K.T)).to(dev)
    alt = a.method == "alt-grid"
    if a.gather or True:
        # the gathered matrix: shards are contiguous column blocks of the column-major p x m_total L
        dLfull = torch.empty((world, m_local, p), dtype=torch.float64, device=dev) if world > 1 else None
    f32 = a.method == "perms" and a.perm_dtype == "f32"
    if f32:   # fp32 permutation matrix (the all-gather, if any, moves fp32 too)
        dLfull = torch.empty((world, m_local, p), dtype=torch.float32, device=dev) if world > 1 else None
    dL = dLfull[rank] if world > 1 else torch.empty((m_local, p), dtype=torch.float32 if f32 else torch.float64, device=dev)
    dH = torch.empty((m_local, p) if alt else (m_local,), dtype=torch.float64, device=dev)
    grid = [i / 16.0 for i in range(16)] if a.method in ("null-grid", "alt-grid") else None

    stream = torch.cuda.current_stream()
    ctx = B.Context(dev_index, stream.cuda_stream)

    perms = a.method == "perms"
    if perms:
        dy1 = dY[0].contiguous()
        dsc = torch.empty(2, dtype=torch.float64, device=dev)
        dlod = torch.empty(p, dtype=torch.float64, device=dev)

    lr_rank = None
    # --streams S > 1: S independent contexts (own stream, own workspace, own outputs); step i runs on context i % S
    extra = []
    for _ in range(max(a.streams, 1) - 1):
        st_i = torch.cuda.Stream(device=dev)
        extra.append((B.Context(dev_index, st_i.cuda_stream), torch.empty_like(dL), torch.empty_like(dH)))
    step_no = [0]

    def step(gather):
        k = step_no[0] % (1 + len(extra))
        step_no[0] += 1
        c_k, L_k, H_k = (ctx, dL, dH) if k == 0 else extra[k - 1]
        if perms:
            B.scan_perms_dev(c_k, dy1, dG, dK, dsc, dlod, L_k, nperms=m_local, seed=1 + rank)
        else:
            B.bulkscan_dev(c_k, dY, dG, dK, L_k, H_k, method=a.method, h2_grid=grid)
        if gather and world > 1 and backend == "nccl":
            dist.all_gather_into_tensor(dLfull.view(-1), dL.reshape(-1))

    for _ in range(max(a.warmup, 1)):
        step(a.gather)
    torch.cuda.synchronize()
    ctx.set_timing(True)
    ctx.read_timings()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(a.gather)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    phases, ncalls = ctx.read_timings()
    ctx.set_timing(False)
    if a.method == "null-exact":   # one extra (untimed) call with a status read-back: rank of the weight basis
        st = B.bulkscan_dev(ctx, dY, dG, dK, dL, dH, method=a.method, h2_grid=grid, status=True)
        lr_rank = int(st.lowrank_rank)
        lr_resid = float(st.lowrank_resid)

    ag_ms = None
    if world > 1 and backend == "nccl":
        torch.cuda.synchronize(); barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_gather_into_tensor(dLfull.view(-1), dL.reshape(-1))
        torch.cuda.synchronize(); barrier()
        e0.record()
        dist.all_gather_into_tensor(dLfull.view(-1), dL.reshape(-1))
        e1.record()
        torch.cuda.synchronize()
        ag_ms = e0.elapsed_time(e1)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # sanity: the output must be finite (a fast kernel with wrong results is not a result)
    chk = torch.isfinite(dL[: min(64, m_local)]).all().item()

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
            if lr_rank:
                kr4 = 4 * (-(-lr_rank // 4))
                flops_launch = 2.0 * (npad8 + (1 + c) * kr4) * p * m_local
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
                "traffic": None, "kernel": "k_scan_lr (exact, low-rank weights)" if a.method == "null-exact" else ("k_scan_f32" if f32 else "k_scan"),
                "kernel_ms": scan_ms, "alg_flops_per_launch": flops_launch,
                "alg_bytes_per_launch": (4.0 if f32 else 8.0) * p * m_local,
                "hbm_write_GBps": (4.0 if f32 else 8.0) * p * m_local / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else None,
                "hbm_peak_GBps": HBM_PEAK_GBS}
        if survey_flops:
            roof["flops_per_test_executed"] = flops_launch / (p * m_local)
            roof["weight_basis_rank"] = lr_rank
            roof["weight_basis_resid"] = lr_resid
            roof["reference_formulation_flops_per_launch"] = survey_flops   # 2n(2+c) per test, SURVEY.md §8(d)
            roof["reference_formulation_equiv_TFLOPs"] = survey_flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else None
            roof["note"] = ("achieved/frac count the flops the kernel EXECUTES (low-rank weights form); the same launch "
                            "delivers the reference formulation's 2n(2+c) flops/test at reference_formulation_equiv_TFLOPs")
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if tj.get("method") == a.method and tj.get("m") == m_local and tj.get("p") == p:
                    roof["traffic"] = tj.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = tj.get("source")
            except Exception:
                pass
        out = {
            "metric": "trait x marker LOD tests/sec", "value": tests / (dt / a.steps), "unit": "tests/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f32" if f32 else "f64", "data": "synthetic",
            "config": {"workload": f"bulkscan_null-shaped: method={a.method} n={n} p={p} m={m_total} fp64 "
                                   f"(BASELINE.json configs[1]; {m_local} traits per GPU)",
                       "n": n, "p": p, "m": m_total, "m_per_gpu": m_local, "method": a.method,
                       "parallelism": f"traits sharded over {world} GPU(s)", "gather_in_step": bool(a.gather),
                       "streams": max(a.streams, 1)},
            "phases_ms": {k: v / max(ncalls, 1) for k, v in phases.items()},
            "allgather_ms": ag_ms, "output_finite": bool(chk),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
