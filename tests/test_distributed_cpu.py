"""world_size-2 gloo test of the multi-GPU path's host logic on CPU: traits are sharded with trait_shard(), every
rank scans its own contiguous column block (here with the CPU oracle standing in for the GPU scan, so the test runs
without a GPU), and allgather_lod() must reassemble exactly the single-process p x m LOD matrix -- including a
ragged split (m not divisible by the world size)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, m, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bulklmm_jl_amd as B
    from common import make_data
    from oracle import bulklmm_oracle as O

    Y, G, K, _ = make_data(p=40, m=m, seed=77)
    lo, hi = B.trait_shard(m, rank, world)
    grid = [i / 10.0 for i in range(10)]
    loc = O.bulkscan_null_grid(Y[:, lo:hi], G, K, grid)          # this rank's column block
    L_local = torch.from_numpy(np.ascontiguousarray(loc.L.T))     # (m_local, p) == p x m_local column-major
    full = B.allgather_lod(L_local, m)
    h2 = B.allgather_lod(torch.from_numpy(loc.h2_null_list.reshape(-1, 1).copy()), m)
    ref = O.bulkscan_null_grid(Y, G, K, grid)
    ok = (full.shape == (m, 40) and np.array_equal(full.numpy().T, ref.L)
          and np.array_equal(h2.numpy()[:, 0], ref.h2_null_list))
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("m", [8, 7])
def test_two_rank_trait_sharding_and_allgather(tmp_path, m):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
