"""Trait sharding across the GPUs of one node (one process per GPU, torch.distributed).

Every trait column is independent given (X0, lambda) -- the reference itself blocks contiguous trait ranges over
threads (src/bulkscan.jl:263-286) -- so rank r scans columns [lo, hi) of Y and owns the contiguous block
L[:, lo:hi] of the column-major p x m LOD matrix.  The only collective the path can need is the all-gather of
those blocks (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
from __future__ import annotations

from typing import Tuple


def trait_shard(m: int, rank: int, world: int) -> Tuple[int, int]:
    """Column range of rank `rank`: blocks of ceil(m / world) columns, the last ranks may be short or empty --
    [r*ceil(m/R), min(m, (r+1)*ceil(m/R))), SURVEY.md §8(e).  The SAME partition as the C ABI's blmm_multi_shard
    (bulklmm.jl_amd/csrc/blmm_multi.hip; tests/test_abi.py checks the two against each other): fixed-size blocks are
    what lets the all-gather run in place on the full-size (padded) matrix."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    blk = -(-m // world)
    lo = min(m, rank * blk)
    return lo, min(m, lo + blk)


def shard_sizes(m: int, world: int):
    return [trait_shard(m, r, world)[1] - trait_shard(m, r, world)[0] for r in range(world)]


def allgather_lod(L_local, m: int, group=None):
    """All-gather the per-rank column blocks into the full matrix on every rank.

    L_local: tensor of shape (m_local, p) -- i.e. the p x m_local column-major block.  Returns (m, p).
    Ragged shards are padded to the largest one for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(m, world)
    p = L_local.shape[1]
    mx = max(sizes)
    if L_local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match trait_shard()")
    if all(s == mx for s in sizes):
        out = torch.empty((world * mx, p), dtype=L_local.dtype, device=L_local.device)
        dist.all_gather_into_tensor(out, L_local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx, p), dtype=L_local.dtype, device=L_local.device)
    pad[: L_local.shape[0]] = L_local
    buf = torch.empty((world * mx, p), dtype=L_local.dtype, device=L_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
