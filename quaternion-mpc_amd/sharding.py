"""Multi-GPU path of the quaternion-MPC solve.

MPC instances are independent, so the batch shards embarrassingly: rank r of
`world` solves the contiguous block [first, first+count) on its own GPU with its
own handle and stream; there is NO collective inside the solve.  One
all_gather of the [count, 12] float64 force block per step returns every
result to every rank (RCCL over xGMI with backend "nccl"; "gloo" in the CPU
tests).  At B = 262144 that is 25 MB in total -- latency-, not bandwidth-bound.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block split: (first instance, count) of `rank`; sizes differ by at most 1."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    first = (total * rank) // world
    last = (total * (rank + 1)) // world
    return first, last - first


def gather_forces(local, world: int, counts=None):
    """The single collective of the path: all_gather of the local [count,12] block.

    `local` is a torch tensor on the device the process group lives on.  Equal
    shard sizes use all_gather_into_tensor; ragged shards pad to the largest.
    Returns the [total, 12] tensor on every rank.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    n = local.shape[0]
    if counts is None:
        t = torch.tensor([n], dtype=torch.int64, device=local.device)
        allc = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allc, t)
        counts = [int(c.item()) for c in allc]
    m = max(counts)
    if all(c == m for c in counts):
        out = torch.empty((world * m, local.shape[1]), dtype=local.dtype, device=local.device)
        try:
            dist.all_gather_into_tensor(out, local.contiguous())
        except (RuntimeError, NotImplementedError):   # backend without the fused form
            parts = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(parts, local.contiguous())
            out = torch.cat(parts, 0)
        return out
    pad = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[:n] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def solve_sharded(total: int, rank: int, world: int, make_inputs: Callable[[int, int], np.ndarray],
                  solve_local: Callable[[np.ndarray], np.ndarray], device="cpu"):
    """Solve instances [0,total) across `world` ranks; every rank gets all forces.

    make_inputs(first, count) -> structured records; solve_local(records) -> [count,12] float64.
    """
    import torch

    first, count = shard_range(total, rank, world)
    rec = make_inputs(first, count)
    f = solve_local(rec) if count else np.zeros((0, 12))
    local = torch.from_numpy(np.ascontiguousarray(f)).to(device)
    counts = [shard_range(total, r, world)[1] for r in range(world)]
    return gather_forces(local, world, counts)
