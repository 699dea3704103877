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


def _fused_gather_ok() -> bool:
    """all_gather_into_tensor exists for every backend this path runs on (nccl = RCCL, gloo); the choice is made
    from the backend name, once and identically on every rank -- never by catching an error on some ranks."""
    import torch.distributed as dist

    return dist.get_backend() in ("nccl", "gloo") and hasattr(dist, "all_gather_into_tensor")


def gather_forces(local, world: int, counts=None):
    """The single collective of the path: all_gather of the local [count, columns] block.

    `local` is a torch tensor on the device the process group lives on.  Equal
    shard sizes use all_gather_into_tensor; ragged shards pad to the largest.
    Returns the [total, columns] tensor on every rank.
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
        if _fused_gather_ok():
            out = torch.empty((world * m, local.shape[1]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local.contiguous())
            return out
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous())
        return torch.cat(parts, 0)
    pad = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[:n] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def solve_sharded(total: int, rank: int, world: int, make_inputs: Callable[[int, int], np.ndarray],
                  solve_local: Callable[[np.ndarray], np.ndarray], device="cpu", columns: int = 12):
    """Solve instances [0,total) across `world` ranks; every rank gets all forces.

    make_inputs(first, count) -> structured records; solve_local(records) -> [count, columns] float64
    (columns = 12 forces for Go1, 24 for the 8-contact-point model); an empty shard contributes a [0, columns] block.
    """
    import torch

    first, count = shard_range(total, rank, world)
    rec = make_inputs(first, count)
    f = np.asarray(solve_local(rec)) if count else np.zeros((0, columns))
    if f.ndim != 2 or f.shape[1] != columns:
        raise ValueError(f"solve_local returned {f.shape}, expected [{count}, {columns}]")
    local = torch.from_numpy(np.ascontiguousarray(f)).to(device)
    counts = [shard_range(total, r, world)[1] for r in range(world)]
    return gather_forces(local, world, counts)


class StepPipeline:
    """The per-step pipeline of a multi-rank run: `launch(block)` fills this rank's result block (forces and status
    words of its shard, one contiguous float64 buffer), then ONE asynchronous all_gather returns every rank's block to
    every rank.  `slots` blocks rotate, so the gather of step i drains under the solve of step i+1 and never sits on
    the solve's critical path.  With world == 1 and multi=False there is no collective at all.  bench.py drives it
    with the HIP launch; tests/_pipeline_worker.py drives the same code over gloo with a stand-in launch.

    Diagnosis of a multi-GPU run (stats()): how long this rank's solve stream was held up by a gather that had not drained
    when its slot came round again -- on a GPU the wait is a stream dependency, so it is measured with a pair of events
    on the current stream around `work.wait()`; on CPU tensors (gloo) the wait blocks the host and is timed there."""

    def __init__(self, world: int, rank: int, block_elems: int, device, slots: int = 2, multi: bool | None = None):
        import torch

        self.world, self.rank, self.slots = world, rank, slots
        self.multi = (world > 1) if multi is None else multi
        self.blocks = [torch.zeros(block_elems, dtype=torch.float64, device=device) for _ in range(slots)]
        self.gathered = ([torch.zeros(world, block_elems, dtype=torch.float64, device=device) for _ in range(slots)]
                         if self.multi else None)
        self.pending = [None] * slots
        self._cuda = self.blocks[0].is_cuda
        self.reset_stats()

    def reset_stats(self) -> None:
        """forget the waits recorded so far (call at the start of a timed region)"""
        self._wait_host_s = 0.0
        self._wait_events = []
        self._gathers = 0
        self._waits = 0

    def _wait(self, buf: int) -> None:
        w = self.pending[buf]
        if w is None:
            return
        self._waits += 1
        if self._cuda:
            import torch

            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            w.wait()                     # nccl: the current stream waits for the collective; the host does not block
            e1.record()
            self._wait_events.append((e0, e1))
        else:
            import time

            t0 = time.perf_counter()
            w.wait()
            self._wait_host_s += time.perf_counter() - t0
        self.pending[buf] = None

    def step(self, i: int, launch) -> None:
        import torch.distributed as dist

        buf = i % self.slots
        self._wait(buf)                  # the buffer's previous gather must have drained
        launch(self.blocks[buf])
        if self.multi:
            self.pending[buf] = dist.all_gather_into_tensor(self.gathered[buf].view(-1), self.blocks[buf], async_op=True)
            self._gathers += 1

    def drain(self) -> None:
        for j in range(self.slots):
            self._wait(j)

    def stats(self) -> dict:
        """{"gathers": collectives issued, "waits": waits performed, "gather_wait_ms": total time the solve stream (GPU) or
        the host (CPU tensors) was held by them} since the last reset_stats().  Synchronises the recorded events."""
        ms = 1e3 * self._wait_host_s
        for e0, e1 in self._wait_events:
            e1.synchronize()
            ms += e0.elapsed_time(e1)
        return {"gathers": self._gathers, "waits": self._waits, "gather_wait_ms": ms}

    def block(self, i: int):
        return self.blocks[i % self.slots]

    def all_blocks(self, i: int):
        """[world, block_elems]: every rank's block of step i (this rank's own block when there is no collective)."""
        return self.gathered[i % self.slots] if self.multi else self.blocks[i % self.slots].view(1, -1)
