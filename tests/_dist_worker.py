"""Worker of tests/test_sharding_cpu.py: one rank of a world_size-N gloo group.
The local solve is the CPU oracle (this is a test: the product path needs a GPU)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402
from oracle import pyoracle  # noqa: E402

pkg = load_pkg()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total = int(sys.argv[1])
p = pyoracle.default_params(10, 0)
out = pkg.solve_sharded(total, rank, world,
                        lambda first, count: pkg.random_go1_trot_states(count, config_id=2, first=first),
                        lambda rec: pyoracle.solve(p, rec)[0])
full, _ = pyoracle.solve(p, pkg.random_go1_trot_states(total, config_id=2), threads=2)
ok = out.shape == (total, 12) and np.array_equal(out.numpy(), full)
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
