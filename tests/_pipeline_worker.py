"""Worker of tests/test_sharding_cpu.py::test_step_pipeline_gloo: one rank of a gloo group driving the multi-rank step
pipeline that bench.py uses on the GPUs (quaternion-mpc_amd/sharding.py: StepPipeline -- result blocks in rotation,
one asynchronous all_gather per step, drain) with the CPU oracle standing in for the kernel launch (this is a test:
the product path needs a GPU)."""
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
B, steps, model = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nu = 24 if model == "biped8" else 12
IW = pkg.INFO_DTYPE.itemsize // 8
if model == "biped8":
    p = pyoracle.default_biped8_params(16, 0); gen, solve, cfg = pkg.random_biped8_states, pyoracle.solve8, 5
else:
    p = pyoracle.default_params(10, 0); gen, solve, cfg = pkg.random_go1_trot_states, pyoracle.solve, 2
rec = gen(B, config_id=cfg, first=rank * B)
f_ref, info_ref = solve(p, rec)
f_ref = torch.from_numpy(f_ref)
i_ref = torch.from_numpy(np.ascontiguousarray(info_ref).view(np.float64).copy())
pipe = pkg.StepPipeline(world, rank, B * (nu + IW), "cpu", slots=2)
calls = []


def launch(blk):
    calls.append(1)
    blk[:B * nu].view(B, nu).copy_(f_ref * float(len(calls)))       # step-dependent content: stale slots would show
    blk[B * nu:].copy_(i_ref)


pipe.reset_stats()
for i in range(steps):
    pipe.step(i, launch)
pipe.drain()
stt = pipe.stats()      # the diagnosis fields of a multi-rank bench line: one gather per step, every one of them waited for
ok_stats = stt["gathers"] == steps and stt["waits"] == steps and stt["gather_wait_ms"] >= 0.0
full, finfo = solve(p, gen(world * B, config_id=cfg), threads=2)
g = pipe.all_blocks(steps - 1)
ok = ok_stats and g.shape == (world, B * (nu + IW))
ok = ok and np.array_equal(g[:, :B * nu].reshape(world * B, nu).numpy(), full * float(steps))
st = np.ascontiguousarray(g[:, B * nu:].numpy()).view(pkg.INFO_DTYPE).reshape(world * B)
ok = ok and np.array_equal(st["status"], finfo["status"]) and np.array_equal(st["iterations"], finfo["iterations"])
# the slot before the last one still holds the previous step of every rank
gp = pipe.all_blocks(steps - 2)
ok = ok and np.array_equal(gp[:, :B * nu].reshape(world * B, nu).numpy(), full * float(steps - 1))
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
