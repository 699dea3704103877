"""CPU suite: the N > 1 path (instance sharding + the single force gather) with
world_size-2/3 gloo groups; rendezvous on 127.0.0.1."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges(pkg):
    for total in (0, 1, 7, 1024, 262144):
        for world in (1, 2, 3, 8):
            parts = [pkg.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        pkg.shard_range(10, 2, 2)


@pytest.mark.parametrize("world,total", [(2, 24), (3, 23)])
def test_sharded_solve_gloo(world, total):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           str(REPO / "tests" / "_dist_worker.py"), str(total)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("world,model", [(2, "quat"), (3, "quat"), (2, "biped8")])
def test_step_pipeline_gloo(world, model):
    """The multi-rank step pipeline bench.py runs on the GPUs (StepPipeline: per-rank shard, result blocks in
    rotation, one async all_gather per step carrying forces AND status, drain) on CPU tensors over gloo; the oracle
    stands in for the kernel launch.  24-column blocks (8 contact points) included."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           str(REPO / "tests" / "_pipeline_worker.py"), "8", "5", model]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_sharded_solve_column_count_and_empty_shards():
    """world > total leaves ranks without instances; their empty block must carry the model's column count
    (24 for the 8-contact-point model), and a wrong local shape is an error, not a silent reshape."""
    import numpy as np
    from conftest import load_pkg
    pkg = load_pkg()
    out = pkg.solve_sharded(0, 0, 1, lambda f, c: np.zeros(c), lambda rec: np.zeros((0, 24)), columns=24)
    assert tuple(out.shape) == (0, 24)
    with pytest.raises(ValueError):
        pkg.solve_sharded(3, 0, 1, lambda f, c: np.zeros(c), lambda rec: np.zeros((3, 12)), columns=24)
