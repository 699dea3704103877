"""CPU suite: bench.py's launcher path.  `python bench.py --gpus N` without a launcher in the environment re-executes
itself under torch.distributed.run with one rank per GPU (the driver's multi-GPU call may be either form).  Here the
ranks stop after the rendezvous (QMPC_BENCH_DRYRUN, gloo): what is checked is the self-launch, the 127.0.0.1
rendezvous and that exactly one JSON line comes out of rank 0."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def _run(args, env_extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300)


def test_plain_invocation_with_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"QMPC_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "n_gpus": 2, "sum_of_ranks": 3.0}


def test_launcher_mismatch_is_an_error():
    """a launcher that set another world size than --gpus says: refuse (exit code 2), do not guess"""
    r = _run(["--gpus", "4"], {"QMPC_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
