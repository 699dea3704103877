#!/usr/bin/env python3
"""How much of the B=1024 step time is the tail (the slowest instance of the batch)?  Launch consecutive,
independent batches on `--inflight` streams (one handle, output and info buffer per stream) so that SIMDs
released by instances that finished early start on the next batch.  Diagnostic only: bench.py keeps one launch
in flight, which is what its per-launch roofline figures refer to.
Run on the GPU box:  python tools/inflight_bench.py [--batch 1024] [--horizon 10] [--steps 200]"""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
pkg = importlib.import_module("quaternion-mpc_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--horizon", type=int, default=10)
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
lib = pkg.load_library()
B, N = a.batch, a.horizon
rec = pkg.random_go1_trot_states(B, config_id=2 if N == 10 else 3)
d_in = torch.from_numpy(rec.view(np.uint8).reshape(B, -1).copy()).cuda()
ref = None
for inflight in (1, 2, 3, 4):
    params = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    solvers = [pkg.Solver(params, B, device=0, lib=lib) for _ in range(inflight)]
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    outs = [torch.zeros(B, 12, dtype=torch.float64, device="cuda") for _ in range(inflight)]
    infos = [torch.zeros(B, pkg.INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(inflight)]

    def run(steps):
        for i in range(steps):
            j = i % inflight
            solvers[j].solve_device(B, d_in.data_ptr(), outs[j].data_ptr(), infos[j].data_ptr(), streams[j].cuda_stream)

    run(2 * inflight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(o, outs[0]) for o in outs)
    if ref is None:
        ref = outs[0].clone()
    print(f"B={B} N={N} batches in flight {inflight}: {B * a.steps / dt:12.0f} solves/s, {1e3 * dt / a.steps:.3f} ms per batch; "
          f"outputs identical across streams: {same}; identical to the single-stream run: {torch.equal(outs[0], ref)}")
