#!/usr/bin/env python3
"""bench.py -- quaternion-MPC solves/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic LeggedState
records: `batch` independent MPC problems (Go1, 12 contact forces, horizon N),
inputs already resident in HBM, one kernel launch per step per GPU.  With
--gpus N > 1 the driver launches one rank per GPU (torch.distributed, RCCL);
instances shard embarrassingly (rank r solves its own `batch` records: weak
scaling) and ONE all_gather of the [batch,12] force block per step returns the
results to every rank.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import re
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector = matrix peak
W_ALG_KFLOP_PER_KNOT = 159.0     # W_alg(N) = 159 N kFLOP per solve


def load_pkg():
    name = "quaternion_mpc_amd"
    spec = importlib.util.spec_from_file_location(name, REPO / "quaternion-mpc_amd" / "__init__.py",
                                                  submodule_search_locations=[str(REPO / "quaternion-mpc_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def flush_c_stdio():
    """RCCL prints its version banner through C stdio (NCCL_DEBUG=VERSION on the GPU boxes), which sits in the C
    buffer until exit when stdout is a pipe: push it out now, so that the JSON line is the last line of stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def usable_cores() -> int:
    """Host threads this process may actually run on: CPU affinity capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants 16 CPUs of quota; oversubscribing it is slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(pkg, horizon: int, config_id: int, seconds: float = 12.0, model: str = "quat"):
    """Time the CPU oracle ("port") on a bounded sample of the same workload."""
    from oracle import pyoracle  # checker / baseline only

    cores = usable_cores()
    p = {"quat": pyoracle.default_params, "convex": pyoracle.default_convex_params,
         "biped8": pyoracle.default_biped8_params}[model](horizon, 0)
    gen = {"quat": pkg.random_go1_trot_states, "convex": pkg.random_go1_convex_states,
           "biped8": pkg.random_biped8_states}[model]
    solve = {"quat": pyoracle.solve, "convex": pyoracle.convex_solve, "biped8": pyoracle.solve8}[model]
    probe = gen(4 * cores, config_id=config_id)
    t0 = time.perf_counter()
    solve(p, probe, threads=cores)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-9)
    n = int(min(max(rate * seconds, 4 * cores), 65536))
    rec = gen(n, config_id=config_id)
    t0 = time.perf_counter()
    forces, info = solve(p, rec, threads=cores)
    dt = time.perf_counter() - t0
    return {
        "_forces": forces,       # popped by the caller: parity of the GPU batch against the same instances
        "value": n / dt, "unit": "solves/s", "cores": cores, "kind": "port",
        "sample": f"first {n} instances of the same synthetic workload (N={horizon}), {cores} host threads "
                  f"(= usable cores: affinity capped by the cgroup CPU quota; {os.cpu_count()} logical CPUs visible), "
                  f"instance-parallel; oracle/ C restatement, {dt:.1f} s; mean {float(info['iterations'].mean()):.1f} iterations",
    }


def two_in_flight(pkg, lib, params, args, d_in, NU):
    """Secondary measurement, never `value`: the same K steps with TWO batches in flight (two handles, two
    streams, separate outputs).  With one launch in flight a step lasts as long as its slowest instance (every
    SIMD holds one instance at B=1024); SIMDs released early then start on the next, independent batch."""
    import torch
    B = args.batch
    biped, convex = args.model == "biped8", args.model == "convex"
    # gains in the workspace (19 KB of LDS per instance at N=10): both batches are resident from the start
    prev = os.environ.get("QMPC_VARIANT")
    if prev is None and B <= 1024 and not biped:   # the 8-point model already picks its workspace variant
        os.environ["QMPC_VARIANT"] = "2"           # read by qmpc_create
    try:
        solvers = [pkg.Solver(params, B, device=torch.cuda.current_device(), lib=lib) for _ in range(2)]
    finally:
        if prev is None:
            os.environ.pop("QMPC_VARIANT", None)
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [torch.zeros(B, NU, dtype=torch.float64, device="cuda") for _ in range(2)]
    infos = [torch.zeros(B, pkg.INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)]

    def run(steps):
        for i in range(steps):
            j = i & 1
            fn = solvers[j].solve8_device if biped else (solvers[j].convex_solve_device if convex else solvers[j].solve_device)
            fn(B, d_in.data_ptr(), outs[j].data_ptr(), infos[j].data_ptr(), streams[j].cuda_stream)

    run(max(args.warmup, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = bool(torch.equal(outs[0], outs[1]))
    for sv in solvers:
        sv.close()
    return {"value": B * args.steps / dt, "unit": "solves/s", "ms_per_step": 1e3 * dt / args.steps, "steps": args.steps,
            "outputs_identical": same,
            "note": "secondary: consecutive independent batches on two streams (two handles, gains in the workspace "
                    "so that both batches are resident); not the contract value"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step (BASELINE config 1)")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--model", choices=["quat", "convex", "biped8"], default="quat",
                    help="quat: legged::QuatMpc's inner loop (the BASELINE metric); convex: legged::ConvexMpc's "
                         "(SURVEY 8f rank 1), same solver core; biped8: the QuatMpc problem with 8 contact points "
                         "(BASELINE config 5, synthetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-in-flight", action="store_true",
                    help="skip the secondary measurement with two batches in flight (N=1 only)")
    ap.add_argument("--check", action="store_true", help="also compare a sample against the oracle")
    ap.add_argument("--force-dist", action="store_true",
                    help="DIAGNOSTIC: run the multi-rank code path (process group, async all_gather per step, barriers) "
                         "even with one rank, to exercise the RCCL calls on a single GPU")
    ap.add_argument("--selftest-gloo", action="store_true",
                    help="TEST ONLY: run the multi-rank pipeline (sharding, double buffering, async gather) on CPU "
                         "tensors over gloo with the CPU oracle standing in for the kernel; prints no metric")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: WORLD_SIZE={world} != --gpus {args.gpus}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    selftest = args.selftest_gloo
    multi = world > 1 or args.force_dist        # the collective path is on
    if not selftest and not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    dev = "cpu" if selftest else "cuda"
    if not selftest:
        torch.cuda.set_device(local)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if selftest else "nccl", rank=rank, world_size=world)

    pkg = load_pkg()
    N, B = args.horizon, args.batch
    convex = args.model == "convex"
    biped = args.model == "biped8"
    NU = 24 if biped else 12
    config_id = 5 if biped else ((12 if N == 10 else 13) if convex else (2 if N == 10 else 3))
    gen = pkg.random_biped8_states if biped else (pkg.random_go1_convex_states if convex else pkg.random_go1_trot_states)
    # synthetic Go1 trot states; rank r owns instances [r*B, (r+1)*B)
    rec = gen(B, config_id=config_id, first=rank * B)
    # one result block per step slot: [B x NU forces | B x qmpc_info (status, iterations, cost, ...)] in ONE buffer,
    # so that a single collective carries forces and status
    IW = pkg.INFO_DTYPE.itemsize // 8           # qmpc_info is 40 bytes = 5 doubles
    assert pkg.INFO_DTYPE.itemsize == 8 * IW
    SLOTS = int(os.environ.get("QMPC_BENCH_SLOTS", "2"))   # result blocks in rotation (gather of step i drains under later solves)
    blocks = [torch.zeros(B * (NU + IW), dtype=torch.float64, device=dev) for _ in range(SLOTS)]

    def forces_of(blk):
        return blk[:B * NU].view(B, NU)

    def info_of(blk):
        return blk[B * NU:]
    def oracle_solve(po, records, threads=1):
        if biped:
            return po.solve8(po.default_biped8_params(N, 0), records, threads=threads)
        if convex:
            return po.convex_solve(po.default_convex_params(N, 0), records, threads=threads)
        return po.solve(po.default_params(N, 0), records, threads=threads)

    if selftest:
        from oracle import pyoracle   # test-only stand-in for the kernel
        f_ref, _ = oracle_solve(pyoracle, rec)
        f_ref = torch.from_numpy(f_ref)
        solver = stream = None

        def launch(blk):
            forces_of(blk).copy_(f_ref)
    else:
        lib = pkg.load_library()
        params = (pkg.default_biped8_params if biped else
                  (pkg.default_convex_params if convex else pkg.default_params))(N, pkg.MODE_CONVERGED, lib)
        solver = pkg.Solver(params, B, device=local, lib=lib)
        d_in = torch.from_numpy(rec.view(np.uint8).reshape(B, -1).copy()).cuda()
        # a real (non-null) stream: the C ABI treats NULL as "the handle's own stream"
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)

        def launch(blk):
            (solver.solve8_device if biped else (solver.convex_solve_device if convex else solver.solve_device))(
                B, d_in.data_ptr(), forces_of(blk).data_ptr(), info_of(blk).data_ptr(), stream.cuda_stream)

    counts = [B] * world
    # double-buffered outputs: the gather of step i (RCCL's own stream) overlaps the solve of
    # step i+1 (launch stream); one collective per step, never on the solve's critical path
    gathered = [torch.zeros(world, B * (NU + IW), dtype=torch.float64, device=dev) for _ in range(SLOTS)] if multi else None
    pending = [None] * SLOTS

    def step(i):
        buf = i % SLOTS
        if pending[buf] is not None:          # the buffer's previous gather must have drained
            pending[buf].wait()
            pending[buf] = None
        launch(blocks[buf])
        if multi:
            pending[buf] = dist.all_gather_into_tensor(gathered[buf].view(-1), blocks[buf], async_op=True)

    def drain():
        for w in pending:
            if w is not None:
                w.wait()
        for j in range(SLOTS):
            pending[j] = None

    def sync():
        if not selftest:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    sync()
    if multi:
        dist.barrier()
    sync()
    flush_c_stdio()
    if not selftest:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if not selftest:
        ev0.record(stream)
    for i in range(args.steps):
        step(i)
    if not selftest:
        ev1.record(stream)
    drain()
    sync()
    if multi:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # HIP events on the launch stream: average launch duration (back-to-back launches)
    kernel_ms = (ev0.elapsed_time(ev1) / args.steps) if not selftest else float("nan")
    last = blocks[(args.steps - 1) % SLOTS]
    d_f = forces_of(last)
    if multi:
        # every rank holds every rank's forces and status: check the gathered block against the local one
        g = gathered[(args.steps - 1) % SLOTS]
        assert torch.equal(g[rank], last), "gathered block differs from the local shard"
    if selftest:
        ok = True
        if multi:
            from oracle import pyoracle
            full, _ = oracle_solve(pyoracle, gen(world * B, config_id=config_id))
            gf = gathered[(args.steps - 1) % SLOTS][:, :B * NU].reshape(world * B, NU)
            ok = bool(np.array_equal(gf.numpy(), full))
        if rank == 0:
            print(json.dumps({"selftest": "gloo", "n_ranks": world, "steps": args.steps, "ok": ok}), flush=True)
        if multi:
            dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    # status of EVERY rank's instances (the gathered blocks carry them); iterations of the local shard
    all_info = (gathered[(args.steps - 1) % SLOTS][:, B * NU:] if multi else info_of(last).view(1, -1))
    info = np.ascontiguousarray(all_info.cpu().numpy()).view(pkg.INFO_DTYPE).reshape(world, B)
    n_ok = int((info["status"] == 0).sum())
    mean_iters = float(info["iterations"].mean())

    if rank == 0:
        # HBM traffic per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
        traffic = None
        try:
            latest = sorted((REPO / "profiles").glob("r*_pmc_traffic.json"),
                            key=lambda q: [int(t) for t in re.findall(r"\d+", q.name)])[-1]
            pm = json.loads(latest.read_text())
            if B == 1024 and N == 10 and args.model == "quat":
                traffic = pm["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        total = world * B * args.steps
        value = total / elapsed
        # algorithmic FP64 flops per solve; SURVEY 8d prices the 24-input model at ~110 kFLOP/knot/iteration
        w_alg = (330.0 if biped else W_ALG_KFLOP_PER_KNOT) * 1e3 * N
        achieved = w_alg * B / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": ("MPC QP solves/sec (synthetic biped, 8 contact points, 24 forces, N=%d)" % N) if biped else
                      "MPC QP solves/sec (Go1%s, 12 forces, N=%d)" % (" ConvexMpc" if convex else "", N),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Batch={B} random {'biped8 ' if biped else 'Go1 '}{'ConvexMpc ' if convex else ''}states per GPU, N={N}, converged mode "
                                   f"(interior point to |dU|<=1e-8 N), generator seed 0x5EED0000+{config_id}",
                       "batch_per_gpu": B, "horizon": N, "parallelism": f"instance-sharded x{world}",
                       "instances": world * B, "converged": n_ok, "mean_iterations": mean_iters},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "qmpc_solve_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_flops_per_launch": w_alg * B,
                         "note": "FP64 MFMA/vector roof (78.6 TF); algorithmic work W_alg=159*N kFLOP/solve (SURVEY 8d); "
                                 "compulsory HBM traffic is 460 B/solve, i.e. not the binding roof"},
        }
        if args.check:
            from oracle import pyoracle
            idx = np.arange(0, B, max(B // 64, 1))
            fo, _ = oracle_solve(pyoracle, rec[idx], threads=usable_cores())
            out["config"]["force_linf_vs_cpu"] = float(np.abs(d_f.cpu().numpy()[idx] - fo).max())
        if world == 1 and not args.no_in_flight:
            out["two_in_flight"] = two_in_flight(pkg, lib, params, args, d_in, NU)
            # the drop-in entry point as the controller calls it: HOST buffers, H2D + kernel + D2H, blocking
            # (SURVEY 8d quotes this beside the resident figure; never `value`)
            hsolve = solver.solve8 if biped else (solver.convex_solve if convex else solver.solve)
            hsolve(rec)
            reps = max(3, min(20, args.steps))
            t0 = time.perf_counter()
            for _ in range(reps):
                hsolve(rec)
            dt = time.perf_counter() - t0
            out["host_buffer_call"] = {"value": B * reps / dt, "unit": "solves/s", "ms_per_call": 1e3 * dt / reps,
                                       "note": "qmpc_solve with host buffers (PCIe both ways + kernel, blocking)"}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(pkg, N, config_id, model=args.model)
            f_cpu = cb.pop("_forces")
            m = min(len(f_cpu), B)      # the sample starts with the instances of the timed batch
            cb["force_linf_gpu_vs_cpu"] = float(np.abs(d_f.cpu().numpy()[:m] - f_cpu[:m]).max())
            cb["force_linf_instances"] = m
            out["cpu_baseline"] = cb
        flush_c_stdio()
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    solver.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
