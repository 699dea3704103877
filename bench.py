#!/usr/bin/env python3
"""bench.py -- quaternion-MPC solves/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic LeggedState
records: `batch` independent MPC problems (Go1, 12 contact forces, horizon N),
inputs already resident in HBM, one kernel launch per step per GPU.  With
--gpus N > 1 the driver launches one rank per GPU (torch.distributed, RCCL);
instances shard embarrassingly (rank r solves its own `batch` records: weak
scaling) and ONE all_gather of the [batch,12] force block per step returns the
results to every rank; such a run also reports BASELINE config 4
(262144 instances sharded over the ranks) under "config4".

`value` is the device-resident rate; rates.host_buffer_call is the PCIe-inclusive
rate of the host-buffer entry point.  Prints ONE JSON line on rank 0.
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
        "sample": f"first {n} instances of the same workload, {cores} threads (cgroup quota; {os.cpu_count()} logical CPUs), "
                  f"oracle/ C restatement, {dt:.1f} s",
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
    return {"value": B * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps, "outputs_identical": same}


def traffic_from_profiles(B, N, model):
    """HBM traffic per launch from the newest committed PMC pass of THIS workload (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs, tools/pmc_traffic.py; PMC counters cannot be collected from inside the timed
    process).  Returns (bytes or None, source description)."""
    if model != "quat":
        return None, "no committed PMC pass for this model"
    pat = "r*_pmc_traffic.json" if (B == 1024 and N == 10) else f"r*_pmc_traffic_b{B}_n{N}.json"
    try:
        latest = sorted((REPO / "profiles").glob(pat),
                        key=lambda q: [int(t) for t in re.findall(r"\d+", q.name)])[-1]
        pm = json.loads(latest.read_text())
        src = f"profiles/{latest.name}: {pm.get('source', '')}; {pm.get('calibration_note', 'FETCH_SIZE uncalibrated for 8-byte-per-lane reads')}"
        return pm.get("traffic_bytes_per_launch_calibrated", pm["traffic_bytes_per_launch"]), src
    except (OSError, ValueError, KeyError, IndexError):
        return None, f"no committed PMC pass for this workload (profiles/{pat})"


def self_launch(n: int, argv, module: str = "torch.distributed.run", extra_env=None) -> int:
    """Re-run this script under the torch.distributed launcher with one rank per GPU (127.0.0.1 rendezvous: the container
    hostname may not resolve).  Returns the launcher's exit code; the ranks' stdout (rank 0 prints the JSON line) passes
    through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", module, "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    return subprocess.call(cmd, env=env)


def sq_from_profiles(B, N, model):
    """MFMA-busy and issue fractions of the dominant kernel from the newest committed SQ counter pass of this workload
    (tools/pmc_sq.sh / tools/pmc_lane.sh: rocprofv3 --pmc in runs of their own)."""
    tag = f"b{B}_n{N}"
    cands = sorted((REPO / "profiles").glob(f"r*_sq_summary_{tag}.json"), key=lambda q: [int(t) for t in re.findall(r"\d+", q.name)])
    if model != "quat" or not cands:
        return None
    try:
        d = json.loads(cands[-1].read_text())
        # say which round and which kernel the pass was taken on (a pass older than the kernel it is quoted for is stale)
        d["source"] = f"profiles/{cands[-1].name} (round {d.get('round', cands[-1].name[:3])}, kernel {d.get('kernel', 'see the file')})"
        return d
    except (OSError, ValueError):
        return None


def compact(o):
    """floats to six significant digits: the line has to fit the driver's 8 KB tail"""
    if isinstance(o, float):
        return float(f"{o:.6g}")
    if isinstance(o, dict):
        return {k: compact(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v) for v in o]
    return o


def loop_form(persistent: bool) -> str:
    return "persistent wave-per-robot kernel" if persistent else "three kernels per tick (graph replay)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step (BASELINE config 2)")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--model", choices=["quat", "convex", "biped8"], default="quat",
                    help="quat: legged::QuatMpc's inner loop (the BASELINE metric); convex: legged::ConvexMpc's "
                         "(SURVEY 8f rank 1), same solver core; biped8: the QuatMpc problem with 8 contact points "
                         "(BASELINE config 5, synthetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-in-flight", action="store_true",
                    help="skip the secondary measurement with two batches in flight (N=1 only)")
    ap.add_argument("--no-reference-mode", action="store_true",
                    help="skip the secondary figure of the reference's own solver mode (AL-iLQR, 10 iterations) on the device")
    ap.add_argument("--no-closed-loop", action="store_true",
                    help="skip the secondary device-resident closed-loop figure (N=1, quat only)")
    ap.add_argument("--no-large-batch", action="store_true",
                    help="skip the secondary legs on the large-batch configurations (B=32768 N=10, B=65536 N=20; N=1, quat only)")
    ap.add_argument("--no-config4", action="store_true",
                    help="multi-rank runs: skip the extra leg on BASELINE config 4 (262144 instances over the ranks)")
    ap.add_argument("--check", action="store_true", help="also compare a sample against the oracle")
    ap.add_argument("--force-dist", action="store_true",
                    help="DIAGNOSTIC: run the multi-rank code path (process group, async all_gather per step, barriers) "
                         "even with one rank, to exercise the RCCL calls on a single GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU of this node, RCCL over xGMI)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: WORLD_SIZE={world} != --gpus {args.gpus}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    multi = world > 1 or args.force_dist        # the collective path is on
    if os.environ.get("QMPC_BENCH_DRYRUN"):
        # launcher check without GPUs (tests/test_bench_cpu.py): rendezvous over gloo, one collective, one JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "sum_of_ranks": float(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")     # only a bare `--force-dist` run lacks the launcher's value
        dist.init_process_group("nccl", rank=rank, world_size=world)

    pkg = load_pkg()
    N, B = args.horizon, args.batch
    convex = args.model == "convex"
    biped = args.model == "biped8"
    NU = 24 if biped else 12
    config_id = 5 if biped else ((12 if N == 10 else 13) if convex else (2 if N == 10 else 3))
    gen = pkg.random_biped8_states if biped else (pkg.random_go1_convex_states if convex else pkg.random_go1_trot_states)
    IW = pkg.INFO_DTYPE.itemsize // 8           # qmpc_info is 40 bytes = 5 doubles
    assert pkg.INFO_DTYPE.itemsize == 8 * IW
    SLOTS = int(os.environ.get("QMPC_BENCH_SLOTS", "2"))   # result blocks in rotation (gather of step i drains under later solves)
    lib = pkg.load_library()
    params = (pkg.default_biped8_params if biped else
              (pkg.default_convex_params if convex else pkg.default_params))(N, pkg.MODE_CONVERGED, lib)
    # a real (non-null) stream: the C ABI treats NULL as "the handle's own stream"
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    def oracle_solve(po, records, threads=1):
        if biped:
            return po.solve8(po.default_biped8_params(N, 0), records, threads=threads)
        if convex:
            return po.convex_solve(po.default_convex_params(N, 0), records, threads=threads)
        return po.solve(po.default_params(N, 0), records, threads=threads)

    def timed_leg(Bl, cfg_id, steps, warmup, prm=None, model=None):
        """One leg of the contract: W untimed + K timed steps of `Bl` instances per rank (inputs resident in HBM),
        barrier + synchronize on both sides, max over ranks.  One result block per step slot holds
        [Bl x NU forces | Bl x qmpc_info] so that a single collective carries forces AND status."""
        prm = params if prm is None else prm
        model = args.model if model is None else model
        nu = 24 if model == "biped8" else 12
        g = {"biped8": pkg.random_biped8_states, "convex": pkg.random_go1_convex_states,
             "quat": pkg.random_go1_trot_states}[model]
        # synthetic states; rank r owns instances [r*Bl, (r+1)*Bl)
        rec = g(Bl, config_id=cfg_id, first=rank * Bl)
        solver = pkg.Solver(prm, Bl, device=local, lib=lib)
        d_in = torch.from_numpy(rec.view(np.uint8).reshape(Bl, -1).copy()).cuda()
        pipe = pkg.StepPipeline(world, rank, Bl * (nu + IW), "cuda", slots=SLOTS, multi=multi)
        fn = {"biped8": solver.solve8_device, "convex": solver.convex_solve_device, "quat": solver.solve_device}[model]

        def launch(blk):
            fn(Bl, d_in.data_ptr(), blk[:Bl * nu].data_ptr(), blk[Bl * nu:].data_ptr(), stream.cuda_stream)

        for i in range(warmup):
            pipe.step(i, launch)
        pipe.drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        flush_c_stdio()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pipe.reset_stats()
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(steps):
            pipe.step(i, launch)
        ev1.record(stream)
        pipe.drain()
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0          # this rank's own clock, before the closing barrier
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # HIP events on the launch stream: average launch duration (back-to-back launches; in a multi-rank run the waits for
        # gathers that had not drained sit between the launches and are inside this figure: gather_wait_ms says how much)
        kernel_ms = ev0.elapsed_time(ev1) / steps
        pst = pipe.stats()
        ranks = None
        if multi:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # diagnosis of a scaling run: every rank's kernel time per step, the time its solve stream was held by gathers
            # that had not drained, and its own wall clock -- min / max over the ranks, gathered with one small collective
            mine = torch.tensor([kernel_ms, pst["gather_wait_ms"] / steps, 1e3 * own_elapsed / steps], dtype=torch.float64, device="cuda")
            allr = torch.zeros(world, 3, dtype=torch.float64, device="cuda")
            dist.all_gather_into_tensor(allr.view(-1), mine)
            a_ = allr.cpu().numpy()
            ranks = {"kernel_ms": {"min": float(a_[:, 0].min()), "max": float(a_[:, 0].max()), "argmax": int(a_[:, 0].argmax())},
                     "gather_wait_ms": {"min": float(a_[:, 1].min()), "max": float(a_[:, 1].max())},
                     "own_ms_per_step": {"min": float(a_[:, 2].min()), "max": float(a_[:, 2].max())},
                     "gathers_per_step": pst["gathers"] / steps,
                     "gather_bytes_per_step": int(world * Bl * (nu + IW) * 8)}
        last = pipe.block(steps - 1)
        allb = pipe.all_blocks(steps - 1)
        if multi:   # every rank holds every rank's forces and status: the gathered block must carry the local one
            assert torch.equal(allb[rank], last), "gathered block differs from the local shard"
        # status of EVERY rank's instances (the gathered blocks carry them)
        info = np.ascontiguousarray(allb[:, Bl * nu:].cpu().numpy()).view(pkg.INFO_DTYPE).reshape(world, Bl)
        forces = last[:Bl * nu].view(Bl, nu).cpu().numpy().copy()
        return {"elapsed": elapsed, "kernel_ms": kernel_ms, "info": info, "forces": forces, "rec": rec,
                "solver": solver, "d_in": d_in, "ranks": ranks, "gather_wait_ms": pst["gather_wait_ms"] / steps}

    def kernel_name(batch, horizon=None, slv=None):
        """dominant kernel of a launch of `batch` instances: the LIBRARY's own answer (qmpc_query: launch_solve's choice for
        this handle, incl. whether the straggler hand-off is available), not a re-derivation of its thresholds"""
        hz = N if horizon is None else horizon
        own = slv is None
        if own:
            prm_ = (pkg.default_biped8_params if biped else (pkg.default_convex_params if convex else pkg.default_params))(hz, pkg.MODE_CONVERGED, lib)
            slv = pkg.Solver(prm_, batch, device=local, lib=lib)
        fam = slv.kernel_for_batch(batch)
        cap = slv.query(pkg.QUERY_LANE_CAP, 1)
        if own:
            slv.close()
        # the family is the library's answer; which instantiation of it runs (WVAR 5 / 6, model) is in profiles/BENCH_NOTES.md
        return {"lane_handoff": f"qmpc_lane_kernel + qmpc_solve_w_list_kernel beyond {cap} iterations",
                "lane": "qmpc_lane_kernel",
                "wform_lds": "wrench-form wave kernel, all LDS",
                "wform_ws": "wrench-form wave kernel, workspace form",
                "dense_lds": "dense wave kernel, all LDS",
                "dense_ws": "dense wave kernel, workspace form"}[fam]

    def roofline_object(kname, ach, tr, tr_src, kms, batch, compulsory):
        """`frac` prices SURVEY 8d's algorithmic flops against the 78.6 TFLOP/s FP64 peak (the matrix and the vector FP64
        rates of gfx950 are the same units).  `bound` says which instructions issue them in the dominant kernel:
        "fp64 (valu+mfma)" for the wave-per-instance kernels (FP64 MFMAs carry the row-mixing stage products, ~40 % of the
        flops; the matrix pipe is 12-13 % busy), "fp64_valu" for the lane-per-instance kernel, whose ISA holds no matrix
        instruction (mfma_busy 0)."""
        lane = kname.startswith("qmpc_lane_kernel")
        return {"bound": "fp64_valu" if lane else "fp64 (valu+mfma)", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_PEAK_TFLOPS, "traffic": tr, "kernel": kname, "kernel_ms": kms,
                "algorithmic_bytes_per_launch": compulsory,
                "hbm_GBps": (tr / (kms * 1e-3) / 1e9) if tr else None,
                "traffic_to_compulsory": (tr / compulsory) if tr else None}

    leg = timed_leg(B, config_id, args.steps, args.warmup)
    elapsed, kernel_ms, info, rec = leg["elapsed"], leg["kernel_ms"], leg["info"], leg["rec"]
    solver, d_in, d_f = leg["solver"], leg["d_in"], leg["forces"]
    n_ok = int((info["status"] == 0).sum())
    mean_iters = float(info["iterations"].mean())

    # BASELINE config 4 beside the weak-scaling value: 262144 Go1 instances, N=10, sharded over the ranks
    # (32768 per GPU at 8 GPUs), same pipeline, a few steps.  Every rank takes part (collectives).
    config4 = None
    if (world > 1 or (args.force_dist and os.environ.get("QMPC_BENCH_FORCE_CONFIG4"))) and not args.no_config4 and args.model == "quat":
        B4 = 262144 // world
        p4 = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
        k4 = max(2, min(5, args.steps))
        leg4 = timed_leg(B4, 4, k4, 1, prm=p4, model="quat")
        kn4 = kernel_name(B4, 10, slv=leg4["solver"]) if rank == 0 else ""
        ach4 = W_ALG_KFLOP_PER_KNOT * 1e3 * 10 * B4 / (leg4["kernel_ms"] * 1e-3) / 1e12
        tr4, _ = traffic_from_profiles(B4, 10, "quat")
        config4 = {"workload": f"config 4: 262144 Go1 states, N=10, x{world} ({B4} per GPU), one all_gather per step",
                   "value": world * B4 * k4 / leg4["elapsed"], "unit": "solves/s", "steps": k4, "warmup": 1,
                   "ms_per_step": 1e3 * leg4["elapsed"] / k4, "kernel_ms": leg4["kernel_ms"],
                   "instances": world * B4, "converged": int((leg4["info"]["status"] == 0).sum()),
                   "mean_iterations": float(leg4["info"]["iterations"].mean()),
                   "roofline": roofline_object(kn4, ach4, tr4, None, leg4["kernel_ms"], B4, B4 * (8 * 48 + 8 * 12 + 40)),
                   "ranks": leg4["ranks"]}
        leg4["solver"].close()
        del leg4

    if rank == 0:
        traffic, traffic_src = traffic_from_profiles(B, N, args.model)
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
            "config": {"workload": f"Batch={B} random {'biped8 ' if biped else 'Go1 '}{'ConvexMpc ' if convex else ''}states per GPU, N={N}, "
                                   f"converged mode, seed 0x5EED0000+{config_id}, device-resident",
                       "value_is": "rates.device_resident",
                       "batch_per_gpu": B, "horizon": N, "parallelism": f"instance-sharded x{world}",
                       "instances": world * B, "converged": n_ok, "mean_iterations": mean_iters},
            "rates": {"device_resident": value},
            "roofline": roofline_object(kernel_name(B, slv=solver), achieved, traffic, traffic_src, kernel_ms, B,
                                        B * (8 * (64 if biped else 48) + 8 * NU + 40)),
        }
        out["roofline"]["algorithmic_flops_per_launch"] = w_alg * B
        out["notes"] = "profiles/BENCH_NOTES.md"          # what every field means, counter sources, kernel instantiations
        sq = sq_from_profiles(B, N, args.model)
        if sq is not None:
            out["roofline"]["mfma_busy"] = sq.get("mfma_busy")
            out["roofline"]["issue_frac"] = sq.get("issue_frac")
        if multi:
            out["ranks"] = leg["ranks"]
        if config4 is not None:
            out["config4"] = config4
        if args.check:
            from oracle import pyoracle
            idx = np.arange(0, B, max(B // 64, 1))
            fo, _ = oracle_solve(pyoracle, rec[idx], threads=usable_cores())
            out["config"]["force_linf_vs_cpu"] = float(np.abs(d_f[idx] - fo).max())
        if world == 1:
            # the drop-in entry point as the controller calls it: HOST buffers, H2D + kernel + D2H, blocking
            # (SURVEY 8d's definition of the metric; reported beside the resident figure, never `value`).  Timed through
            # the C ABI on caller-owned buffers, as a C++ host calls it: once with pinned buffers (qmpc_host_alloc: the
            # library reads records / writes results in place, zero-copy) and once with pageable ones (staged by the handle)
            hname = "qmpc_solve8" if biped else ("qmpc_convex_solve" if convex else "qmpc_solve")
            hfn = getattr(lib, hname)
            reps = max(20, min(200, args.steps * 4))
            hb = {}
            for kind in ("pinned", "pageable"):
                if kind == "pinned":
                    hin = solver.pinned((B, rec.dtype.itemsize), np.uint8)
                    hf = solver.pinned((B, NU)); hi = solver.pinned((B,), pkg.INFO_DTYPE)
                else:
                    hin = np.empty((B, rec.dtype.itemsize), np.uint8)
                    hf = np.zeros((B, NU)); hi = np.zeros(B, dtype=pkg.INFO_DTYPE)
                hin[...] = rec.view(np.uint8).reshape(B, -1)
                args_ = (solver._h, B, hin.ctypes.data, hf.ctypes.data, hi.ctypes.data)
                for _ in range(3):
                    assert hfn(*args_) == 0
                t0 = time.perf_counter()
                for _ in range(reps):
                    hfn(*args_)
                dt = time.perf_counter() - t0
                assert np.array_equal(hf, d_f) and (hi["status"] == 0).sum() == n_ok, "host-buffer call differs from the resident one"
                hb[kind] = {"value": B * reps / dt, "unit": "solves/s", "ms_per_call": 1e3 * dt / reps, "calls": reps}
            out["rates"]["host_buffer_call"] = hb["pinned"]["value"]
            out["rates"]["host_buffer_call_pageable"] = hb["pageable"]["value"]
            out["rates"]["host_buffer_ms_per_call"] = hb["pinned"]["ms_per_call"]
            out["value_host_inclusive"] = hb["pinned"]["value"]     # SURVEY 8d's own definition of the metric
            out["host_inclusive_over_resident"] = out["value_host_inclusive"] / value
        if world == 1 and args.model == "quat" and not args.no_large_batch and B == 1024 and N == 10:
            # secondary: the large-batch configurations of BASELINE.json on this GPU (the lane-per-instance kernel):
            # the per-GPU share of config 4 (32768 instances, N=10) and config 3 (65536 instances, N=20); never `value`
            out["large_batch"] = []
            for (Bl, Nl, cfg, what) in ((32768, 10, 4, "per-GPU share of config 4"), (65536, 20, 3, "config 3")):
                pl_ = pkg.default_params(Nl, pkg.MODE_CONVERGED, lib)
                kl = max(3, min(10, args.steps))
                lg = timed_leg(Bl, cfg, kl, 2, prm=pl_, model="quat")
                # secondary inside the secondary: consecutive independent batches on two streams (two handles) -- the tail of
                # one launch (its slowest wavefront) overlaps the head of the next
                s2 = pkg.Solver(pl_, Bl, device=local, lib=lib)
                st2 = [torch.cuda.Stream(), torch.cuda.Stream()]
                o2 = [torch.zeros(Bl, 12, dtype=torch.float64, device="cuda") for _ in range(2)]
                i2 = [torch.zeros(Bl, pkg.INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)]
                hs = [lg["solver"], s2]
                torch.cuda.synchronize()
                for w_ in range(2):
                    hs[w_].solve_device(Bl, lg["d_in"].data_ptr(), o2[w_].data_ptr(), i2[w_].data_ptr(), st2[w_].cuda_stream)
                torch.cuda.synchronize()
                t02 = time.perf_counter()
                for w_ in range(2 * kl):
                    j_ = w_ % 2
                    hs[j_].solve_device(Bl, lg["d_in"].data_ptr(), o2[j_].data_ptr(), i2[j_].data_ptr(), st2[j_].cuda_stream)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t02
                same2 = bool(torch.equal(o2[0], o2[1]))
                s2.close()
                del o2, i2
                kn_l = kernel_name(Bl, Nl, slv=lg["solver"])
                lg["solver"].close()
                w_l = W_ALG_KFLOP_PER_KNOT * 1e3 * Nl
                ach = w_l * Bl / (lg["kernel_ms"] * 1e-3) / 1e12
                tr, tr_src = traffic_from_profiles(Bl, Nl, "quat")
                ent = {"workload": f"B={Bl} N={Nl}: {what}",
                       "value": Bl * kl / lg["elapsed"], "unit": "solves/s", "steps": kl, "ms_per_step": 1e3 * lg["elapsed"] / kl,
                       "converged": int((lg["info"]["status"] == 0).sum()), "mean_iterations": float(lg["info"]["iterations"].mean()),
                       "two_in_flight": {"value": Bl * 2 * kl / dt2, "outputs_identical": same2},
                       "roofline": roofline_object(kn_l, ach, tr, tr_src, lg["kernel_ms"], Bl, Bl * (8 * 48 + 8 * 12 + 40))}
                # the reference's own solver mode on the same batch (AL-iLQR, <= 10 iterations): the lane kernel's AL passes
                # from 19456 instances on (14848 beyond N=12; lane pairs below 32769), the wave-per-instance reference kernels below -- the library says which
                if not args.no_reference_mode:
                    prl = pkg.default_params(Nl, pkg.MODE_REFERENCE, lib)
                    srl = pkg.Solver(prl, Bl, device=local, lib=lib)
                    frl = torch.zeros(Bl, 12, dtype=torch.float64, device="cuda")
                    irl = torch.zeros(Bl, pkg.INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
                    d_inl = torch.from_numpy(pkg.random_go1_trot_states(Bl, config_id=cfg).view(np.float64).reshape(Bl, -1).copy()).cuda()
                    kms = []
                    for r_ in range(4):
                        srl.solve_device(Bl, d_inl.data_ptr(), frl.data_ptr(), irl.data_ptr())
                        srl.wait()
                        if r_:
                            kms.append(srl.last_kernel_ms())
                    inf_l = irl.cpu().numpy().view(pkg.INFO_DTYPE).reshape(Bl)
                    fam_l = srl.kernel_for_batch(Bl)
                    srl.close()
                    ent["reference_mode"] = {
                        "value": Bl / (float(np.median(kms)) * 1e-3), "kernel_ms": float(np.median(kms)),
                        "kernel": "qmpc_lane_ref_kernel" if fam_l == "lane" else "qmpc_ref_w_kernel (workspace form)",
                        "mean_iterations": float(inf_l["iterations"].mean())}
                    del frl, irl, d_inl
                sql = sq_from_profiles(Bl, Nl, "quat")
                if sql is not None:
                    ent["roofline"]["issue_frac"] = sql.get("issue_frac")
                out["large_batch"].append(ent)
                del lg
            # secondary: the other workloads the wrench-form wave kernels serve since round 5 -- the per-GPU share of BASELINE
            # config 5 (8-point model, N=16, 65536 instances over 8 GPUs), ConvexMpc at its own configuration, and QuatMpc at
            # the reference's horizon (N=20) at a mid-size batch and for the single robot; never `value`
            out["other_workloads"] = []
            for (mdl, Nl, Bl, cfg, what, *md) in (
                    ("biped8", 16, 8192, 5, "per-GPU share of config 5"),
                    ("convex", 20, 1024, 13, "ConvexMpc, YAML horizon"),
                    ("quat", 20, 8192, 3, "reference's horizon, mid-size"),
                    ("quat", 20, 1, 3, "one robot"),
                    ("convex", 20, 65536, 13, "ConvexMpc, own solver mode", 1),
                    ("biped8", 16, 65536, 5, "8-point model, reference mode", 1)):
                mode_ = pkg.MODE_REFERENCE if md and md[0] else pkg.MODE_CONVERGED
                pm_ = {"biped8": pkg.default_biped8_params, "convex": pkg.default_convex_params, "quat": pkg.default_params}[mdl](Nl, mode_, lib)
                kl = max(3, min(10, args.steps))
                lg = timed_leg(Bl, cfg, kl, 2, prm=pm_, model=mdl)
                fam = lg["solver"].kernel_for_batch(Bl)
                lg["solver"].close()
                out["other_workloads"].append({
                    "workload": f"{mdl} B={Bl} N={Nl}: {what}", "value": Bl * kl / lg["elapsed"],
                    "ms_per_step": 1e3 * lg["elapsed"] / kl, "kernel_family": fam,
                    "mean_iterations": float(lg["info"]["iterations"].mean())})
                del lg
        if world == 1 and not args.no_in_flight:
            out["two_in_flight"] = two_in_flight(pkg, lib, params, args, d_in, NU)
        if world == 1 and args.model in ("quat", "convex") and not args.no_reference_mode:
            # secondary: the reference's OWN operating mode on the device (QMPC_MODE_REFERENCE: AL-iLQR, 10 iterations,
            # penalty scaling 20, status ignored; QuatMpc.cpp:21-26,256) on the same batch, and how far the truncated
            # iterate it returns is from the converged KKT point that `value` computes; never `value`
            pr = (pkg.default_convex_params if convex else pkg.default_params)(N, pkg.MODE_REFERENCE, lib)
            sr = pkg.Solver(pr, B, device=local, lib=lib)
            fr = torch.zeros(B, NU, dtype=torch.float64, device="cuda")
            ir = torch.zeros(B, pkg.INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
            rfn = sr.convex_solve_device if convex else sr.solve_device
            for _ in range(max(2, args.warmup)):
                rfn(B, d_in.data_ptr(), fr.data_ptr(), ir.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                rfn(B, d_in.data_ptr(), fr.data_ptr(), ir.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ri = np.ascontiguousarray(ir.cpu().numpy()).view(pkg.INFO_DTYPE).reshape(B)
            du = np.abs(fr.cpu().numpy() - d_f).max(axis=1)
            sr.close()
            out["reference_mode"] = {
                "value": B * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                "iterations_max": int(pr.iterations_max), "mean_iterations": float(ri["iterations"].mean()),
                "status_counts": {"converged": int((ri["status"] == 0).sum()), "iteration_cap": int((ri["status"] == 1).sum()),
                                  "linesearch_fail": int((ri["status"] == 4).sum()), "not_pd": int((ri["status"] == 5).sum())},
                "u0_distance_to_converged_N": {"median": float(np.median(du)), "max": float(du.max())}}
        if world == 1 and args.model == "quat" and not args.no_closed_loop:
            # secondary: the device-resident closed loop (front end + solve + plant per tick, state in HBM) for B robots
            # with DIFFERENT commands (the Monte-Carlo use; 10 % stand): 8 stand ticks, 40 ticks into the gait, then 100
            # timed ticks; never `value`
            lp = pkg.default_loop_params(lib)
            rng = np.random.default_rng(11)
            cmds = np.zeros((B, 7))
            cmds[:, 0] = rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
            cmds[:, 5] = rng.uniform(-0.5, 0.5, B)
            walk = rng.random(B) < 0.9
            cmds[~walk, :2] = 0.0
            cmds[~walk, 5] = 0.0
            st = pkg.loop_states(cmds, lp, height=0.3, yaw=rng.uniform(-3.1, 3.1, B), lib=lib)     # movement_mode 0: stand
            pl = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
            pl.drop_ang_vel = 0          # the MPC sees the body's angular velocity (with the reference's x_init quirk the
            sl = pkg.Solver(pl, B, device=local, lib=lib)   # ideal plant is undamped: DESIGN 3e); its own handle
            sl_form = B <= 2048 and os.environ.get("QMPC_LOOP_FUSED") != "0"
            st = sl.loop_run(st, 8, lp)
            st["movement_mode"] = walk.astype(float)
            d_st = torch.from_numpy(st.view(np.uint8).reshape(B, -1).copy()).cuda()
            ticks = 100
            sl.loop_run_device(B, d_st.data_ptr(), 40, lp, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sl.loop_run_device(B, d_st.data_ptr(), ticks, lp, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            # the same robots once more with the loop's own options: warm start + a low initial barrier (same forces, about
            # half the iterations)
            pw = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
            pw.drop_ang_vel = 0
            pw.ipm_mu0 = 1e-6
            lpw = pkg.default_loop_params(lib)
            lpw.warm_start = 1.0
            sw = pkg.Solver(pw, B, device=local, lib=lib)
            d_sw = torch.from_numpy(st.view(np.uint8).reshape(B, -1).copy()).cuda()
            sw.loop_run_device(B, d_sw.data_ptr(), 40, lpw, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            t0w = time.perf_counter()
            sw.loop_run_device(B, d_sw.data_ptr(), ticks, lpw, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            dtw = time.perf_counter() - t0w
            finw = np.ascontiguousarray(d_sw.cpu().numpy()).view(pkg.LOOP_STATE_DTYPE).reshape(B)
            sw.close()
            sl.close()
            fin = np.ascontiguousarray(d_st.cpu().numpy()).view(pkg.LOOP_STATE_DTYPE).reshape(B)
            out["closed_loop"] = {"value": B * ticks / dt, "unit": "robot-ticks/s", "ticks": ticks, "robots": B,
                                  "solver_ok": int((fin["status"] == 0).sum()),
                                  "mean_iterations": float(fin["iterations"].mean()),
                                  "warm_start": {"value": B * ticks / dtw, "solver_ok": int((finw["status"] == 0).sum()),
                                                 "mean_iterations": float(finw["iterations"].mean()),
                                                 "position_difference_to_cold_start_m": float(np.abs(finw["pos_world"] - fin["pos_world"]).max())},
                                  "launch_form": loop_form(sl_form)}
            if B == 1024 and N == 10 and not args.no_large_batch:
                # ... and the Monte-Carlo scale: 65536 robots, per-tick launch form, the solves on the lane-per-instance
                # kernel (cold start); 8 stand ticks, 30 ticks into the gait, 40 timed ticks
                BL = 65536
                cl = np.zeros((BL, 7))
                cl[:, 0] = rng.uniform(-0.5, 0.5, BL); cl[:, 1] = rng.uniform(-0.2, 0.2, BL); cl[:, 2] = rng.uniform(0.26, 0.32, BL)
                cl[:, 5] = rng.uniform(-0.5, 0.5, BL)
                wl = rng.random(BL) < 0.9
                cl[~wl, :2] = 0.0
                cl[~wl, 5] = 0.0
                stl = pkg.loop_states(cl, lp, height=0.3, yaw=rng.uniform(-3.1, 3.1, BL), lib=lib)
                sll = pkg.Solver(pl, BL, device=local, lib=lib)
                stl = sll.loop_run(stl, 8, lp)
                stl["movement_mode"] = wl.astype(float)
                d_stl = torch.from_numpy(stl.view(np.uint8).reshape(BL, -1).copy()).cuda()
                sll.loop_run_device(BL, d_stl.data_ptr(), 30, lp, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                tl = 40
                t0l = time.perf_counter()
                sll.loop_run_device(BL, d_stl.data_ptr(), tl, lp, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                dtl = time.perf_counter() - t0l
                finl = np.ascontiguousarray(d_stl.cpu().numpy()).view(pkg.LOOP_STATE_DTYPE).reshape(BL)
                sll.close()
                swl = pkg.Solver(pw, BL, device=local, lib=lib)      # the same robots, warm-started
                d_swl = torch.from_numpy(stl.view(np.uint8).reshape(BL, -1).copy()).cuda()
                swl.loop_run_device(BL, d_swl.data_ptr(), 30, lpw, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                t0lw = time.perf_counter()
                swl.loop_run_device(BL, d_swl.data_ptr(), tl, lpw, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                dtlw = time.perf_counter() - t0lw
                finlw = np.ascontiguousarray(d_swl.cpu().numpy()).view(pkg.LOOP_STATE_DTYPE).reshape(BL)
                swl.close()
                out["closed_loop"]["large"] = {"value": BL * tl / dtl, "robots": BL, "ticks": tl,
                                               "solver_ok": int((finl["status"] == 0).sum()),
                                               "mean_iterations": float(finl["iterations"].mean()),
                                               "warm_start": {"value": BL * tl / dtlw,
                                                              "solver_ok": int((finlw["status"] == 0).sum()),
                                                              "mean_iterations": float(finlw["iterations"].mean())},
                                               "launch_form": "per-tick graph, lane kernel + hand-off"}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(pkg, N, config_id, model=args.model)
            f_cpu = cb.pop("_forces")
            m = min(len(f_cpu), B)      # the sample starts with the instances of the timed batch
            cb["force_linf_gpu_vs_cpu"] = float(np.abs(d_f[:m] - f_cpu[:m]).max())
            cb["force_linf_instances"] = m
            out["cpu_baseline"] = cb
        flush_c_stdio()
        sys.stdout.flush()
        print(json.dumps(compact(out), separators=(",", ":")), flush=True)
    solver.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
