#!/usr/bin/env python3
"""bench.py - MLUPS of the collide-and-stream hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (pull-stream + regularised-BGK/WALE collide; no Bouzidi cells in this workload)
over every cell of a synthetic uniform periodic box, inputs resident in HBM when the timed region starts.

  N = 1 : BASELINE configs[1] - 256^3 periodic box, D3Q27 reg-BGK + WALE, FP32 (SURVEY.md section 8d "C2").
  N > 1 : --scaling weak (default): every rank owns 256^3 cells of one global periodic box (2, 4 ranks: cubic bricks cut in z,
          then y; 8 ranks = configs[3], the 512^3 box, cut 1 x 2 x 4 into bricks of 512 x 256 x 128 cells:
          partition.weak_scaling_layout);
          --scaling strong [--size 512]: the SAME global box at every N - configs[3] as BASELINE states it, "512^3 ... 1/2/4/8-GPU
          scaling" - cut 1x1x2, 1x2x2, 1x2x4 (partition.strong_scaling_layout; N = 1 steps the whole box on one GPU).
          One-cell halo of f and u exchanged every step over RCCL (torch.distributed backend "nccl").

`python3 bench.py --gpus N` starts its own N ranks: when WORLD_SIZE is not in the environment and N > 1, a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` runs this file with the same arguments
(a child process started before anything touches the GPU - never an exec), rank 0's JSON line and the child's exit code are relayed.
Launched under torch.distributed.run by somebody else (WORLD_SIZE set), it is a rank. LUDWIG_BENCH_FORCE_DEVICE=0 rehearses N ranks on
ONE GPU over gloo with host-staged messages (RCCL refuses two ranks on a device); --plan-only stops after the rendezvous and the halo
plan (no GPU needed) and prints the `comm` object with "value": null.

Prints ONE JSON line (rank 0). `roofline` prices the stream-collide kernel against the 8 TB/s HBM peak with the
algorithmic 216 B per lattice update; `cpu_baseline` is the CPU oracle (a port, NOT the reference's Julia CPU
path, which cannot run here) timed on a bounded 64^3 sample of the same workload.

Before the W warm-up steps the device is kept busy for --preheat-ms (default 60 ms) with plain device-to-device copies of two
scratch buffers - no stepping - because the step time shows a 25-step power-management transient after any idle period and the set-up
leaves the device idle; `config.device_preheat_ms` reports it, `--preheat-ms 0` turns it off.

Kernel time: HIP events on the launch stream around the timed region, divided by K (one launch per step at N = 1); every
--event-every-th step (default 10) is bracketed on its own as well, for the spread. stdout carries the JSON line only.

`roofline.traffic` comes from rocprofv3 PMC passes, which cannot run inside this process: it is read from
profiles/traffic.json and reported ONLY if that file was captured with the very sources the loaded library was built from
(`source_digest`) on the same workload size; otherwise it is null and `traffic_source` says why. Capture: tools/final_profile.sh.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_LUP = 216.0        # 27 x 4 B read + 27 x 4 B write
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --size^3 cells per GPU whatever N; strong: one global box of --size^3 cells cut over the N GPUs")
    ap.add_argument("--size", type=int, default=None,
                    help="cells per side: of the per-GPU brick (weak, default 256) or of the global box (strong, default 512)")
    ap.add_argument("--plan-only", action="store_true",
                    help="N > 1: rendezvous + halo plan only, on the CPU over gloo; prints the comm object with value null (no measurement)")
    ap.add_argument("--eager-rho-steps", type=int, default=20,
                    help="N = 1: after the timed region, time this many steps with rho stored by every step as the reference's kernel does "
                         "(roofline.frac_eager_rho); 0 = skip")
    ap.add_argument("--order", default=None, help="launch-order builder (open_ludwig_amd/order.py); default = library default")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--cpu-size", type=int, default=64)
    ap.add_argument("--no-overlap", action="store_true", help="multi-GPU: exchange halos after the whole step")
    ap.add_argument("--event-every", type=int, default=10,
                    help="bracket every N-th step with its own pair of HIP events (an event between two launches costs the stream a few "
                         "microseconds); the timed region as a whole is always bracketed")
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="keep the device busy with plain memory copies (no stepping) for this long right before the warm-up steps, so that "
                         "the W + K steps run at settled clocks (0 = off)")
    a = ap.parse_args()
    if a.size is None:
        a.size = 512 if a.scaling == "strong" else 256
    return a


def launch_ranks(n: int) -> int:
    """`python3 bench.py --gpus N` from a plain shell: run the N ranks as a CHILD torch.distributed.run (this process has not touched
    the GPU and never does), relay rank 0's JSON line - the only thing that reaches stdout - and the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for l in res.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        elif l.strip():
            print(l, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return res.returncode if res.returncode else (0 if line is not None else 1)


def host_cores() -> int:
    """Cores this process may really use: affinity, capped by the cgroup CPU quota and by the 16-core share a 1-GPU box
    grants (the box reports 256 logical CPUs but schedules a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("LUDWIG_BENCH_CPU_CORES", "16"))))


def cpu_baseline(size: int, seconds: float):
    """Oracle (a port of the reference's arithmetic, all host cores via OpenMP) on a size^3 periodic Taylor-Green box
    with the bench parameters; steps until the time budget is used."""
    import numpy as np
    from open_ludwig_amd import cases
    from oracle import oracle
    nb = size // 8
    grids, params = cases.periodic_box((nb, nb, nb))
    cores = host_cores()
    oracle.set_num_threads(cores)
    oracle.execute_timestep_batch(grids, 1, 1, np.float32(0.0), params)   # touch pages
    t0 = time.perf_counter()
    steps = 0
    t = 2
    while time.perf_counter() - t0 < seconds:
        oracle.execute_timestep_batch(grids, t, 2, np.float32(0.0), params)
        t += 2
        steps += 2
    dt = time.perf_counter() - t0
    return {"value": round(size ** 3 * steps / dt / 1e6, 3), "unit": "MLUPS", "cores": cores, "kind": "port",
            "sample": f"{size}^3 periodic box, same parameters, {steps} steps in {dt:.1f} s (CPU oracle, OpenMP)"}


def workload_text(args, world, brick, rgrid, box, rehearsal: bool = False) -> str:
    base = "uniform periodic box, D3Q27 regularized-BGK + WALE, Taylor-Green start (SURVEY 8d C2)"
    if world == 1:
        return f"{base}; {box[0]}^3 cells on one GPU" + (" (strong-scaling base: BASELINE configs[3] on one GPU)" if args.scaling == "strong" else "")
    cut = (f"{world} bricks of {brick[0] * 8}x{brick[1] * 8}x{brick[2] * 8} cells in a {rgrid[0]}x{rgrid[1]}x{rgrid[2]} rank grid, "
           "one-cell halo of f,u per step " + ("over gloo with host-staged messages (one-GPU rehearsal)" if rehearsal else "over RCCL"))
    if args.scaling == "strong":
        return f"{base}; STRONG scaling: fixed global box of {box[0]}x{box[1]}x{box[2]} cells (BASELINE configs[3] at --size 512), {cut}"
    return f"{base}; WEAK scaling: {args.size}^3 cells per GPU, global box {box[0]}x{box[1]}x{box[2]} cells, {cut}"


def plan_only(args, world, rank, brick, rgrid, json_fd) -> None:
    """--plan-only: the N ranks meet over gloo on the CPU, build and exchange the halo plan of the workload, and rank 0 prints the
    comm object. No GPU, no stepping, no number: "value" is null."""
    import numpy as np
    import torch.distributed as dist
    from open_ludwig_amd import partition
    if world < 2:
        raise SystemExit("--plan-only is for --gpus N > 1")
    dist.init_process_group("gloo")
    view, plan, _, n_global = partition.periodic_box_plan(rank, world, brick, rgrid, init=False)
    peers = [p for p in plan.peers if p != rank]
    per_peer = {int(p): 4 * sum(a.size for a in plan.send[p].values()) for p in peers}
    stats = [None] * world
    dist.all_gather_object(stats, {"halo_bytes": plan.bytes_per_step(), "peers": len(peers), "owned": int(view.n_owned), "blocks": int(view.level.n_blocks)})
    if rank == 0:
        box = tuple(8 * brick[i] * rgrid[i] for i in range(3))
        out = {"metric": f"MLUPS (million lattice updates/s) at {args.size}^3 D3Q27; % of HBM roofline", "value": None, "unit": "MLUPS",
               "n_gpus": world, "plan_only": True, "scaling": args.scaling,
               "config": {"workload": workload_text(args, world, brick, rgrid, box), "scaling_mode": args.scaling, "global_box_cells": list(box),
                          "cells_per_gpu": 512 * brick[0] * brick[1] * brick[2], "global_blocks": int(n_global)},
               "comm": {"backend": dist.get_backend(), "backend_world_size": dist.get_world_size(), "rank_grid": list(rgrid), "blocks_per_rank": list(brick),
                        "halo_bytes_per_rank_per_step": [s["halo_bytes"] for s in stats], "peers_per_rank": [s["peers"] for s in stats],
                        "owned_blocks_per_rank": [s["owned"] for s in stats], "view_blocks_per_rank": [s["blocks"] for s in stats],
                        "bytes_to_each_peer_of_rank0": per_peer}}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def kernel_label(info) -> str:
    """the stepping kernel exactly as rocprofv3 --kernel-trace --stats prints it (profiles/*_kernel_stats.csv can be joined on it):
    template arguments NW, GENERAL, POST, WALL, RHO_ONLY, WIDE (64-bit per-lane addresses: levels whose f array is 4 GiB or more)"""
    general = "true" if info.n_general_blocks > 0 and info.n_fast_blocks == 0 else "false"
    wide = "true" if (info.n_blocks * 27 * 2048 >= 2 ** 32 or os.environ.get("LUDWIG_WIDE_ADDR")) else "false"
    return f"void lw::k_stream_collide_xrun<4, {general}, false, false, false, {wide}>(lw::SCParams)"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    # stdout carries ONE line, the JSON: libraries that chat on stdout (gloo's "connected to n peer ranks", RCCL's banner) are sent
    # to stderr at the file-descriptor level for the whole run; the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch   # first: its bundled HIP runtime is the one the whole process shares (same SONAME as /opt/rocm's)
    from open_ludwig_amd import _lib, adapt, cases, order as order_mod
    from open_ludwig_amd.physics import stream_collide

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = bool(os.environ.get("LUDWIG_BENCH_FORCE_DEVICE"))   # N > 1 on a 1-GPU box: gloo + host-staged messages
    if rehearsal:
        local_rank = int(os.environ["LUDWIG_BENCH_FORCE_DEVICE"])
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.size % 8:
        raise SystemExit("--size must be a multiple of the block size 8")
    nb = args.size // 8
    if world > 1:
        from open_ludwig_amd import partition
        brick, rgrid = (partition.strong_scaling_layout(world, nb) if args.scaling == "strong" else partition.weak_scaling_layout(world, nb))
    else:
        brick, rgrid = (nb, nb, nb), (1, 1, 1)
    cells_per_rank = 512 * brick[0] * brick[1] * brick[2]
    box = tuple(8 * brick[i] * rgrid[i] for i in range(3))
    if args.plan_only:
        return plan_only(args, world, rank, brick, rgrid, json_fd)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: --gpus {world} needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(LUDWIG_BENCH_FORCE_DEVICE=0 rehearses all ranks on one GPU over gloo)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")       # RCCL refuses two ranks on one device
        else:
            partition.init_rccl(local_rank)
    stream = torch.cuda.current_stream()

    if world == 1:
        grids, params = cases.periodic_box((nb, nb, nb), upload_only=True)
        level = adapt(grids[0], local_rank)
        coords = np.asarray(grids[0].active_block_coords)
        del grids
        level.set_stream(stream.cuda_stream)
        if args.order:
            level.set_order(order_mod.build(args.order, coords))
        runner = None

        def step(t):
            stream_collide(level, None, np.float32(0.5), np.float32(0.0), params, t)
    else:
        runner = partition.periodic_weak_scaling_box(rank, world, brick, device=local_rank,
                                                      overlap=not args.no_overlap, order=args.order,
                                                      stage_through_host=rehearsal, grid=rgrid)
        level = runner.level
        stream = runner.s_comp               # the stepping stream (leaves LUDWIG_COMM_RESERVED_CUS compute units to the exchange)
        if rank == 0:
            print(f"[bench] backend {dist.get_backend()} reports world size {dist.get_world_size()}; rank grid {rgrid}, {brick} blocks per rank; "
                  f"halo {runner.ex.plan.bytes_per_step() / 1e6:.2f} MB per rank per step", file=sys.stderr, flush=True)

        def step(t):
            runner.step(t)

    def barrier():
        if runner is not None:
            runner.flush()            # the last step's exchange is enqueued one launch late (DistributedLevelRunner.step): inside the timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The step time of this kernel shows a power-management transient whenever the device goes from idle to load: 0.73 ms for
    # a handful of steps, 0.80 ms around steps 6-10, 0.725 ms from step ~35 on (profiles/r02_step_time_transient_after_idle.txt) -
    # and the set-up above leaves the device idle (the upload is host-bound). With the driver's --warmup 5 --steps 20 the timed region
    # would sit exactly on that transient. So the device is kept busy with plain device-to-device copies of two scratch buffers - no
    # stepping, nothing of the workload - for --preheat-ms before the warm-up steps; reported in config.device_preheat_ms.
    if args.preheat_ms > 0:
        a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
            for _ in range(20):
                b.copy_(a)
            torch.cuda.synchronize()
        del a, b
    t = 1
    for _ in range(args.warmup):
        step(t); t += 1
    barrier()
    if runner is not None:
        runner.ex.exchange_ms()          # drop the warm-up exchanges
        runner.ex.timing = True
    # timed region: EXACTLY --steps steps
    # kernel time: HIP events on the launch stream around the WHOLE timed region (average launch duration = span / K; with one launch per
    # step and nothing else on the stream the span is the launches plus ~1 us between them) and, for the spread, around every
    # --event-every-th step on its own. Not around every step: an event record between two launches costs the stream 5-7 us.
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev = []
    every = max(1, args.event_every)
    t0 = time.perf_counter()
    region[0].record(stream)
    for i in range(args.steps):
        sampled = i % every == every // 2
        if sampled:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
        step(t); t += 1
        if sampled:
            b.record(stream)
            ev.append((a, b))
    region[1].record(stream)
    barrier()
    wall = time.perf_counter() - t0
    per_launch = [a.elapsed_time(b) for a, b in ev]
    kern_ms = region[0].elapsed_time(region[1]) / args.steps
    kern_med = float(np.median(per_launch)) if per_launch else kern_ms
    if runner is not None:
        # N > 1: a step is two launches (interior, boundary) with a wait for the halo in between: the per-step wall time of the
        # timed region stands in for the kernel time (an upper bound).
        kern_ms = kern_med = wall / args.steps * 1e3

    if dist is not None:
        red_dev = "cpu" if rehearsal else "cuda"
        w = torch.tensor([wall, kern_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)      # slowest rank defines the step time
        wall, kern_ms = float(w[0].item()), float(w[1].item())

    # N = 1, after the timed region: the same kernel with rho stored by every step, as the reference's kernel does
    # (src/physics_kernels.jl:243-246) - the driver-observed price of the elided store
    eager_ms = None
    if runner is None and args.eager_rho_steps > 0 and not os.environ.get("LUDWIG_EAGER_RHO"):
        level.set_rho_store(True)
        for _ in range(5):
            step(t); t += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.eager_rho_steps):
            step(t); t += 1
        e1.record(stream)
        torch.cuda.synchronize()
        eager_ms = e0.elapsed_time(e1) / args.eager_rho_steps

    # sanity: the state must still be finite and the flow non-trivial (no skipped work)
    rho = level.download("rho")
    ok = bool(np.isfinite(rho).all() and rho.std() > 0)
    del rho

    # multi-GPU: what the exchange cost, measured with events on the stream it ran on (diagnosis of the driver's scaling runs)
    comm = None
    if runner is not None:
        ms = runner.ex.exchange_ms()
        x = torch.tensor([float(np.mean(ms)) if ms else 0.0, float(np.max(ms)) if ms else 0.0], dtype=torch.float64,
                         device="cpu" if rehearsal else "cuda")
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        comm = {"backend": dist.get_backend(), "backend_world_size": dist.get_world_size(), "rank_grid": list(rgrid), "blocks_per_rank": list(brick),
                "halo_bytes_per_rank_per_step": runner.ex.plan.bytes_per_step(), "peers_of_rank0": len([p for p in runner.ex.plan.peers if p != rank]),
                "exchange_ms_mean_max_over_ranks": round(float(x[0].item()), 4), "exchange_ms_worst": round(float(x[1].item()), 4),
                "overlap": not args.no_overlap, "compute_units_left_to_the_exchange": runner.reserved_cus,
                "transport": ("librccl called from libludwig_hip.so (ludwig_step_distributed: ncclSend / ncclRecv per peer in one group)"
                              if runner.transport == "native" else "torch.distributed batch_isend_irecv from Python" + (" over gloo, host-staged" if rehearsal else "")),
                "note": "exchange = pack -> grouped send/recv -> unpack on a high-priority stream of its own, timed with events on that stream "
                        "(the span includes waiting beside the interior launch it runs under); queued at the end of the step that produced the "
                        "data, joined before the next step's boundary blocks; schedule verified over RCCL on one GPU by tests/test_rccl_loopback.py"}

    if rank == 0:
        total_cells = cells_per_rank * world
        ms_per_step = wall / args.steps * 1e3
        mlups = total_cells * args.steps / wall / 1e6
        achieved = ALGO_BYTES_PER_LUP * cells_per_rank / (kern_ms * 1e-3) / 1e9      # GB/s, one launch on one GPU
        from open_ludwig_amd import build as build_mod
        digest = build_mod.source_digest()
        traffic, traffic_source = None, "profiles/traffic.json absent"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("cells_per_launch") != cells_per_rank:
                    traffic_source = f"profiles/traffic.json is for {tj.get('cells_per_launch')} cells per launch, this run has {cells_per_rank}: not reported"
                elif tj.get("source_digest") != digest:
                    traffic_source = (f"profiles/traffic.json was captured with sources {tj.get('source_digest')}, the loaded library is {digest}: "
                                      "stale, not reported")
                elif args.order or world > 1 or os.environ.get("LUDWIG_WIDE_ADDR") or os.environ.get("LUDWIG_EAGER_RHO"):
                    traffic_source = "profiles/traffic.json is for the default single-GPU launch: not reported for this configuration"
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = (f"replayed from profiles/traffic.json (rocprofv3 PMC passes of tools/final_profile.sh, captured {tj.get('captured', '?')} "
                                      f"with sources {digest}); not measured inside this process")
            except Exception as e:
                traffic_source = f"profiles/traffic.json unreadable: {e}"
        kernel = kernel_label(level.info())
        out = {
            "metric": f"MLUPS (million lattice updates/s) at {args.size}^3 D3Q27; % of HBM roofline",
            "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(args, world, brick, rgrid, box, rehearsal),
                       "scaling_mode": args.scaling, "global_box_cells": list(box),
                       "cells_per_gpu": cells_per_rank, "global_cells": total_cells, "tau": 0.5006, "c_wale": 0.5,
                       "nu_sgs_background": 0.0005, "launch_order": args.order or "library default",
                       "device_preheat_ms": args.preheat_ms,
                       "parallelism": "single GPU" if world == 1 else f"spatial domain decomposition x{world}" + (" (1-GPU rehearsal over gloo)" if rehearsal else ""),
                       "halo_bytes_per_rank_per_step": None if runner is None else runner.ex.plan.bytes_per_step(),
                       "state_finite": ok},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": kernel, "kernel_ms": round(kern_ms, 4), "kernel_ms_median": round(kern_med, 4),
                         "kernel_ms_sampled_first_last": [round(per_launch[0], 4), round(per_launch[-1], 4)] if per_launch else None,
                         "kernel_ms_source": ("HIP events on the launch stream around the timed region / steps; median of individually "
                                              f"bracketed steps (every {every}th)") if runner is None else "wall clock per step (N > 1)",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_LUP * cells_per_rank,
                         "traffic_unit": "fabric-side bytes per launch (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE; Infinity-Cache hits included)",
                         "traffic_source": traffic_source, "source_digest": digest,
                         "rho_store": ("stored (boundary / interior part launches always store it)" if runner is not None else
                                       "elided (reproduced on demand, DESIGN 3.1)" if not os.environ.get("LUDWIG_EAGER_RHO") else "eager"),
                         "kernel_ms_eager_rho": None if eager_ms is None else round(eager_ms, 4),
                         "frac_eager_rho": None if eager_ms is None else round(ALGO_BYTES_PER_LUP * cells_per_rank / (eager_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "eager_rho_note": None if eager_ms is None else
                                           (f"{args.eager_rho_steps} further steps after the timed region with rho stored by every step, the reference "
                                            "kernel's store pattern (ludwig_level_set_rho_store); same kernel, same bits")},
        }
        if comm is not None:
            out["comm"] = comm
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    # teardown in an order somebody chose (round 2 left it to interpreter shutdown; profiles/README.md "exit-time SIGSEGV"): device idle,
    # torch's event / stream wrappers dropped, the process group (RCCL's streams and kernels) gone, and only then the level's memory
    # and the CU-masked stream underneath them
    del region, ev
    if runner is not None:
        runner.close(dist)
    else:
        torch.cuda.synchronize()
        level.close()
    if not ok:
        raise SystemExit("state went non-finite or trivial during the benchmark")


if __name__ == "__main__":
    main()
