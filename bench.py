#!/usr/bin/env python3
"""bench.py -- env steps/sec of the vectorised EWN step + depth-3 expectiminimax opponent.

One "step" = one pass of the hot path over one batch: a device-side stand-in agent
plays a uniformly random LEGAL action per lane (RandomAgent; by default emitted by the step
kernel itself for the new observation, or by a separate policy kernel with
--separate-agent-kernel), and ewn_step applies it, rolls the opponent's dice, runs the
opponent's search/reply, tests for the win, rolls the next dice and auto-resets finished
lanes -- for every lane.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--lanes 65536] [--opponent minimax] ...
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Inputs (boards, dice, RNG state) are resident in HBM
before the timed region; no host<->device traffic happens inside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALGO_BYTES_PER_LANE_STEP = {5: 64, 7: 112}  # read board+dice+action, write board+dice+reward+flags


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--lanes", type=int, default=65536, help="parallel games per GPU")
    ap.add_argument("--board-size", type=int, default=5)
    ap.add_argument("--cube-layer", type=int, default=3)
    ap.add_argument("--opponent", default="minimax", choices=["random", "minimax", "mcts"])
    ap.add_argument("--max-depth", type=int, default=3)
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--num-simulations", type=int, default=10, help="MctsAgent.num_simulations")
    ap.add_argument("--num-env-copies", type=int, default=5, help="MctsAgent.num_env_copies")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--separate-agent-kernel", action="store_true",
                    help="sample the agent's action with a separate policy kernel (ewn_predict_random) instead of ewn_step's fused "
                         "random_action output")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly from Python instead of replaying a hipGraph")
    ap.add_argument("--graph-steps", type=int, default=50, help="steps captured per graph")
    ap.add_argument("--mt-window", type=int, default=0, help="MT19937-compat mode: precomputed outputs per episode window (0 = engine default)")
    ap.add_argument("--pipeline-groups", type=int, default=2,
                    help="also report (extra field, not `value`) the rate of the same lanes split into this many groups that step on "
                         "their own streams without a barrier between them; 0 = skip")
    return ap.parse_args()


def _cpu_worker(job):
    """one host process: the CPU oracle on its own 2048 lanes for ~budget_s seconds"""
    (board_size, cube_layer, opponent, max_depth, rng, nsim, ncopies, budget_s, idx) = job
    from oracle import pyoracle as po
    import numpy as np
    n = 2048
    env = po.OracleVecEnv(n, board_size=board_size, cube_layer=cube_layer, opponent=opponent, max_depth=max_depth, rng=rng,
                          autoreset=True, philox_key=2024, seed_stride=n, lane_offset=idx * n, num_simulations=nsim,
                          num_env_copies=ncopies)
    env.reset(seeds=np.arange(n, dtype=np.uint32) + 9487 + idx * n)
    t0 = time.perf_counter()
    steps = 0
    while True:
        env.step(env.sample_legal_actions(steps))
        steps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    return n * steps, time.perf_counter() - t0


def cpu_baseline(args, budget_s):
    """The CPU oracle (oracle/ewn_oracle.c, a C restatement of the reference's algorithm) on the same workload shape,
    one process per host core (the reference's own parallelism is also process-level: SubprocVecEnv / Pool),
    bounded to ~budget_s seconds of wall time.  Also reports the single-thread rate."""
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))  # 16 = one GPU's host-core share on the pool's boxes
    job = (args.board_size, args.cube_layer, args.opponent, args.max_depth, args.rng, args.num_simulations, args.num_env_copies)
    from oracle import pyoracle as po
    po.build()
    s1, t1 = _cpu_worker(job + (min(3.0, budget_s / 3), 0))
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [job + (budget_s, i) for i in range(cores)])
    total = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return {"value": total / dt, "unit": "env steps/sec", "cores": cores, "kind": "port", "single_thread_value": s1 / t1,
            "sample": "%d processes x 2048 lanes, %.1f s wall (%d lane-steps) of the same workload on the CPU oracle" % (cores, dt, total)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The host-core baseline runs FIRST, before anything touches the GPU: it starts worker processes, and a process
    # that has initialised HIP must not be the one that forks/execs them.  Rank 0 at N=1 only.
    cpub = None
    if not args.no_cpu_baseline and world == 1:
        cpub = cpu_baseline(args, args.cpu_baseline_seconds)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    import ewn_gym_amd as ea
    from ewn_gym_amd.sharding import lane_range

    N = args.lanes
    lo, hi = lane_range(N * world, world, rank)  # weak scaling: every GPU owns N lanes, global ids [rank*N, (rank+1)*N)
    env = ea.VecEWN(N, board_size=args.board_size, cube_layer=args.cube_layer, opponent_policy=args.opponent,
                    max_depth=args.max_depth, rng=args.rng, autoreset=True, lane_offset=lo, seed_stride=N * world,
                    philox_key=2024, num_simulations=args.num_simulations, num_env_copies=args.num_env_copies,
                    want_random_action=not args.separate_agent_kernel, mt_window=args.mt_window)
    from ewn_gym_amd.sharding import lane_seeds
    env.reset(seeds=lane_seeds(lo, hi).cuda())  # reference default seed 9487 + global lane id
    counter = torch.zeros((), dtype=torch.int32, device="cuda")  # device-side step index: lets the captured graph advance
    fused_agent = not args.separate_agent_kernel
    if fused_agent:
        # The stand-in agent is RandomAgent: ewn_step itself emits RandomAgent.predict(new observation) into
        # ewn_step_out.random_action, and that buffer is the next step's `actions` (one kernel per step).
        actions = env.random_action
        env.sample_legal_actions(0, out=actions)      # the very first action
    else:
        actions = torch.zeros((N, 2), dtype=torch.int8, device="cuda")

    def one_step():
        if not fused_agent:
            env.sample_legal_actions(0, out=actions, step_tensor=counter)   # RandomAgent as a separate device policy kernel
        env.step(actions)                                                   # the hot path
        if not fused_agent:
            counter.add_(1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    # The per-step launch sequence is captured once in a hipGraph (same kernels, same order, same stream semantics) so the
    # timed region measures the GPU, not Python's ctypes launch overhead (~7 us/step on this host).
    graph = None
    if not args.no_graph:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(args.graph_steps):
                one_step()
    n_rep = args.steps // args.graph_steps if graph is not None else 0
    n_tail = args.steps - n_rep * (args.graph_steps if graph is not None else 0)

    barrier()
    t0 = time.perf_counter()
    for _ in range(n_rep):
        graph.replay()
    for _ in range(n_tail):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # Extra, clearly labelled: the same lanes as H independent groups, each stepping on its own stream inside one hipGraph.  With
    # the fused RandomAgent a group's step k+1 depends only on its own step k, so nothing forces a device-wide barrier per step.
    pipelined = None
    if args.pipeline_groups > 1 and fused_agent and graph is not None and N % args.pipeline_groups == 0:
        try:
            H, n = args.pipeline_groups, N // args.pipeline_groups
            genvs, gacts, gstreams = [], [], []
            for h in range(H):
                e = ea.VecEWN(n, board_size=args.board_size, cube_layer=args.cube_layer, opponent_policy=args.opponent,
                              max_depth=args.max_depth, rng=args.rng, autoreset=True, lane_offset=lo + h * n, seed_stride=N * world,
                              philox_key=2024, num_simulations=args.num_simulations, num_env_copies=args.num_env_copies,
                              want_random_action=True)
                e.reset(seeds=lane_seeds(lo + h * n, lo + (h + 1) * n).cuda())
                e.sample_legal_actions(0, out=e.random_action)
                genvs.append(e); gacts.append(e.random_action); gstreams.append(torch.cuda.Stream())
            for _ in range(min(20, args.warmup)):
                for e, a in zip(genvs, gacts):
                    e.step(a)
            torch.cuda.synchronize()
            pgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(pgraph):
                cur = torch.cuda.current_stream()
                for st_ in gstreams:
                    st_.wait_stream(cur)
                for e, a, st_ in zip(genvs, gacts, gstreams):
                    with torch.cuda.stream(st_):
                        for _ in range(args.graph_steps):
                            e.step(a)
                for st_ in gstreams:
                    cur.wait_stream(st_)
            torch.cuda.synchronize()
            reps = max(1, n_rep)
            tp0 = time.perf_counter()
            for _ in range(reps):
                pgraph.replay()
            torch.cuda.synchronize()
            pdt = time.perf_counter() - tp0
            pipelined = {"groups": H, "lanes_per_group": n, "value_this_rank": N * reps * args.graph_steps / pdt, "unit": "env steps/sec",
                         "us_per_step_of_all_lanes": pdt / (reps * args.graph_steps) * 1e6,
                         "note": "same lanes, %d groups on %d streams in one hipGraph, no barrier between the groups' steps; not the headline value" % (H, H)}
            del pgraph, genvs, gacts
        except Exception as exc:   # never let the extra measurement break the bench line
            pipelined = {"error": repr(exc)[:200]}

    # Dominant-kernel duration: the same K steps again, launched eagerly with a HIP event pair around every ewn_step on the
    # launch stream (a graph replay cannot be bracketed per kernel).  rocprofv3 --kernel-trace of this command must agree.
    kms = None
    if not args.no_kernel_timing:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for k in range(args.steps):
            if not fused_agent:
                env.sample_legal_actions(0, out=actions, step_tensor=counter)
            ev[k][0].record()
            env.step(actions)
            ev[k][1].record()
            if not fused_agent:
                counter.add_(1)
        torch.cuda.synchronize()
        kms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / dt
        bytes_per = ALGO_BYTES_PER_LANE_STEP.get(args.board_size, 2 * args.board_size ** 2 + 10)
        roof = None
        if kms is not None:
            achieved = N * bytes_per / (kms * 1e-3) / 1e9
            traffic = valu = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath) and N == 65536 and args.board_size == 5:
                try:   # PMC results of this kernel at this size, collected by tools/pmc_passes.sh in separate rocprofv3 passes
                    pmc = json.load(open(tpath))
                    key = "%s_d%d_%s" % (args.opponent, args.max_depth, args.rng)
                    traffic = pmc.get(key)
                    vi = pmc.get(key + "_valu_wave_insts")
                    if vi:   # the bound that actually binds: integer VALU issue, one wave-instruction per 4 cycles per SIMD
                        peak = 1024 * 2.4e9 / 4
                        valu = {"wave_insts_per_launch": vi, "achieved_per_s": vi / (kms * 1e-3), "peak_per_s": peak,
                                "frac": vi / (kms * 1e-3) / peak, "unit": "VALU wave-instructions/s (256 CUs x 4 SIMDs, 2.4 GHz, 4 cycles each)"}
                except Exception:
                    traffic = valu = None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "valu_issue": valu, "kernel": "k_step (fused agent move + opponent search + reply + auto-reset)",
                    "kernel_ms": kms, "algorithmic_bytes_per_launch": N * bytes_per,
                    "note": "integer/fp64-compare search work: VALU-bound, far from the HBM roof by construction (SURVEY 8d)"}
        line = {
            "metric": "env steps/sec (whole node), 5x5 EWN, depth-3 expectiminimax opponent" if
                      (args.board_size == 5 and args.opponent == "minimax" and args.max_depth == 3) else
                      "env steps/sec (whole node), %dx%d EWN, %s opponent" % (args.board_size, args.board_size, args.opponent),
            "value": value, "unit": "env steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 bitboards (int8 boards in HBM), f64 expectation", "data": "synthetic",
            "config": {"workload": ("%%dx%%d EWN, %%d parallel envs per GPU, %%s opponent%%s, random-legal agent (%s), auto-reset, %%s dice RNG"
                                    % ("emitted by the step kernel" if fused_agent else "separate policy kernel"))
                                   % (args.board_size, args.board_size, N, args.opponent,
                                      " depth %d (hybrid heuristic)" % args.max_depth if args.opponent == "minimax" else "", args.rng),
                       "lanes_per_gpu": N, "board_size": args.board_size, "cube_layer": args.cube_layer,
                       "opponent": args.opponent, "max_depth": args.max_depth, "rng": args.rng,
                       "launch": "hipGraph replay (%d steps per graph)" % args.graph_steps if graph is not None else "eager",
                       "parallelism": "lanes sharded across %d GPU(s), no data-path collective" % world},
            "leaf_positions_per_sec": value * 108 if (args.opponent == "minimax" and args.max_depth == 3) else None,
            "pipelined": pipelined,
            "roofline": roof, "cpu_baseline": cpub,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
