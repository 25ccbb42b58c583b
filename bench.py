#!/usr/bin/env python3
"""bench.py -- env steps/sec of the vectorised EWN step + depth-3 expectiminimax opponent.

One "step" = one pass of the hot path over one batch: for every lane a device-side stand-in agent (RandomAgent:
a uniformly random LEGAL action) moves, the engine rolls the opponent's dice, runs the opponent's search / reply,
tests for the win, rolls the next dice and auto-resets finished lanes.

Launch modes (`config.launch` says which one was timed):
  rollout  (default)  ewn_step_k: K env steps per kernel launch, the game state stays in registers between the steps of a
                      launch and EVERY step's observation, action, reward and flags are written to a [K][N] trajectory
                      buffer in HBM (what a trainer's rollout buffer or an evaluation loop consumes);
  step                one ewn_step launch per env step (what `env.step()` of one vector env is), the agent's next action
                      emitted by the step kernel (ewn_step_out.random_action) -- round 1's headline, still reported in
                      the extra field `single_step_launch`.
Either way the timed region is hipGraph replays only, bracketed by barrier + synchronize and by one HIP event pair on the
launch stream (`roofline.kernel_ms` = event time / launches, so it can never exceed the wall-clock figure).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--lanes 65536] [--opponent minimax] ...
`--gpus N` (N > 1) outside torchrun starts the N ranks itself (torch.distributed.run, one process per GPU, RCCL).
Rank 0 prints ONE JSON line.  Inputs (boards, dice, RNG state) are resident in HBM before the timed region; no
host<->device traffic happens inside it.
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
    ap.add_argument("--heuristic", default="hybrid")
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--num-simulations", type=int, default=10, help="MctsAgent.num_simulations")
    ap.add_argument("--num-env-copies", type=int, default=5, help="MctsAgent.num_env_copies")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--mode", default="auto", choices=["auto", "rollout", "step"],
                    help="rollout: ewn_step_k, --steps-per-launch env steps per kernel launch; step: one ewn_step launch per env step; "
                         "auto: rollout where the engine has it for the configuration, else step")
    ap.add_argument("--steps-per-launch", type=int, default=50, help="rollout mode: env steps per ewn_step_k launch")
    ap.add_argument("--trajectory-layout", default="record", choices=["record", "columns"],
                    help="rollout mode: record = one 16-byte aligned record per lane-step (board | dice | action | flags; 32 B for 5x5) + the f64 "
                         "reward column; columns = packed [K][N][S*S] boards + one array per field (round 2's layout)")
    ap.add_argument("--no-trajectory", action="store_true",
                    help="rollout mode: do not write the per-step trajectory (only final state and per-lane counters)")
    ap.add_argument("--agent", default="legal", choices=["legal", "uniform6"],
                    help="stand-in agent: uniformly random LEGAL action (RandomAgent), or uniform over all 6 [flag, dir] actions "
                         "(what an untrained A2C policy plays: illegal-move terminations; step mode only)")
    ap.add_argument("--separate-agent-kernel", action="store_true",
                    help="step mode: sample the agent's action with a separate policy kernel (ewn_predict_random) instead of "
                         "ewn_step's fused random_action output")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly from Python instead of replaying a hipGraph")
    ap.add_argument("--graph-steps", type=int, default=50, help="env steps captured per graph (step mode)")
    ap.add_argument("--graph-rollout", action="store_true",
                    help="rollout mode: replay a hipGraph of the ewn_step_k launch(es) instead of calling the C ABI directly.  A K-step launch "
                         "is one long kernel: nothing to batch, and a graph launch costs the host more than the kernel launch it wraps "
                         "(tools/sync_latency.py: 203 us against 198 us wall for one 20-step launch); graphs are for the one-kernel-per-step mode")
    ap.add_argument("--mt-window", type=int, default=0, help="MT19937-compat mode: precomputed outputs per episode window (0 = engine default)")
    ap.add_argument("--no-spin", action="store_true", help="skip the clock warm-up on a scratch env before the timed region")
    ap.add_argument("--spin-ms", type=float, default=6.0,
                    help="minimum length of that clock warm-up; it goes on, in ~1 ms chunks that are each waited for, until the scratch env's step "
                         "time is stable (300 ms at most)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the N>1 code path (process group, barrier, MAX / ones / digest all-reduces) even at WORLD_SIZE=1: "
                         "exercises the RCCL branch on a one-GPU box")
    ap.add_argument("--collective-barrier", action="store_true",
                    help="N > 1: bracket the timed region with dist.barrier() (an RCCL all-reduce) instead of the node-local shared-memory barrier")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra measurements (single-step launch, no-trajectory rollout, uniform6 agent)")
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


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become the launcher.  Nothing has touched the GPU yet (torch is
    not even imported), so starting the N ranks as a child process and handing back its exit code is safe on this pool."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def source_hash():
    """sha256 over the kernel sources: the PMC summary under profiles/ carries it, so a stale counter file is detected"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ewn_gym_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_key(mode, opponent, max_depth, rng, board_size, lanes, steps_per_launch, trajectory, layout="record"):
    key = "%s_%s_d%d_%s_%dx%d_%d" % (mode, opponent, max_depth, rng, board_size, board_size, lanes)
    if mode == "rollout":
        key += "_k%d%s" % (steps_per_launch, "" if trajectory else "_notraj")
        if trajectory and layout != "record":
            key += "_" + layout
    return key


class ShmBarrier:
    """Barrier across the ranks of one node through a shared-memory segment: rank r writes the epoch into slot r and spins until every slot
    has reached it.  Aligned 8-byte stores; a slot has one writer.  Falls back to dist.barrier() (self.ok False) if the segment cannot be set up."""

    def __init__(self, world, rank, dist):
        import numpy as np
        from multiprocessing import shared_memory
        self.world, self.rank, self.dist, self.epoch, self.ok, self.shm = world, rank, dist, 0, False, None
        name = "ewn_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run"))
        name = "".join(c if c.isalnum() or c == "_" else "_" for c in name)[:60]
        try:
            if rank == 0:
                try:
                    stale = shared_memory.SharedMemory(name=name)
                    stale.close(); stale.unlink()
                except FileNotFoundError:
                    pass
                self.shm = shared_memory.SharedMemory(name=name, create=True, size=8 * world)
                np.ndarray((world,), dtype=np.int64, buffer=self.shm.buf)[:] = 0
            dist.barrier()                     # the segment exists and is zeroed
            if rank != 0:
                self.shm = shared_memory.SharedMemory(name=name)
                try:   # Python < 3.13 registers an ATTACHED segment with this process's resource tracker too, which then unlinks it a second time at exit
                    from multiprocessing import resource_tracker
                    resource_tracker.unregister(self.shm._name, "shared_memory")
                except Exception:
                    pass
            self.slots = np.ndarray((world,), dtype=np.int64, buffer=self.shm.buf)
            flag = 1
        except Exception as exc:               # every rank must take the same decision: agree on it below
            print("note: shared-memory barrier unavailable on rank %d (%r)" % (rank, exc), file=sys.stderr)
            flag = 0
        import torch
        t = torch.tensor([flag], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        self.ok = bool(int(t.item()))

    def wait(self):
        if not self.ok:
            self.dist.barrier()
            return
        self.epoch += 1
        e = self.epoch
        self.slots[self.rank] = e
        t0 = time.perf_counter()
        while int(self.slots.min()) < e:
            if time.perf_counter() - t0 > 120.0:
                raise RuntimeError("shared-memory barrier timed out (a rank died?)")

    def close(self):
        try:
            if self.shm is not None:
                self.slots = None
                self.shm.close()
                if self.rank == 0:
                    self.shm.unlink()
        except Exception:
            pass


class HipGraph:
    """A hipGraph captured, instantiated, UPLOADED and launched through the HIP runtime directly (ctypes on the libamdhip64 the
    process already has).  torch.cuda.CUDAGraph has no upload call, so the first replay of a fresh graph pays for it inside the
    timed region (~35 us measured, tools/sync_latency.py: noticeable when the region is one 20-step launch).  Only for launch
    sequences made of this engine's C-ABI calls (no torch operator, hence no allocator, is involved in the capture)."""

    def __init__(self, torch, fn):
        import ctypes as C
        import re
        libs = sorted(set(re.findall(r"(/\S*libamdhip64[^\s]*)", open("/proc/self/maps").read())), key=lambda p: ("torch" not in p, p))
        self.hip = C.CDLL(libs[0])
        self.torch, self.C = torch, C
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph, self.exe = C.c_void_p(), C.c_void_p()
        with torch.cuda.stream(side):
            self._ck(self.hip.hipStreamBeginCapture(C.c_void_p(side.cuda_stream), 2), "hipStreamBeginCapture")   # 2 = relaxed
            try:
                fn()
            finally:
                rc = self.hip.hipStreamEndCapture(C.c_void_p(side.cuda_stream), C.byref(graph))
            self._ck(rc, "hipStreamEndCapture")
        self._ck(self.hip.hipGraphInstantiate(C.byref(self.exe), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        self.hip.hipGraphDestroy(graph)
        cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._ck(self.hip.hipGraphUpload(self.exe, cur), "hipGraphUpload")
        torch.cuda.synchronize()

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (hipError %d)" % (what, rc))

    def replay(self):
        self._ck(self.hip.hipGraphLaunch(self.exe, self.C.c_void_p(self.torch.cuda.current_stream().cuda_stream)), "hipGraphLaunch")

    def __del__(self):
        try:
            if self.exe:
                self.hip.hipGraphExecDestroy(self.exe)
        except Exception:
            pass


class Runner:
    """One way of advancing `env` by env steps, captured into hipGraphs so that the timed region contains nothing but replays."""

    def __init__(self, torch, env, args, mode, trajectory=True, agent="legal"):
        self.torch, self.env, self.args, self.mode, self.agent = torch, env, args, mode, agent
        self.N = env.N
        self.counter = torch.zeros((), dtype=torch.int32, device="cuda")
        self.fused_agent = mode == "step" and agent == "legal" and not args.separate_agent_kernel
        self.kernels_per_step = 1.0
        self.trajectory = trajectory
        if mode == "rollout":
            self.K = max(1, min(args.steps_per_launch, args.steps))
            self.traj = env.alloc_rollout(self.K, layout=args.trajectory_layout) if trajectory else None
            self.kernels_per_step = 1.0 / self.K
        else:
            self.K = 1
            if self.fused_agent:
                # the stand-in agent is RandomAgent: ewn_step itself emits RandomAgent.predict(new observation) into
                # ewn_step_out.random_action, and that buffer is the next step's `actions` (one kernel per step)
                self.actions = env.random_action
                env.sample_legal_actions(0, out=self.actions)
            else:
                self.actions = torch.zeros((self.N, 2), dtype=torch.int8, device="cuda")
                self.kernels_per_step = 2.0
            if agent == "uniform6":
                self.gen = torch.Generator(device="cuda")
                self.gen.manual_seed(2024 + env.cfg.lane_offset)
                self.scale = torch.tensor([2.0, 3.0], device="cuda")
        self.graphs = {}
        self.bound_calls = {}
        self.graph_kind = "hipGraph (torch.cuda.CUDAGraph)"

    def launch(self, n_steps):
        """enqueue n_steps env steps"""
        env, torch = self.env, self.torch
        if self.mode == "rollout":
            done = 0
            while done < n_steps:
                k = min(self.K, n_steps - done)
                env.rollout(k, agent="sample" if self.agent == "uniform6" else "random", traj=self.traj)
                done += k
            return
        for _ in range(n_steps):
            if self.agent == "uniform6":
                # uniform over MultiDiscrete([2, 3]): what an untrained policy plays (illegal moves included)
                u = torch.rand((self.N, 2), device="cuda", generator=self.gen)
                self.actions.copy_((u * self.scale).to(torch.int8))
            elif not self.fused_agent:
                env.sample_legal_actions(0, out=self.actions, step_tensor=self.counter)   # RandomAgent as a separate policy kernel
            env.step(self.actions)                                                        # the hot path
            if not self.fused_agent and self.agent == "legal":
                self.counter.add_(1)

    def plan(self, steps):
        """[(steps per graph, replays)] covering exactly `steps` env steps"""
        g = max(1, min(self.args.graph_steps, steps))
        if self.mode == "rollout":
            g = min(steps, max(self.K, g // self.K * self.K))
        out = [(g, steps // g)]
        if steps % g:
            out.append((steps % g, 1))
        return out

    def graph(self, n_steps):
        torch = self.torch
        if n_steps not in self.graphs and self.mode == "rollout":
            torch.cuda.synchronize()
            try:      # captured, instantiated and uploaded ahead of the timed region
                self.graphs[n_steps] = HipGraph(torch, lambda: self.launch(n_steps))
                self.graph_kind = "hipGraph (captured and uploaded through the HIP runtime)"
            except Exception as exc:
                print("note: direct hipGraph path unavailable (%r); using torch.cuda.CUDAGraph" % (exc,), file=sys.stderr)
        if n_steps not in self.graphs:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if self.agent == "uniform6" and self.mode == "step":
                g.register_generator_state(self.gen)
            with torch.cuda.graph(g):
                self.launch(n_steps)
            self.graphs[n_steps] = g
        return self.graphs[n_steps]

    def bound(self, k):
        if k not in self.bound_calls:
            self.bound_calls[k] = self.env.bind_rollout(k, agent="sample" if self.agent == "uniform6" else "random", traj=self.traj)
        return self.bound_calls[k]

    def launches(self, steps):
        """kernel launches of the dominant kernel the plan issues"""
        if self.mode != "rollout":
            return steps
        n = 0
        for g, reps in self.plan(steps):
            n += reps * ((g + self.K - 1) // self.K)
        return n

    def run(self, steps, barrier, use_graph=True, spin=None):
        """time exactly `steps` env steps; returns (wall seconds, event milliseconds)"""
        torch = self.torch
        plan = self.plan(steps)
        direct = None
        if use_graph and self.mode == "rollout" and not self.args.graph_rollout:
            # K-step launches go straight through the C ABI, arguments marshalled beforehand (VecEWN.bind_rollout)
            use_graph = False
            direct = []
            for g, reps in plan:
                calls = [self.bound(min(self.K, g - d)) for d in range(0, g, self.K)]
                direct += calls * reps
            self.graph_kind = "direct C-ABI calls, arguments pre-marshalled (VecEWN.bind_rollout)"
        if use_graph:
            for g, _ in plan:
                self.graph(g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()   # torch creates the hipEvent at an event's first record(): not inside the timed region (measured: 13-18 us against 4.5)
        if spin is not None:
            spin()      # capture / instantiate / upload left the GPU idle for milliseconds: bring its clocks back up (untimed, other state)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        ta = time.perf_counter()
        if direct is not None:
            for call in direct:
                call()
        else:
            for g, reps in plan:
                for _ in range(reps):
                    if use_graph:
                        self.graphs[g].replay()
                    else:
                        self.launch(g)
        tb = time.perf_counter()
        e1.record()
        tc = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        self.host_us = {"e0.record": (ta - t0) * 1e6, "launches": (tb - ta) * 1e6, "e1.record": (tc - tb) * 1e6, "wait": (t0 + dt - tc) * 1e6}
        return dt, e0.elapsed_time(e1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
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
    dist_on = world > 1 or args.force_collectives
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    import ewn_gym_amd as ea
    from ewn_gym_amd.sharding import lane_range, lane_seeds

    N = args.lanes
    lo, hi = lane_range(N * world, world, rank)  # weak scaling: every GPU owns N lanes, global ids [rank*N, (rank+1)*N)

    def make_env():
        env = ea.VecEWN(N, board_size=args.board_size, cube_layer=args.cube_layer, opponent_policy=args.opponent,
                        max_depth=args.max_depth, heuristic=args.heuristic, rng=args.rng, autoreset=True, lane_offset=lo,
                        seed_stride=N * world, philox_key=2024, num_simulations=args.num_simulations,
                        num_env_copies=args.num_env_copies, want_random_action=True, mt_window=args.mt_window)
        env.reset(seeds=lane_seeds(lo, hi).cuda())  # reference default seed 9487 + global lane id
        return env

    # The barrier that brackets the timed region.  Across ranks of ONE node it is a host-side barrier in shared memory (every rank
    # publishes an epoch number in its own slot and waits until all slots carry it: ~2 us), taken after the rank's own
    # torch.cuda.synchronize().  A collective barrier (dist.barrier() on RCCL: a tiny all-reduce kernel plus its wait, measured +17 us
    # at world_size 1 and more across GPUs) would sit INSIDE a timed region that is 0.17 ms long in the driver's --steps 20 shape and
    # read as a scaling loss of a path that has no data-path collective at all.  --collective-barrier keeps dist.barrier(); so does
    # any failure to set the shared segment up, and a job that spans nodes.
    shm_bar = None
    if dist_on and world > 1 and not args.collective_barrier and int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world:
        shm_bar = ShmBarrier(world, rank, dist)

    def barrier():
        if dist_on and shm_bar is None:
            dist.barrier()
        torch.cuda.synchronize()
        if shm_bar is not None:
            shm_bar.wait()

    env = make_env()
    mode = args.mode
    if mode == "auto":
        # the flat Monte-Carlo opponent has a K-step kernel too (k_rollout_mcts), but its per-step block barriers cost more than the two
        # launches they save (65 536 lanes MCTS(10 x 5): 426 us per step against 408): auto keeps the three-launch step for it
        mode = "rollout" if (not args.separate_agent_kernel and args.opponent != "mcts"
                             and env.supports_rollout("sample" if args.agent == "uniform6" else "random")) else "step"
    runner = Runner(torch, env, args, mode, trajectory=not args.no_trajectory, agent=args.agent)
    runner.launch(args.warmup)                                   # W untimed warm-up steps
    # Clock warm-up.  Between the warm-up steps and the timed region the host captures, instantiates and uploads the graph(s): the
    # GPU sits idle for milliseconds and drops its clocks, and a timed region as short as the driver's (ONE 20-step launch, ~0.2 ms)
    # then runs partly at the low clock (measured: one run in four 1.4-2x slower).  So a SCRATCH env of the same configuration -- other
    # state, never the timed one -- is stepped for a few milliseconds right before the first barrier.  Not counted, not timed.
    spin = None
    if not args.no_spin:
        scratch = Runner(torch, make_env(), args, mode, trajectory=not args.no_trajectory, agent=args.agent)
        k0 = scratch.K if mode == "rollout" else 8
        scratch.launch(k0)                                       # loads the kernels, allocates what the first call allocates
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scratch.launch(k0)
        torch.cuda.synchronize()
        est = max(1e-7, (time.perf_counter() - t0) / k0)         # seconds per env step, roughly (this configuration, eager)
        chunk = max(k0, min(100000, int(1e-3 / est)))            # ~1 ms of scratch steps per chunk

        def spin():
            # chunks of scratch steps, each waited for (nothing stays queued in front of the timed launch), until the per-step time has
            # stopped improving for three chunks in a row -- the clocks are up -- and at least --spin-ms have passed; 300 ms at most.  A fixed
            # 6 ms was enough after a short idle phase, not after the 15 s of host-only work of the cpu_baseline leg (a default run then
            # measured its first 2 000 steps at a fifth of the rate of the runs behind it).
            t_begin, best, stable = time.perf_counter(), None, 0
            while True:
                t0 = time.perf_counter()
                scratch.launch(chunk)
                torch.cuda.synchronize()
                now = time.perf_counter()
                dt = (now - t0) / chunk
                stable = stable + 1 if (best is not None and dt > 0.97 * best) else 0
                best = dt if best is None else min(best, dt)
                if (stable >= 3 and now - t_begin >= args.spin_ms * 1e-3) or now - t_begin >= 0.3:
                    break
    dt, ev_ms = runner.run(args.steps, barrier, use_graph=not args.no_graph, spin=spin)   # exactly K timed steps
    if rank == 0:
        print("host side of the timed region (us): " + ", ".join("%s %.1f" % kv for kv in runner.host_us.items()) + "; events %.1f" % (ev_ms * 1e3), file=sys.stderr)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if (dist_on and args.backend == "nccl") else "cpu")
    ranks_seen = None
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        one = torch.ones(1, dtype=torch.int32, device=tmax.device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)               # every rank of the process group took part
        ranks_seen = int(one.item())
    dt = float(tmax.item())
    # a digest of the final state: the same lanes give the same digest however they are sharded (tests compare it)
    digest = int((env.board.to(torch.int64).reshape(N, -1) * torch.arange(1, args.board_size ** 2 + 1, device="cuda")).sum().item())
    if dist_on:
        dg = torch.tensor([digest], dtype=torch.int64, device=tmax.device)
        dist.all_reduce(dg, op=dist.ReduceOp.SUM)
        digest = int(dg.item())

    # ---- extras (N=1 only; clearly labelled, never `value`)
    extras = {}
    if world == 1 and not args.no_extras and not args.no_graph:
        def extra(name, mode_, **kw):
            try:
                r = Runner(torch, make_env(), args, mode_, **kw)
                steps = min(args.steps, 1000)
                r.launch(min(args.warmup, 50))
                # the faster of two runs: a run this short (0.2 ms in the driver's shape) is at the mercy of one host hiccup, and a
                # torch-captured graph pays its upload in its first replay (the main measurement above uploads its graph beforehand)
                d, evm = min(r.run(steps, barrier, spin=spin), r.run(steps, barrier, spin=spin))
                extras[name] = {"value": N * steps / d, "unit": "env steps/sec", "us_per_step": d / steps * 1e6,
                                "event_us_per_step": evm * 1e3 / steps, "steps": steps, "runs": 2}
            except Exception as exc:   # never let an extra measurement break the bench line
                extras[name] = {"error": repr(exc)[:200]}
        if mode == "rollout":
            extra("single_step_launch", "step")                  # round 1's headline: one ewn_step launch per env step
            if not args.no_trajectory:
                extra("rollout_without_trajectory", "rollout", trajectory=False)
        if args.agent == "legal":
            # illegal-move terminations.  In the engine (ewn_step_k's EWN_AGENT_SAMPLE = env.action_space.sample()) when the
            # configuration has a rollout kernel, and as a torch policy in front of one ewn_step per step (rand + cast + copy) either way
            if env.supports_rollout("sample"):
                extra("agent_uniform_over_all_6_actions", "rollout", agent="uniform6")
            extra("agent_uniform_over_all_6_actions_torch_policy", "step", agent="uniform6")

    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / dt
        bytes_per = ALGO_BYTES_PER_LANE_STEP.get(args.board_size, 2 * args.board_size ** 2 + 10)
        launches = runner.launches(args.steps)
        steps_per_launch = args.steps / launches
        kms = ev_ms / launches                                   # per ewn_step / ewn_step_k launch (launch gaps included)
        algo = N * bytes_per * steps_per_launch
        achieved = algo / (kms * 1e-3) / 1e9
        traffic = valu = rocprof_ms = None
        note = ""
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:   # PMC results of this kernel at this size, collected by tools/pmc_passes.sh in separate rocprofv3 passes
                pmc = json.load(open(tpath))
                ent = pmc.get(pmc_key(mode, args.opponent, args.max_depth, args.rng, args.board_size, N, runner.K, runner.trajectory, args.trajectory_layout))
                if ent and pmc.get("source_hash") == source_hash():
                    traffic = ent["hbm_bytes_per_launch"]
                    rocprof_ms = ent.get("rocprof_avg_us", 0.0) / 1e3 or None
                    vi, peak = ent.get("valu_wave_insts_per_launch"), ent.get("valu_issue_peak_per_s") or pmc.get("valu_issue_peak_per_s")   # the entry's own mix, if priced
                    if vi and peak:
                        valu = {"wave_insts_per_launch": vi, "achieved_per_s": vi / (kms * 1e-3), "peak_per_s": peak,
                                "frac": vi / (kms * 1e-3) / peak,
                                "unit": "VALU wave-instructions/s; peak = 1024 SIMDs x clock / the measured issue cycles of this kernel's "
                                        "instruction mix (tools/valu_probe.py, profiles/r02/valu_probe.json)"}
                elif ent:
                    note = "; profiles/pmc_traffic.json was collected on other kernel sources (hash mismatch): traffic not reported"
            except Exception:
                traffic = valu = None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "valu_issue": valu,
                "kernel": ("k_rollout_slots / k_rollout_d3 (K fused env steps per launch; the slot-task kernel from 131 072 lanes x lanes-per-game)" if mode == "rollout"
                           else ("k_mcts_rollout_lean (the playouts: ~97 % of the step; traffic / valu_issue are this kernel's) between the two half-step kernels" if args.opponent == "mcts"
                                 else "k_step (fused agent move + opponent search + reply + auto-reset)")),
                "kernel_ms": kms, "kernel_ms_rocprof": rocprof_ms, "env_steps_per_launch": steps_per_launch, "algorithmic_bytes_per_launch": algo,
                "note": "integer/fp64-compare search work: VALU-bound, far from the HBM roof by construction (SURVEY 8d); kernel_ms = one HIP "
                        "event pair on the launch stream around the timed region / launches; kernel_ms_rocprof = rocprofv3 --kernel-trace --stats average of the same "
                        "kernel at this launch shape (profiles/pmc_traffic.json, same kernel sources)" + note}
        if mode == "rollout":
            launch = "ewn_step_k: %d env steps per launch, %s; %s" % (
                runner.K, ("per-step trajectory (obs, action, reward, flags) written to HBM, %s" %
                           ("one aligned %d-byte record per lane-step + f64 reward" % ((args.board_size ** 2 + 6 + 15) & ~15) if args.trajectory_layout == "record"
                            else "packed columns")) if runner.trajectory else "no trajectory output",
                (runner.graph_kind if runner.graph_kind.startswith("direct") else runner.graph_kind + " replay") if not args.no_graph else "eager")
        else:
            launch = "one ewn_step launch per env step; " + ("hipGraph replay (%s)" % ", ".join("%d x %d steps" % (r, g) for g, r in runner.plan(args.steps))
                                                             if not args.no_graph else "eager")
        agent_txt = {"legal": "random-legal agent", "uniform6": "agent uniform over all 6 actions (env.action_space.sample())"}[args.agent]
        line = {
            "metric": "env steps/sec (whole node), 5x5 EWN, depth-3 expectiminimax opponent" if
                      (args.board_size == 5 and args.opponent == "minimax" and args.max_depth == 3) else
                      "env steps/sec (whole node), %dx%d EWN, %s opponent" % (args.board_size, args.board_size, args.opponent),
            "value": value, "unit": "env steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 bitboards (int8 boards in HBM), f64 expectation", "data": "synthetic",
            "config": {"workload": "%dx%d EWN, %d parallel envs per GPU, %s opponent%s, %s (in-engine), auto-reset, %s dice RNG"
                                   % (args.board_size, args.board_size, N, args.opponent,
                                      " depth %d (%s heuristic)" % (args.max_depth, args.heuristic) if args.opponent == "minimax" else "",
                                      agent_txt, args.rng),
                       "lanes_per_gpu": N, "board_size": args.board_size, "cube_layer": args.cube_layer,
                       "opponent": args.opponent, "max_depth": args.max_depth, "rng": args.rng, "mode": mode, "launch": launch,
                       "kernel_launches_in_timed_region": launches,
                       "clock_warmup": None if args.no_spin else "a scratch env of the same configuration stepped in ~1 ms chunks until its step time is stable (>= %g ms, <= 300 ms) right before the timed region (untimed, other state)" % args.spin_ms,
                       "timing_barrier": (None if not dist_on else ("host-side barrier in node-local shared memory after each rank's own synchronize" if (shm_bar is not None and shm_bar.ok)
                                                                    else "dist.barrier() (%s) + synchronize" % args.backend)),
                       "parallelism": "lanes sharded across %d GPU(s), no data-path collective" % world},
            "rccl_ranks": ranks_seen if args.backend == "nccl" else None,
            "collective": {"backend": args.backend if dist_on else None, "ranks": ranks_seen},
            "state_digest": digest,
            "extras": extras or None,
            "roofline": roof, "cpu_baseline": cpub,
        }
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        if shm_bar is not None:
            shm_bar.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
