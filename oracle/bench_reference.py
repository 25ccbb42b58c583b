#!/usr/bin/env python3
"""Time the UNMODIFIED reference's env.step() next to the C restatement, in THIS container.

Container-only: needs /root/reference (absent on the GPU box, where bench.py's `cpu_baseline` times the restatement
alone).  The ratio printed here (restatement / reference, same machine, same workload) is what bridges the two
machines: reference rate on the GPU box's host ~= cpu_baseline.value / ratio.

Workload = the bench's: 5x5, cube_layer 3, episodes `reset(seed=9487+k)`, uniform-random LEGAL agent, reset time
included.  TEST INFRASTRUCTURE, like everything under oracle/.

Usage: python oracle/bench_reference.py [--opponent minimax|random] [--max-depth 3] [--seconds 10] [--procs 1]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def _run_reference(args_tuple):
    ref, opponent, depth, seconds, worker = args_tuple
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, "ref_import_stubs"))
    sys.path.insert(0, ref)
    import numpy as np
    import envs
    from constants import ClassicalPolicy
    kw = {"opponent_policy": ClassicalPolicy.minimax, "max_depth": depth} if opponent == "minimax" else {}
    env = envs.EinsteinWuerfeltNichtEnv(board_size=5, cube_layer=3, **kw)
    rs = np.random.RandomState(1234 + worker)       # the agent's own stream: must not disturb the env's global one
    steps = episodes = 0
    t0 = time.perf_counter()
    k = worker * 1000003
    while time.perf_counter() - t0 < seconds:
        env.reset(seed=9487 + k)
        k += 1
        episodes += 1
        done = False
        while not done:
            legal = env.get_legal_actions(env.current_player)
            a = legal[rs.randint(0, len(legal))]
            _, _, term, trunc, _ = env.step(a)
            steps += 1
            done = term or trunc
    return steps, episodes, time.perf_counter() - t0


def _run_oracle(opponent, depth, seconds):
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle import pyoracle
    N = 4096
    env = pyoracle.OracleVecEnv(N, 5, 3, opponent=opponent, max_depth=depth, rng="philox", autoreset=True, philox_key=9487)
    env.reset(seeds=[9487 + i for i in range(N)])
    steps = 0
    t0 = time.perf_counter()
    s = 0
    while time.perf_counter() - t0 < seconds:
        a = env.sample_legal_actions(s)
        env.step(a)
        s += 1
        steps += N
    return steps, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--opponent", default="minimax", choices=["minimax", "random"])
    ap.add_argument("--max-depth", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--procs", type=int, default=1)
    args = ap.parse_args()
    if not os.path.isdir(args.ref):
        print("reference not present; nothing to do")
        return 0
    with mp.get_context("spawn").Pool(args.procs) as pool:
        res = pool.map(_run_reference, [(args.ref, args.opponent, args.max_depth, args.seconds, w) for w in range(args.procs)])
    ref_rate = sum(r[0] / r[2] for r in res)
    o_steps, o_t = _run_oracle(args.opponent, args.max_depth, min(args.seconds, 5.0))
    out = {"workload": "5x5, %s opponent%s, random-legal agent, reset included" % (args.opponent, " depth %d" % args.max_depth if args.opponent == "minimax" else ""),
           "reference_steps_per_s": ref_rate, "reference_procs": args.procs,
           "reference_steps_per_episode": sum(r[0] for r in res) / max(1, sum(r[1] for r in res)),
           "restatement_steps_per_s_1thread": o_steps / o_t,
           "restatement_over_reference_per_core": (o_steps / o_t) / (ref_rate / args.procs)}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
