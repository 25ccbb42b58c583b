#!/usr/bin/env python3
"""tests/golden/g9b_mcts_large.json: win counts of the UNMODIFIED reference's flat Monte-Carlo playouts
(MctsAgent.simulate, classical_policies/mcts.py:21-45) on many positions with many playouts per root move, so that the
HIP playout policy can be pinned to the reference tightly (an aggregate chi-square over all (position, move) cells,
tests/test_gpu_parity.py) instead of the loose per-cell 5-sigma check of g9_mcts.json.

Container-only (needs /root/reference; third-party imports satisfied by oracle/ref_import_stubs, which carry no game logic).
Positions (inputs) are reachable 5x5 and 7x7 positions produced by the CPU oracle's random self-play; every output number
comes from the reference.  One worker process per host core, each with its own `random.seed`.

Usage: python oracle/gen_golden_mcts.py [--positions 64] [--playouts 5000] [--procs 8]
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def positions(S, L, n, seed):
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import pyoracle as po
    orc = po.OracleVecEnv(4 * n, board_size=S, cube_layer=L, rng="philox", philox_key=seed, autoreset=True)
    orc.reset(seeds=np.arange(4 * n) + seed)
    gen = np.random.Generator(np.random.PCG64(seed))
    when = gen.integers(0, 16, 4 * n)
    out, dice = orc.obs()
    out = out.copy()
    for t in range(16):
        b = orc.step(orc.sample_legal_actions(t))[0]
        out[when == t] = b[when == t]
    dice = gen.integers(1, 7, 4 * n)
    keep = []
    for b, d in zip(out, dice):           # non-terminal positions only
        if b[S - 1, S - 1] > 0 or b[0, 0] < 0 or not (b > 0).any() or not (b < 0).any():
            continue
        keep.append((b.astype(int).tolist(), int(d)))
        if len(keep) == n:
            break
    return keep


def work(job):
    ref, S, L, items, nsim, seed = job
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, "ref_import_stubs"))
    sys.path.insert(0, ref)
    import copy
    import random as pyrandom
    import numpy as np
    import classical_policies as cp
    from constants import Player
    pyrandom.seed(seed)
    agent = cp.MctsAgent(L, S, num_simulations=nsim, num_env_copies=1)
    out = []
    for board, dice in items:
        agent.restore_env_with_obs({"board": np.array(board, dtype=np.int16), "dice_roll": dice})
        legal = agent.env.get_legal_actions(Player.TOP_LEFT)
        wins = []
        for a in legal:
            agent.env.make_simulated_action(Player.TOP_LEFT, a)
            env_copy = copy.deepcopy(agent.env)   # as tree_search_and_get_move does (mcts.py:77)
            wins.append(int(agent.simulate(env_copy)))
            agent.env.undo_simulated_action()
        out.append({"S": S, "L": L, "board": [v for row in board for v in row], "dice": dice,
                    "legal": [[int(a[0]), int(a[1])] for a in legal], "wins": wins, "n": nsim})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "g9b_mcts_large.json"))
    ap.add_argument("--positions", type=int, default=64)
    ap.add_argument("--playouts", type=int, default=5000)
    ap.add_argument("--positions7", type=int, default=8)
    ap.add_argument("--playouts7", type=int, default=1500)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    if not os.path.isdir(a.ref):
        print("reference not present; nothing to do")
        return 0
    import multiprocessing as mp
    jobs = []
    for (S, L, n, nsim, seed) in ((5, 3, a.positions, a.playouts, 101), (7, 3, a.positions7, a.playouts7, 202)):
        pos = positions(S, L, n, seed)
        for i in range(0, len(pos), 2):
            jobs.append((a.ref, S, L, pos[i:i + 2], nsim, 1000 * S + i))
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = pool.map(work, jobs, chunksize=1)
    out = [r for part in res for r in part]
    with open(a.out, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("%d positions, %.0f s, %d bytes" % (len(out), time.time() - t0, os.path.getsize(a.out)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
