#!/usr/bin/env python3
"""Supplement to gen_golden.py: cube_layer 4 and 5 (10 / 15 cubes a side; the search's dice loop still runs 1..6 upstream,
SURVEY App. D) -> tests/golden/g13_layers.json: search results and trajectories against a minimax opponent, produced by the
UNMODIFIED reference through the import stubs.  Container-only; data only (floats as C99 hex).

Usage:  python oracle/gen_golden_layers.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def fhex(x):
    return float(x).hex()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    args = ap.parse_args()
    if not os.path.isdir(args.ref):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, "ref_import_stubs"))
    sys.path.insert(0, args.ref)
    import numpy as np
    import envs
    import classical_policies as cp
    from constants import Player, ClassicalPolicy

    gen = np.random.Generator(np.random.PCG64(4513))
    out = []
    for S, L, ngames, keys, ntraj in ((7, 4, 4, ((1, "hybrid"), (2, "hybrid"), (3, "hybrid"), (2, "attk")), 4),
                                      (8, 5, 3, ((1, "hybrid"), (2, "hybrid"), (2, "min_dist")), 3),
                                      (6, 4, 3, ((2, "two_min_dist"), (3, "hybrid")), 3)):
        grp = {"S": S, "L": L, "minimax": [], "traj": []}
        positions = []
        for g in range(ngames):
            env = envs.MinimaxEnv(board_size=S, cube_layer=L)
            env.reset(seed=int(gen.integers(0, 2**31)))
            player = Player.TOP_LEFT
            plies = 0
            while not env.check_win() and plies < 70:
                dice = int(gen.integers(1, 7))   # an observation's dice may be any cube number, but predict() only handles what the search's 1..6 loop does
                env.set_dice_roll(dice)
                positions.append((env.board.copy(), dice, player))
                acts = env.get_legal_actions(player)
                env.make_simulated_action(player, acts[int(gen.integers(0, len(acts)))])
                player = Player.get_opponent(player)
                plies += 1
        agents = {(d, h): cp.ExpectiMinimaxAgent(d, L, S, heuristic=h) for d, h in keys}
        for board, dice, player in positions[::4]:
            if player != Player.TOP_LEFT:
                board = np.rot90(-board, 2).copy()
            res = {}
            skip = False
            for (d, h), ag in agents.items():
                ag.restore_env_with_obs({"board": board, "dice_roll": dice})
                if ag.env.check_win():
                    skip = True
                    break
                val, act = ag.expectiminimax(d, ag.env.agent_player, None, -float("inf"), float("inf"))
                res["%d/%s" % (d, h)] = [int(act[0]), int(act[1]), fhex(val)]
            if not skip:
                grp["minimax"].append({"board": [int(v) for v in board.reshape(-1)], "dice": dice, "res": res})
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.minimax, max_depth=2, heuristic="hybrid")
        for seed in range(ntraj):
            obs, _ = env.reset(seed=seed)
            rec = {"seed": seed, "S": S, "L": L, "opp": "minimax", "depth": 2, "heuristic": "hybrid",
                   "board0": [int(v) for v in env.board.reshape(-1)], "dice0": int(obs["dice_roll"]), "steps": []}
            t, done = 0, False
            while not done and t < 200:
                acts = env.get_legal_actions(env.current_player)
                a = acts[(seed + t) % len(acts)]
                obs, r, term, trunc, info = env.step(np.array(a))
                rec["steps"].append({"a": [int(a[0]), int(a[1])], "board": [int(v) for v in env.board.reshape(-1)], "dice": int(obs["dice_roll"]),
                                     "r": fhex(r), "term": bool(term), "trunc": bool(trunc), "msg": info.get("message")})
                done = term
                t += 1
            grp["traj"].append(rec)
        out.append(grp)
        print((S, L), len(grp["minimax"]), "search positions,", len(grp["traj"]), "trajectories,", sum(len(r["steps"]) for r in grp["traj"]), "steps", flush=True)
    p = os.path.join(args.out, "g13_layers.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("g13_layers.json", os.path.getsize(p), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
