#!/usr/bin/env python3
"""Supplement to gen_golden.py: golden vectors on the LARGEST board this build supports (8x8, cube_layer 3) -> tests/golden/
g12_maxboard.json.  Same rules as gen_golden.py: container-only, runs the UNMODIFIED reference through the import stubs, writes
data only (inputs + the outputs the reference produced; floats as C99 hex).  Kept separate so that regenerating it does not
touch the other fixtures (G9's rollouts are unseeded upstream and would change).

Usage:  python oracle/gen_golden_maxboard.py [--ref /root/reference] [--out tests/golden]
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

    S, L = 8, 3
    gen = np.random.Generator(np.random.PCG64(8803))

    def B(env):
        return [int(v) for v in env.board.reshape(-1)]

    # positions from random legal self-play on the reference's own primitives
    positions = []
    for g in range(6):
        env = envs.MinimaxEnv(board_size=S, cube_layer=L)
        env.reset(seed=int(gen.integers(0, 2**31)))
        player = Player.TOP_LEFT
        plies = 0
        while not env.check_win() and plies < 60:
            dice = int(gen.integers(1, env.cube_num + 1))
            env.set_dice_roll(dice)
            positions.append((env.board.copy(), dice, player))
            acts = env.get_legal_actions(player)
            env.make_simulated_action(player, acts[int(gen.integers(0, len(acts)))])
            player = Player.get_opponent(player)
            plies += 1

    out = {"S": S, "L": L, "minimax": [], "eval": [], "traj": []}
    agents = {(d, h): cp.ExpectiMinimaxAgent(d, L, S, heuristic=h) for d, h in ((1, "hybrid"), (2, "hybrid"), (3, "hybrid"), (3, "attk"), (2, "two_min_dist"))}
    ev = cp.ExpectiMinimaxAgent(1, L, S)
    for board, dice, player in positions[::3]:
        if player != Player.TOP_LEFT:
            board = np.rot90(-board, 2).copy()
        ev.restore_env_with_obs({"board": board, "dice_roll": dice})
        out["eval"].append({"board": [int(v) for v in board.reshape(-1)],
                            **{h: fhex(ev.env.evaluate(h)) for h in ("hybrid", "min_dist", "two_min_dist", "attk")}})
        if ev.env.check_win():
            continue
        res = {}
        for (d, h), ag in agents.items():
            ag.restore_env_with_obs({"board": board, "dice_roll": dice})
            val, act = ag.expectiminimax(d, ag.env.agent_player, None, -float("inf"), float("inf"))
            res["%d/%s" % (d, h)] = [int(act[0]), int(act[1]), fhex(val)]
        out["minimax"].append({"board": [int(v) for v in board.reshape(-1)], "dice": dice, "res": res})

    def run_traj(env, seed):
        obs, _ = env.reset(seed=seed)
        rec = {"seed": seed, "board0": B(env), "dice0": int(obs["dice_roll"]), "steps": []}
        t, done = 0, False
        while not done and t < 200:
            acts = env.get_legal_actions(env.current_player)
            a = acts[(seed + t) % len(acts)] if (seed + t) % 11 else [int(t % 2), int((seed + t) % 3)]   # now and then a raw, possibly illegal action
            obs, r, term, trunc, info = env.step(np.array(a))
            rec["steps"].append({"a": [int(a[0]), int(a[1])], "board": B(env), "dice": int(obs["dice_roll"]), "r": fhex(r),
                                 "term": bool(term), "trunc": bool(trunc), "msg": info.get("message")})
            done = term
            t += 1
        return rec

    env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.random)
    for seed in range(10):
        rec = run_traj(env, seed)
        rec.update({"S": S, "L": L, "opp": "random"})
        out["traj"].append(rec)
    env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.minimax, max_depth=3, heuristic="hybrid")
    for seed in range(5):
        rec = run_traj(env, seed)
        rec.update({"S": S, "L": L, "opp": "minimax", "depth": 3, "heuristic": "hybrid"})
        out["traj"].append(rec)

    p = os.path.join(args.out, "g12_maxboard.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("g12_maxboard.json", os.path.getsize(p), "bytes;", len(out["minimax"]), "search positions,", len(out["eval"]), "evaluations,",
          len(out["traj"]), "trajectories,", sum(len(r["steps"]) for r in out["traj"]), "steps")
    return 0


if __name__ == "__main__":
    sys.exit(main())
