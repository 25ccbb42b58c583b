#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the UNMODIFIED reference.

Container-only: needs /root/reference (absent on the GPU box; the committed
fixtures travel instead).  The reference's third-party imports that are not
installed here (gymnasium, stable_baselines3, pygame) are satisfied by the
import stubs in oracle/ref_import_stubs/, which carry no game logic.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]

Fixtures are DATA: inputs and the outputs the reference produced for them
(floats as C99 hex strings so they round-trip bit-exactly).
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

    os.makedirs(args.out, exist_ok=True)

    def dump(name, obj):
        p = os.path.join(args.out, name)
        with open(p, "w") as f:
            json.dump(obj, f, separators=(",", ":"))
        print(name, os.path.getsize(p), "bytes")

    def B(env):
        return [int(v) for v in env.board.reshape(-1)]

    SIZES = [(5, 3), (6, 3), (7, 3), (7, 4), (7, 5), (9, 3)]

    # ---- G1: initial boards + first dice -------------------------------------------------
    g1 = []
    for S, L in SIZES:
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L)
        for seed in range(32):
            obs, _ = env.reset(seed=seed)
            g1.append({"S": S, "L": L, "seed": seed, "board": B(env), "dice": int(obs["dice_roll"])})
    dump("g1_initial.json", g1)

    # ---- self-play position generator over the reference's own primitives ---------------
    def selfplay_positions(S, L, n_games, rng, max_per_game=40):
        """Random legal self-play on a MinimaxEnv; yields (board, dice, player) at every ply."""
        out = []
        for g in range(n_games):
            env = envs.MinimaxEnv(board_size=S, cube_layer=L)
            env.reset(seed=int(rng.integers(0, 2**31)))
            player = Player.TOP_LEFT
            plies = 0
            while not env.check_win() and plies < max_per_game:
                dice = int(rng.integers(1, env.cube_num + 1))
                env.set_dice_roll(dice)
                out.append((env.board.copy(), dice, player))
                acts = env.get_legal_actions(player)
                a = acts[int(rng.integers(0, len(acts)))]
                env.make_simulated_action(player, a)
                player = Player.get_opponent(player)
                plies += 1
            out.append((env.board.copy(), int(rng.integers(1, env.cube_num + 1)), player))  # terminal position too
        return out

    gen = np.random.Generator(np.random.PCG64(20240607))

    # ---- G2: legal actions / cube selection / win test -----------------------------------
    g2 = []
    for (S, L), ng in (((5, 3), 60), ((6, 3), 10), ((7, 3), 12), ((7, 4), 12), ((7, 5), 8), ((9, 3), 4)):
        env = envs.MinimaxEnv(board_size=S, cube_layer=L)
        agent = cp.ExpectiMinimaxAgent(1, L, S)
        for board, dice, player in selfplay_positions(S, L, ng, gen):
            for pl in (Player.TOP_LEFT, Player.BOTTOM_RIGHT):
                agent.restore_env_with_obs({"board": board, "dice_roll": dice})
                e = agent.env
                win = bool(e.check_win())
                rec = {"S": S, "L": L, "board": [int(v) for v in board.reshape(-1)], "dice": dice,
                       "player": pl.value, "win": win}
                has = (board > 0).any() if pl == Player.TOP_LEFT else (board < 0).any()
                if has:
                    rec["legal"] = [[int(a[0]), int(a[1])] for a in e.get_legal_actions(pl)]
                    sel = []
                    for flag in (False, True):
                        idx = e.find_cube_to_move(flag, pl)
                        sel.append(int(idx + 1 if pl == Player.TOP_LEFT else -idx))
                    rec["cube_small"], rec["cube_large"] = sel
                g2.append(rec)
    dump("g2_legal.json", g2)

    # ---- G4: heuristics -------------------------------------------------------------------
    g4 = []
    for (S, L), ng in (((5, 3), 40), ((7, 3), 8), ((7, 5), 6), ((6, 3), 6)):
        agent = cp.ExpectiMinimaxAgent(1, L, S)
        for board, dice, player in selfplay_positions(S, L, ng, gen):
            agent.restore_env_with_obs({"board": board, "dice_roll": dice})
            rec = {"S": S, "L": L, "board": [int(v) for v in board.reshape(-1)]}
            for h in ("hybrid", "min_dist", "two_min_dist", "attk"):
                rec[h] = fhex(agent.env.evaluate(h))
            g4.append(rec)
    dump("g4_eval.json", g4)

    # ---- G5: ExpectiMinimaxAgent.predict --------------------------------------------------
    g5 = []

    def minimax_cases(S, L, positions, depths, heuristics):
        agents = {(d, h): cp.ExpectiMinimaxAgent(d, L, S, heuristic=h) for d in depths for h in heuristics}
        for board, dice, player in positions:
            if player != Player.TOP_LEFT:
                board = np.rot90(-board, 2).copy()
            if dice > 6:
                continue
            res = {}
            skip = False
            for (d, h), ag in agents.items():
                ag.restore_env_with_obs({"board": board, "dice_roll": dice})
                if ag.env.check_win():
                    skip = True
                    break
                val, act = ag.expectiminimax(d, ag.env.agent_player, None, -float("inf"), float("inf"))
                pact, _ = ag.predict({"board": board, "dice_roll": dice})
                assert list(pact) == list(act)
                res["%d/%s" % (d, h)] = [int(act[0]), int(act[1]), fhex(val)]
            if not skip:
                g5.append({"S": S, "L": L, "board": [int(v) for v in board.reshape(-1)], "dice": dice, "res": res})

    pos55 = selfplay_positions(5, 3, 40, gen)
    minimax_cases(5, 3, pos55[:260], (1, 2, 3), ("hybrid", "min_dist", "two_min_dist", "attk"))
    minimax_cases(5, 3, pos55[260:330], (4,), ("hybrid", "attk"))
    minimax_cases(5, 3, pos55[330:370], (5,), ("hybrid",))
    pos77 = selfplay_positions(7, 3, 6, gen)
    minimax_cases(7, 3, pos77[:60], (1, 2, 3), ("hybrid", "two_min_dist"))
    minimax_cases(7, 3, pos77[60:70], (4,), ("hybrid",))
    pos63 = selfplay_positions(6, 3, 3, gen)
    minimax_cases(6, 3, pos63[:30], (3,), ("hybrid",))
    dump("g5_minimax.json", g5)

    # ---- trajectories ----------------------------------------------------------------------
    def run_traj(env, seed, rule, max_steps=200):
        obs, _ = env.reset(seed=seed)
        rec = {"seed": seed, "board0": B(env), "dice0": int(obs["dice_roll"]), "steps": []}
        t = 0
        done = False
        while not done and t < max_steps:
            a = rule(env, t)
            obs, r, term, trunc, info = env.step(np.array(a))
            rec["steps"].append({"a": [int(a[0]), int(a[1])], "board": B(env), "dice": int(obs["dice_roll"]),
                                 "r": fhex(r), "term": bool(term), "trunc": bool(trunc),
                                 "msg": info.get("message")})
            done = term
            t += 1
        return rec

    def rule_kth(seed):
        def rule(env, t):
            acts = env.get_legal_actions(env.current_player)
            return acts[(seed + t) % len(acts)]
        return rule

    def rule_with_flags(seed):
        # legal direction, but an arbitrary larger/smaller flag (exercises the flag-ignored paths)
        def rule(env, t):
            acts = env.get_legal_actions(env.current_player)
            a = acts[(seed * 7 + t * 3) % len(acts)]
            return a
        return rule

    def rule_illegal_at(seed, when):
        def rule(env, t):
            if t == when:
                return [(seed + t) % 2, (seed + 2 * t) % 3]  # arbitrary, often illegal
            acts = env.get_legal_actions(env.current_player)
            return acts[(seed + t) % len(acts)]
        return rule

    def rule_raw(seed):
        # uniform over all 6 actions from a private generator ("untrained policy")
        g = np.random.Generator(np.random.PCG64(seed + 999))
        def rule(env, t):
            return [int(g.integers(0, 2)), int(g.integers(0, 3))]
        return rule

    # G3: RandomAgent opponent
    g3 = []
    for (S, L), nseed in (((5, 3), 96), ((7, 3), 16), ((7, 5), 8), ((6, 3), 8)):
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.random)
        for seed in range(nseed):
            for kind, rule in (("kth", rule_kth(seed)), ("illegal", rule_illegal_at(seed, seed % 5)), ("raw", rule_raw(seed))):
                if kind != "kth" and seed >= nseed // 2:
                    continue
                rec = run_traj(env, seed, rule)
                rec.update({"S": S, "L": L, "rule": kind, "opp": "random"})
                g3.append(rec)
    dump("g3_traj_random.json", g3)

    # G6: minimax opponents
    g6 = []
    for (S, L, depth, heur), nseed in (((5, 3, 3, "hybrid"), 48), ((5, 3, 1, "hybrid"), 8), ((5, 3, 2, "min_dist"), 8),
                                        ((5, 3, 4, "hybrid"), 4), ((5, 3, 5, "hybrid"), 4), ((7, 3, 3, "hybrid"), 6)):
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.minimax,
                                            max_depth=depth, heuristic=heur)
        for seed in range(nseed):
            rec = run_traj(env, seed, rule_kth(seed))
            rec.update({"S": S, "L": L, "rule": "kth", "opp": "minimax", "depth": depth, "heuristic": heur})
            g6.append(rec)
    dump("g6_traj_minimax.json", g6)

    # G7: MiniMaxHeuristicEnv (shaped reward + tolerance).  One env object, episodes run
    # back to back so the ctor-only prev_score / never-reset tolerance quirks (App. D1, D3) show.
    g7 = []
    for tol in (10, 3):
        env = envs.MiniMaxHeuristicEnv(board_size=5, cube_layer=3, illegal_move_tolerance=tol,
                                       opponent_policy=ClassicalPolicy.minimax, goal_reward=10., seed=123)
        g7.append({"tol": tol, "ctor_prev_score": fhex(env.prev_score), "ctor_reward": fhex(env.reward),
                   "opp_class": type(env.opponent_policy).__name__, "episodes": []})
        for seed in range(12):
            rule = rule_raw(seed) if seed % 2 else rule_kth(seed)
            rec = run_traj(env, seed, rule)
            rec["tol_after"] = int(env.illegal_move_tolerance)
            rec["prev_score_after"] = fhex(env.prev_score)
            g7[-1]["episodes"].append(rec)
    dump("g7_shaped.json", g7)

    # ---- G8: RNG stream ---------------------------------------------------------------------
    g8 = {"episodes": [], "raw": []}
    real_randint = np.random.randint
    log = []

    def logged(lo, hi=None, *a, **k):
        v = real_randint(lo, hi, *a, **k)
        log.append([int(lo), int(hi), int(v)])
        return v

    env = envs.EinsteinWuerfeltNichtEnv(board_size=5, cube_layer=3, opponent_policy=ClassicalPolicy.random)
    np.random.randint = logged
    try:
        for seed in list(range(40)) + [9487, 2**31 - 1, 2**32 - 1, 123456789]:
            del log[:]
            run_traj(env, seed, rule_kth(seed))
            g8["episodes"].append({"seed": seed, "calls": [list(c) for c in log]})
    finally:
        np.random.randint = real_randint
    for seed in (0, 1, 5489, 9487, 4294967295):
        np.random.seed(seed)
        seq = []
        for i in range(64):
            hi = [7, 3, 2, 11, 16, 4, 6, 5][i % 8]
            seq.append([0 if i % 3 else 1, hi + (0 if i % 3 else 1), 0])
            seq[-1][2] = int(np.random.randint(seq[-1][0], seq[-1][1]))
        rs = np.random.RandomState(seed)
        raw = [int(v) for v in rs.randint(0, 2**32, size=700, dtype=np.uint64)]  # 700 tempered outputs (crosses pos==624)
        g8["raw"].append({"seed": seed, "randint": seq, "u32": raw})
    dump("g8_rng.json", g8)

    # ---- G10: evaluation() of eval_minimax.py:16-50 with the deterministic minimax agent -------------------------
    g10 = []
    for (adepth, opp, odepth), nseed in (((3, ClassicalPolicy.random, None), 64), ((2, ClassicalPolicy.minimax, 3), 32),
                                         ((3, ClassicalPolicy.minimax, 3), 32), ((1, ClassicalPolicy.random, None), 64)):
        kw = {} if odepth is None else {"max_depth": odepth}
        env = envs.EinsteinWuerfeltNichtEnv(board_size=5, cube_layer=3, opponent_policy=opp, **kw)
        model = cp.ExpectiMinimaxAgent(adepth, 3, 5)
        scores, lengths = [], []
        for seed in range(nseed):
            obs, info = env.reset(seed=seed)
            done, n, reward = False, 0, 0
            while not done:
                action, _ = model.predict(obs, deterministic=True)
                obs, reward, done, _, info = env.step(action)
                n += 1
            scores.append(float(reward))
            lengths.append(n)
        g10.append({"agent_depth": adepth, "opp": str(opp), "opp_depth": odepth, "scores": scores, "lengths": lengths})
    dump("g10_eval_loop.json", g10)

    # ---- G11: make_simulated_action / undo (envs/ewn.py:377-434) and MinimaxEnv.simulate (minimax_ewn.py:215-238) ---
    g11 = {"moves": [], "simulate": []}
    env = envs.MinimaxEnv(board_size=5, cube_layer=3)
    for board, dice, player in selfplay_positions(5, 3, 12, gen):
        for pl in (Player.TOP_LEFT, Player.BOTTOM_RIGHT):
            if not ((board > 0).any() if pl == Player.TOP_LEFT else (board < 0).any()):
                continue
            for flag in (0, 1):
                for d in (0, 1, 2):
                    env.board[:] = board
                    env.dice_roll = dice
                    agent = cp.ExpectiMinimaxAgent(1, 3, 5)
                    agent.restore_env_with_obs({"board": board, "dice_roll": dice})
                    e = agent.env
                    e.history = []
                    e.make_simulated_action(pl, [flag, d])
                    after = [int(v) for v in e.board.reshape(-1)]
                    legal = e.history[-1] is not None
                    e.undo_simulated_action()
                    assert (e.board == board).all()
                    g11["moves"].append({"board": [int(v) for v in board.reshape(-1)], "dice": dice, "player": pl.value,
                                         "action": [flag, d], "after": after, "legal": legal})
    import random as _r
    _r.seed(777)
    for board, dice, player in selfplay_positions(5, 3, 2, gen)[::3][:8]:
        agent = cp.ExpectiMinimaxAgent(1, 3, 5)
        agent.restore_env_with_obs({"board": board, "dice_roll": dice})
        e = agent.env
        if e.check_win():
            continue
        tot = 0.0
        reps = 12
        for _ in range(reps):
            e.current_player = Player.TOP_LEFT
            tot += e.simulate()
        g11["simulate"].append({"board": [int(v) for v in board.reshape(-1)], "winrate": tot / reps, "n": reps * 100})
    dump("g11_simulated_action.json", g11)

    # ---- G9: flat Monte-Carlo statistics ---------------------------------------------------
    import copy
    import random as pyrandom
    pyrandom.seed(4242)
    g9 = []
    for (S, L), npos, nsim in (((5, 3), 14, 600), ((7, 3), 4, 200)):
        agent = cp.MctsAgent(L, S, num_simulations=nsim, num_env_copies=1)
        poss = [p for p in selfplay_positions(S, L, 6, gen) if p[2] == Player.TOP_LEFT and p[1] <= 6]
        step = max(1, len(poss) // npos)
        for board, dice, _ in poss[::step][:npos]:
            agent.restore_env_with_obs({"board": board, "dice_roll": dice})
            if agent.env.check_win():
                continue
            legal = agent.env.get_legal_actions(Player.TOP_LEFT)
            wins = []
            for a in legal:
                agent.env.make_simulated_action(Player.TOP_LEFT, a)
                env_copy = copy.deepcopy(agent.env)  # as tree_search_and_get_move does (mcts.py:77): simulate() rolls the copy's dice
                wins.append(int(agent.simulate(env_copy)))
                agent.env.undo_simulated_action()
            g9.append({"S": S, "L": L, "board": [int(v) for v in board.reshape(-1)], "dice": dice,
                       "legal": [[int(a[0]), int(a[1])] for a in legal], "wins": wins, "n": nsim})
    dump("g9_mcts.json", g9)
    return 0


if __name__ == "__main__":
    sys.exit(main())
