"""Batched agent-vs-opponent evaluation: the counterpart of the reference's
eval_{random,minimax,mcts}.py loops (eval_minimax.py:16-50) and of eval_pairs.py:10-35.

Every evaluation episode is one lane: episode k is `reset(seed=k)` exactly as upstream
(`for seed in range(num): env.reset(seed=seed)`), all lanes are stepped together until
each has terminated, the agent's `predict` is the same batched policy kernel the env
uses for opponents, and the score of an episode is its final reward (win <=> > 0).
With the MT19937-compat dice stream and a deterministic agent (minimax) the per-episode
results are bit-identical to the reference's; for random / MCTS agents they agree
statistically (different host RNG interleaving upstream).

The confidence interval is a Wilson interval on the win COUNT; the reference passes the
win *rate* as the count to proportion_confint (eval_minimax.py:102).
"""
import argparse
import json
import math

import torch

from .vec_env import VecEWN, predict_mcts, predict_minimax, predict_random


def wilson(wins, n, z=1.959963984540054):
    if n == 0:
        return (0.0, 1.0)
    p = wins / n
    den = 1 + z * z / n
    c = (p + z * z / (2 * n)) / den
    h = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / den
    return (max(0.0, c - h), min(1.0, c + h))


def _policy(spec, cube_layer, key):
    if callable(spec):          # e.g. A2CTrainer.policy_fn(): (board, dice, t) -> int8 [N, 2]
        return spec
    kind = spec["kind"]
    if kind == "random":
        return lambda b, d, t: predict_random(b, d, key=key, step=t, cube_layer=cube_layer)
    if kind == "minimax":
        return lambda b, d, t: predict_minimax(b, d, spec.get("max_depth", 3), spec.get("heuristic", "hybrid"),
                                               cube_layer=cube_layer)[0]
    if kind == "mcts":
        return lambda b, d, t: predict_mcts(b, d, spec.get("num_simulations", 10), spec.get("num_env_copies", 5),
                                            key=(key + 0x9E3779B97F4A7C15 * (t + 1)) & 0xFFFFFFFFFFFFFFFF,
                                            cube_layer=cube_layer)[0]
    raise ValueError("unknown agent kind %r" % kind)


def evaluate(agent, opponent, num=1024, board_size=5, cube_layer=3, rng="mt19937", seed_offset=0, key=12345, max_steps=400,
             use_rollout=True, chunk=16):
    """agent: a dict like the opponent's or a callable policy (board, dice, t) -> actions.
    agent / opponent: dicts {"kind": "random"|"minimax"|"mcts", max_depth=, heuristic=, num_simulations=, num_env_copies=}.
    Returns per-episode scores (float64 tensor), episode lengths and summary statistics.
    When both sides are engine policies the table-driven kernels cover (RandomAgent, 'hybrid' minimax; cube_layer 3) the whole
    predict/step loop runs on the device, `chunk` steps per launch (ewn_step_k); otherwise one policy kernel + one ewn_step per
    step.  Both give identical per-episode results for deterministic agents ("engine" in the result says which one ran)."""
    env = VecEWN(num, board_size=board_size, cube_layer=cube_layer, opponent_policy=opponent["kind"],
                 max_depth=opponent.get("max_depth", 3), heuristic=opponent.get("heuristic", "hybrid"),
                 num_simulations=opponent.get("num_simulations", 10), num_env_copies=opponent.get("num_env_copies", 5),
                 rng=rng, autoreset=False, philox_key=key ^ 0x5DEECE66D)
    env.reset(seeds=torch.arange(seed_offset, seed_offset + num, dtype=torch.int64).to(torch.int32))
    if (use_rollout and isinstance(agent, dict) and agent["kind"] in ("random", "minimax") and agent.get("heuristic", "hybrid") == "hybrid"
            and env.supports_rollout(agent["kind"], agent.get("max_depth", 3))):
        totals = env.alloc_totals()
        for _ in range(0, max_steps, chunk):
            env.rollout(chunk, agent=agent["kind"], agent_max_depth=agent.get("max_depth", 3), totals=totals)
            if bool((env.done != 0).all()):
                break
        env.check_rng()
        score, length = totals["return_sum"], totals["n_steps"]   # un-shaped env: the only non-zero reward of an episode is its last
        wins = int((score > 0).sum().item())
        lo, hi = wilson(wins, num)
        return {"scores": score, "lengths": length, "wins": wins, "episodes": num, "win_rate": wins / num, "engine": "ewn_step_k",
                "ci95": [lo, hi], "avg_score": float(score.mean().item()), "avg_length": float(length.float().mean().item())}
    policy = _policy(agent, cube_layer, key)
    score = torch.zeros(num, dtype=torch.float64, device=env.device)
    length = torch.zeros(num, dtype=torch.int32, device=env.device)
    for t in range(max_steps):
        alive = env.done == 0
        if not bool(alive.any()):
            break
        actions = policy(env.board, env.dice, t)
        _, _, reward, terminated, _, _ = env.step(actions)
        just = alive & (terminated != 0)
        score = torch.where(just, reward, score)
        length += alive.to(torch.int32)
    env.check_rng()
    wins = int((score > 0).sum().item())
    lo, hi = wilson(wins, num)
    return {"scores": score, "lengths": length, "wins": wins, "episodes": num, "win_rate": wins / num, "engine": "ewn_step",
            "ci95": [lo, hi], "avg_score": float(score.mean().item()), "avg_length": float(length.float().mean().item())}


def tournament(names=("random", "minimax", "mcts"), num=1024, max_depth=5, num_simulations=10, num_env_copies=5,
               board_size=5, cube_layer=3, heuristic="hybrid", rng="mt19937"):
    """eval_pairs.py:10-35: every (agent, opponent) pair, 1024 episodes, depth 5, 10 simulations by default."""
    def spec(n):
        return {"kind": n, "max_depth": max_depth, "heuristic": heuristic, "num_simulations": num_simulations,
                "num_env_copies": num_env_copies}
    table = {}
    for a in names:
        for o in names:
            r = evaluate(spec(a), spec(o), num=num, board_size=board_size, cube_layer=cube_layer, rng=rng)
            table["%s vs %s" % (a, o)] = {k: r[k] for k in ("wins", "episodes", "win_rate", "ci95", "avg_length")}
    return table


def main():
    ap = argparse.ArgumentParser(description="agent-vs-opponent win-rate matrix (counterpart of eval_pairs.py)")
    ap.add_argument("--agents", nargs="+", default=["random", "minimax", "mcts"])
    ap.add_argument("--num", type=int, default=1024)
    ap.add_argument("--max_depth", type=int, default=5)
    ap.add_argument("--heuristic", default="hybrid")
    ap.add_argument("--num_simulations", type=int, default=10)
    ap.add_argument("--num_env_copies", type=int, default=5)
    ap.add_argument("--board_size", type=int, default=5)
    ap.add_argument("--cube_layer", type=int, default=3)
    ap.add_argument("--rng", default="mt19937")
    a = ap.parse_args()
    t = tournament(a.agents, a.num, a.max_depth, a.num_simulations, a.num_env_copies, a.board_size, a.cube_layer,
                   a.heuristic, a.rng)
    for k, v in t.items():
        print("%-22s win rate %.3f  (95%% CI %.3f-%.3f, %d/%d, %.1f steps/episode)"
              % (k, v["win_rate"], v["ci95"][0], v["ci95"][1], v["wins"], v["episodes"], v["avg_length"]))
    print(json.dumps(t))


if __name__ == "__main__":
    main()
