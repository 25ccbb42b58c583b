"""Command-line counterpart of the reference's train.py (argument names follow train.py:162-259
where they apply): A2C (or PPO: `python -m ewn_gym_amd.train_a2c PPO ...`) on VecEWN lanes instead of SubprocVecEnv workers, per-epoch evaluation
against minimax on the un-shaped env (train.py:66-117), best-model checkpointing.

  python -m ewn_gym_amd.train_a2c --num_envs 4096 --epoch_num 10 --timesteps_per_epoch 200000
  python -m torch.distributed.run --nproc-per-node 8 -m ewn_gym_amd.train_a2c ...   (one process per GPU, RCCL)
"""
import argparse
import json
import os

import torch
import torch.distributed as dist

from .a2c import A2CTrainer, FusedA2CTrainer
from .sharding import all_reduce_counters, lane_range, lane_seeds
from .tournament import evaluate
from .vec_env import VecEWN


def main():
    ap = argparse.ArgumentParser(description="Trainer for EWN on VecEWN lanes (counterpart of the reference's train.py)")
    # train.py:174-184 selects the algorithm with a sub-command; A2C is the one built here
    ap.add_argument("algorithm", nargs="?", default="A2C", choices=["A2C", "PPO"], help="train.py:174-184's sub-command")
    ap.add_argument("--batch_size", "-b", type=int, default=None,
                    help="PPO: minibatch size in samples of the n_steps x lanes rollout buffer (default: a quarter of it; train.py:178-183)")
    ap.add_argument("--n_epochs", type=int, default=10, help="PPO: passes over the rollout buffer per update (SB3 default)")
    ap.add_argument("--trainer", default="auto", choices=["auto", "fused", "torch"],
                    help="A2C only.  fused: the whole loop in the engine (ewn_step_k_policy + ewn_a2c_grad / ewn_a2c_apply, five kernel launches "
                         "per update); torch: torch policy forward per step + ewn_step, autograd update; auto: fused where the engine has it")
    ap.add_argument("--checkpoint", default=None, help="path of a checkpoint written by this trainer to resume from (train.py:137-139, 248-251)")
    ap.add_argument("--model_seed", type=int, default=None, help="seed of the policy initialisation and sampling (default: --env_seed)")
    ap.add_argument("--num_envs", "-ne", type=int, default=4096, help="lanes per GPU")
    ap.add_argument("--n_steps", "-n", type=int, default=5)
    ap.add_argument("--learning_rate", "-lr", type=float, default=3e-4)
    ap.add_argument("--epoch_num", "-e", type=int, default=10)
    ap.add_argument("--timesteps_per_epoch", "-t", type=int, default=200000)
    ap.add_argument("--eval_episode_num", "-ee", type=int, default=256)
    ap.add_argument("--eval_max_depth", type=int, default=5)
    ap.add_argument("--board_size", type=int, default=5)
    ap.add_argument("--cube_layer", type=int, default=3)
    ap.add_argument("--opponent_policy", "-op", default="random")
    ap.add_argument("--max_depth", type=int, default=3)
    ap.add_argument("--goal_reward", type=float, default=10.0)
    ap.add_argument("--illegal_move_reward", type=float, default=-1.0)
    ap.add_argument("--illegal_move_tolerance", type=int, default=10)
    ap.add_argument("--reference_quirks", action="store_true",
                    help="reproduce MinimaxEnv's ctor-argument dropping: RandomAgent opponent, reward 1.0 (SURVEY App. D1)")
    ap.add_argument("--seed", "--env_seed", dest="seed", type=int, default=9487)
    ap.add_argument("--save_dir", default="models")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    lo, hi = lane_range(a.num_envs * world, world, rank)
    opp, reward = (("random", 1.0) if a.reference_quirks else (a.opponent_policy, a.goal_reward))
    env = VecEWN(a.num_envs, board_size=a.board_size, cube_layer=a.cube_layer, opponent_policy=opp, max_depth=a.max_depth,
                 rng="philox", shaped=True, reward=reward, illegal_move_reward=a.illegal_move_reward,
                 illegal_move_tolerance=a.illegal_move_tolerance, autoreset=True, lane_offset=lo,
                 seed_stride=a.num_envs * world, philox_key=a.seed, shaped_refresh_on_reset=not a.reference_quirks)
    env.reset(seeds=lane_seeds(lo, hi, a.seed).cuda())
    mseed = a.seed if a.model_seed is None else a.model_seed
    if a.algorithm == "PPO":   # train.py:39-49: SB3's PPO with batch_size and learning_rate given, the rest at its defaults (ewn_gym_amd/ppo.py)
        from .ppo import PPOTrainer
        trainer = PPOTrainer(env, n_steps=a.n_steps, batch_size=a.batch_size, n_epochs=a.n_epochs, learning_rate=a.learning_rate, seed=mseed)
    elif a.trainer == "fused" or (a.trainer == "auto" and env.supports_policy_rollout()):
        trainer = FusedA2CTrainer(env, n_steps=a.n_steps, learning_rate=a.learning_rate, seed=mseed)
    else:
        trainer = A2CTrainer(env, n_steps=a.n_steps, learning_rate=a.learning_rate, seed=mseed)
    if a.checkpoint is not None:      # train.py:137-139: resume the model (and here the optimiser and the step counter too)
        trainer.load(a.checkpoint)
        if rank == 0:
            print(json.dumps({"resumed_from": a.checkpoint, "timesteps": trainer.num_timesteps * world}), flush=True)
    if rank == 0:
        print(json.dumps({"trainer": type(trainer).__name__, "lanes_per_gpu": a.num_envs, "world": world}), flush=True)
    best = trainer.best_score   # -1 for a fresh run; a resumed one keeps its best model until it is beaten
    for epoch in range(a.epoch_num):
        stats = trainer.learn(a.timesteps_per_epoch // world)   # dict of the last update's statistics
        # train.py:73-81: evaluate on the UN-shaped env against minimax(depth 5), seeds 0..n-1, deterministic actions
        n_eval = a.eval_episode_num // world
        r = evaluate(trainer.policy_fn(True), {"kind": "minimax", "max_depth": a.eval_max_depth}, num=n_eval,
                     board_size=a.board_size, cube_layer=a.cube_layer, rng="mt19937", seed_offset=rank * n_eval)
        c = all_reduce_counters(torch.tensor([r["wins"], r["episodes"]], dtype=torch.int64, device="cuda"))
        win_rate = c[0].item() / max(1, c[1].item())
        if rank == 0:
            print(json.dumps({"epoch": epoch, "timesteps": trainer.num_timesteps * world, "win_rate": win_rate, **stats}), flush=True)
            if win_rate > best:          # train.py:109-112
                best = trainer.best_score = win_rate
                os.makedirs(a.save_dir, exist_ok=True)
                trainer.save(os.path.join(a.save_dir, "best.pt"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
