"""Throughput of the on-device A2C loop (BASELINE config 4 on one GPU): shaped env, depth-3 minimax opponent, n_steps 5.
   python tools/a2c_throughput.py [--trainer fused|torch|both] [--lanes 65536] [--updates 200]
fused: ewn_step_k_policy + ewn_a2c_grad + ewn_a2c_apply (five kernel launches per update, one hipGraph replay);
torch: the round-2 loop (torch policy forward per step + ewn_step, torch autograd update)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ewn_gym_amd as ea  # noqa: E402
from ewn_gym_amd.a2c import A2CTrainer, FusedA2CTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trainer", default="both", choices=["fused", "torch", "both"])
ap.add_argument("--lanes", type=int, nargs="*", default=[4096, 65536])
ap.add_argument("--updates", type=int, default=200)
ap.add_argument("--n-steps", type=int, default=5)
a = ap.parse_args()
for kind in (("fused", "torch") if a.trainer == "both" else (a.trainer,)):
    for N in a.lanes:
        env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0,
                        illegal_move_tolerance=10, autoreset=True, shaped_refresh_on_reset=True, philox_key=1)
        env.reset(seeds=torch.arange(N, dtype=torch.int32))
        cls = FusedA2CTrainer if kind == "fused" else A2CTrainer
        tr = cls(env, n_steps=a.n_steps, learning_rate=3e-4, seed=0)
        n_upd = a.updates if kind == "fused" else max(10, a.updates // 5)
        for _ in range(5):
            tr.collect_and_update()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n0 = tr.num_timesteps
        for _ in range(n_upd):
            st = tr.collect_and_update()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sd = tr.stats_dict() if kind == "fused" else tr.stats_dict(st)
        print("A2C %s N=%d: %.3e env steps/s (%.3f ms per %d-step update), mean reward %.3f" %
              (kind, N, (tr.num_timesteps - n0) / dt, dt / n_upd * 1e3, a.n_steps, sd["mean_reward"]), flush=True)
