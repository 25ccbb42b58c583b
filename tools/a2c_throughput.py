"""Throughput of the on-device A2C loop (BASELINE config 4 on one GPU): shaped env, depth-3 minimax opponent."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ewn_gym_amd as ea
from ewn_gym_amd.a2c import A2CTrainer
for N in (4096, 65536):
    env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0,
                    illegal_move_tolerance=10, autoreset=True, shaped_refresh_on_reset=True, philox_key=1)
    env.reset(seeds=torch.arange(N, dtype=torch.int32))
    tr = A2CTrainer(env, n_steps=5, learning_rate=3e-4, seed=0)
    for _ in range(5):
        tr.collect_and_update()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = tr.num_timesteps
    for _ in range(40):
        st = tr.collect_and_update()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("A2C N=%d: %.3e env steps/s (%.2f ms per 5-step update), mean reward %.3f" % (N, (tr.num_timesteps - n0) / dt, dt / 40 * 1e3, tr.stats_dict(st)["mean_reward"]), flush=True)
