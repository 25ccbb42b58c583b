"""How much of a short timed region is host-side: graph launch + the wake-up latency of torch.cuda.synchronize() against polling
an event (GPU box).  The region is ONE 20-step ewn_step_k launch, as in `bench.py --steps 20`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ewn_gym_amd as ea
N, K = 65536, 20
env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", autoreset=True, philox_key=2024, seed_stride=N)
env.reset(seeds=np.arange(N) + 9487)
traj = env.alloc_rollout(K)
env.rollout(5, traj=traj)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    env.rollout(K, traj=traj)
bound = env.bind_rollout(K, traj=traj)      # the same launch as a direct C-ABI call with pre-marshalled arguments
bound(); torch.cuda.synchronize()
res = {"sync": [], "poll": [], "direct": [], "direct_noevents": [], "eager": [], "event_ms": [], "event_ms_direct": []}
for rep in range(12):
    for mode in ("sync", "poll", "direct", "direct_noevents", "eager"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "direct_noevents":
            bound()
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) * 1e6)
            continue
        e0.record()
        if mode == "direct":
            bound()
        elif mode == "eager":
            env.rollout(K, traj=traj)
        else:
            g.replay()
        e1.record()
        if mode == "poll":
            while not e1.query():
                pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[mode].append(dt * 1e6)
        res["event_ms_direct" if mode in ("direct", "eager") else "event_ms"].append(e0.elapsed_time(e1) * 1e3)
for k, v in res.items():
    v = sorted(v)
    print("%-9s median %.1f us  min %.1f  max %.1f" % (k, v[len(v) // 2], v[0], v[-1]))
