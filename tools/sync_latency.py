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

# host-side breakdown of the direct-call region: fresh events (torch creates the hipEvent at the first record) against re-used ones
for fresh in (True, False, False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not fresh:
        e0.record(); e1.record()
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    e0.record(); t.append(time.perf_counter())
    bound(); t.append(time.perf_counter())
    e1.record(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    print("fresh events %-5s: e0.record %.1f us, call %.1f, e1.record %.1f, synchronize %.1f, total %.1f; event time %.1f us"
          % (fresh, *[(b - a) * 1e6 for a, b in zip(t, t[1:])], (t[-1] - t[0]) * 1e6, e0.elapsed_time(e1) * 1e3))
# the same region with the wait done by polling the second event from the host
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(); bound(); e1.record()
    while not e1.query():
        pass
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("re-used events, host polls e1: %.1f us until the event is seen, %.1f with the synchronize after it; event time %.1f us"
          % ((t1 - t0) * 1e6, (time.perf_counter() - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
