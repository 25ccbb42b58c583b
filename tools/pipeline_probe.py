"""Probe: H independent groups of N/H lanes, each stepping on its own stream inside one hipGraph (no global sync between the
groups' steps), against one group of N lanes.  usage: python tools/pipeline_probe.py [N] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ewn_gym_amd as ea
from ewn_gym_amd.sharding import lane_seeds

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
G = 50
for H in (1, 2, 4):
    n = N // H
    envs, acts, streams = [], [], []
    for h in range(H):
        e = ea.VecEWN(n, opponent_policy="minimax", max_depth=3, rng="philox", autoreset=True, lane_offset=h * n, seed_stride=N,
                      philox_key=2024, want_random_action=True)
        e.reset(seeds=lane_seeds(h * n, (h + 1) * n).cuda())
        a = e.random_action
        e.sample_legal_actions(0, out=a)
        envs.append(e); acts.append(a); streams.append(torch.cuda.Stream())
    for _ in range(20):
        for e, a in zip(envs, acts):
            e.step(a)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for e, a, s in zip(envs, acts, streams):
            with torch.cuda.stream(s):
                for _ in range(G):
                    e.step(a)
        for s in streams:
            cur.wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS // G):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("N=%d H=%d: %.4g env steps/s, %.2f us per step of all %d lanes" % (N, H, N * (STEPS // G) * G / dt, dt / ((STEPS // G) * G) * 1e6, N), flush=True)
