"""Wall time of the reference's headline evaluation (assets/p16.png: 1 024 episodes, minimax agent depth 1..5 vs RandomAgent)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ewn_gym_amd.tournament import evaluate
for depth in (1, 2, 3, 4, 5):
    evaluate({"kind": "minimax", "max_depth": depth}, {"kind": "random"}, num=64)  # warm-up / compile caches
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = evaluate({"kind": "minimax", "max_depth": depth}, {"kind": "random"}, num=1024)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("minimax depth %d vs random: 1024 episodes in %.3f s, win rate %.3f" % (depth, dt, r["win_rate"]), flush=True)
t0 = time.perf_counter()
r = evaluate({"kind": "minimax", "max_depth": 5}, {"kind": "minimax", "max_depth": 5}, num=1024)
torch.cuda.synchronize()
print("minimax(5) vs minimax(5): 1024 episodes in %.3f s, win rate %.3f" % (time.perf_counter() - t0, r["win_rate"]))
t0 = time.perf_counter()
r = evaluate({"kind": "mcts"}, {"kind": "minimax", "max_depth": 5}, num=1024, rng="philox")
torch.cuda.synchronize()
print("mcts(50) vs minimax(5): 1024 episodes in %.3f s, win rate %.3f" % (time.perf_counter() - t0, r["win_rate"]))
