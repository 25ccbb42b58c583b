import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ewn_gym_amd as ea
N = 65536
env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng=sys.argv[1] if len(sys.argv) > 1 else "philox", autoreset=True, philox_key=2024, want_terminal_obs=False)
env.reset(seeds=np.arange(N) + 9487)
acts = torch.zeros((N, 2), dtype=torch.int8, device="cuda")
for t in range(30):
    env.sample_legal_actions(t, out=acts)
    env.step(acts)
# diagnostic build writes its stamps over the (normally unused) terminal-board output: give it a private buffer
dbg = torch.zeros(N * 25 + 64, dtype=torch.int8, device="cuda")
env._out.terminal_board = dbg.data_ptr()
env.sample_legal_actions(31, out=acts)
env.step(acts)
torch.cuda.synchronize()
T = int(os.environ.get("EWN_D3_T", "4"))
nb = N * T // 256
mt = len(sys.argv) > 1 and sys.argv[1] == 'mt19937'
st = dbg[:(2 * nb if mt else nb) * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
if mt:
    st = st[nb:]   # the first half of the MT launch's blocks are refill blocks (one per step block)
order = [0, 1, 2, 7, 3, 4, 5, 6]
st = st[:, order]
d = np.diff(st, axis=1)
names = ["issue loads", "copy_in+barrier", "decode", "agent half", "search", "opp half+reset+encode+stores", "barrier+copy_out"]
print("T=%d blocks=%d  per-phase cycles (s_memtime ticks): median / p90" % (T, nb))
for i, n in enumerate(names):
    print("  %-32s %8.0f %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("  total %8.0f ; kernel span (max end - min start) %d ticks" % (np.median(st[:, 7] - st[:, 0]), st[:, 7].max() - st[:, 0].min()))
print("  block start spread: p50 %d p99 %d" % (np.percentile(st[:, 0] - st[:, 0].min(), 50), np.percentile(st[:, 0] - st[:, 0].min(), 99)))
