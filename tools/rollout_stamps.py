"""In-kernel phase timing of k_rollout_d3 (diagnostic build, never shipped).
  container:  python tools/rollout_stamps.py --build      -> ewn_gym_amd/lib/libewn_hip_stamps.so  (-DEWN_ROLLOUT_STAMPS)
  GPU box:    EWN_HIP_LIB=ewn_gym_amd/lib/libewn_hip_stamps.so python tools/rollout_stamps.py [--no-traj] [K]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "ewn_gym_amd", "lib", "libewn_hip_stamps.so")
if "--build" in sys.argv:
    src = os.path.join(ROOT, "ewn_gym_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DEWN_ROLLOUT_STAMPS",
                           "-o", LIB, os.path.join(src, "ewn_kernels.hip"), os.path.join(src, "ewn_step_d3.hip"), os.path.join(src, "ewn_rollout_s5.hip"),
                           os.path.join(src, "ewn_rollout_s6.hip"), os.path.join(src, "ewn_rollout_s7.hip"), os.path.join(src, "ewn_rollout_s8.hip")])
    print(LIB)
    sys.exit(0)
import numpy as np, torch, ewn_gym_amd as ea
assert "stamps" in ea.LIB_PATH, "run with EWN_HIP_LIB=%s" % LIB
args = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(args[0]) if args else 50
N = 65536
env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", autoreset=True, philox_key=2024, seed_stride=N)
env.reset(seeds=np.arange(N) + 9487)
traj = None if "--no-traj" in sys.argv else env.alloc_rollout(K, board="--no-board" not in sys.argv)
if traj is not None and "--only-board" in sys.argv:
    traj = {"board": traj["board"]}
tot = env.alloc_totals()
for _ in range(3):
    env.rollout(K, traj=traj, totals=tot)
torch.cuda.synchronize()
T = int(os.environ.get("EWN_ROLLOUT_T", "2"))
nw = N * T // 64
st = tot["return_sum"].view(torch.int64).cpu().numpy()[:nw * 8].reshape(nw, 8)
names = ["agent action + RNG block", "agent half", "opponent search", "opponent half + reset", "trajectory row"]
print("K=%d T=%d waves=%d   cycles per wave per STEP (s_memtime ticks): median / p90" % (K, T, nw))
for i, n in enumerate(names):
    print("  %-28s %8.0f %8.0f" % (n, np.median(st[:, i]) / K, np.percentile(st[:, i], 90) / K))
span = st[:, 7] - st[:, 6]
print("  loop total per step %8.0f ; launch span (max end - min begin) %d ticks" % (np.median(span) / K, st[:, 7].max() - st[:, 6].min()))
