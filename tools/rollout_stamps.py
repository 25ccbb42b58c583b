"""In-kernel phase timing of k_rollout_d3 (diagnostic build, never shipped).
  container:  python tools/rollout_stamps.py --build      -> ewn_gym_amd/lib/libewn_hip_stamps.so  (-DEWN_ROLLOUT_STAMPS)
  GPU box:    EWN_HIP_LIB=ewn_gym_amd/lib/libewn_hip_stamps.so python tools/rollout_stamps.py [--no-traj] [K]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "ewn_gym_amd", "lib", "libewn_hip_stamps.so")
if "--build" in sys.argv:
    # only the 5x5 rollout translation unit carries the stamps; the rest are the objects of the shipped build (python -m ewn_gym_amd.build)
    src, obj = os.path.join(ROOT, "ewn_gym_amd", "csrc"), os.path.join(ROOT, "ewn_gym_amd", "lib", "obj")
    so = os.path.join(obj, "ewn_rollout_s5_stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DEWN_ROLLOUT_STAMPS", "-c",
                           "-o", so, os.path.join(src, "ewn_rollout_s5.hip")])
    others = [os.path.join(obj, f) for f in sorted(os.listdir(obj)) if f.endswith(".o") and not f.startswith("ewn_rollout_s5")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, so] + others)
    print(LIB)
    sys.exit(0)
import numpy as np, torch, ewn_gym_amd as ea
assert "stamps" in ea.LIB_PATH, "run with EWN_HIP_LIB=%s" % LIB
args = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(args[0]) if args else 50
N = 65536
env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", autoreset=True, philox_key=2024, seed_stride=N)
env.reset(seeds=np.arange(N) + 9487)
slots = "--slots" in sys.argv    # the slot-task kernel (k_rollout_slots): N = 65 536 x 2 lanes per game
traj = None if "--no-traj" in sys.argv else (env.alloc_rollout(K, layout="record") if "--record" in sys.argv else env.alloc_rollout(K, board="--no-board" not in sys.argv))
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
if slots:
    names = ["step start: action, RNG block, agent half", "search: one cube's three roots", "opponent half + auto-reset", "trajectory row"]
    it = st[:, 4].astype(np.float64)
    print("K=%d T=%d waves=%d  iterations per wave: median %.1f  p90 %.1f  max %d" % (K, T, nw, np.median(it), np.percentile(it, 90), it.max()))
    print("cycles per wave per ITERATION (s_memtime ticks): median / p90")
    for i, n in enumerate(names):
        print("  %-44s %8.0f %8.0f" % (n, np.median(st[:, i] / it), np.percentile(st[:, i] / it, 90)))
print("K=%d T=%d waves=%d   cycles per wave per STEP (s_memtime ticks): median / p90" % (K, T, nw))
for i, n in enumerate(names):
    print("  %-28s %8.0f %8.0f" % (n, np.median(st[:, i]) / K, np.percentile(st[:, i], 90) / K))
span = st[:, 7] - st[:, 6]
print("  loop total per step %8.0f ; launch span (max end - min begin) %d ticks" % (np.median(span) / K, st[:, 7].max() - st[:, 6].min()))
