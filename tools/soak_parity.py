"""One-off soak: lock-step HIP vs CPU oracle on many lanes and steps (all six outputs bit-exact), beyond what tests/ runs.
usage (GPU box): python tools/soak_parity.py [mcts | predict | r02 | d5 | r03 | r03b | long]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ewn_gym_amd as ea
from oracle import pyoracle as po

def bits(x): return np.asarray(x, np.float64).view(np.int64)
def run(N, steps, lo, hi, **kw):
    okw = dict(kw); opp = okw.pop("opponent_policy")
    env = ea.VecEWN(N, opponent_policy=opp, autoreset=True, seed_stride=N, **okw)
    seeds = (np.arange(N, dtype=np.uint64) * 11 + 17).astype(np.uint32)
    env.reset(seeds=seeds)
    orc = po.OracleVecEnv(hi - lo, opponent=opp, autoreset=True, seed_stride=N, lane_offset=lo, **okw)
    orc.reset(seeds=seeds[lo:hi])
    gen = np.random.Generator(np.random.PCG64(N))
    t0 = time.time(); nterm = 0
    for t in range(steps):
        a = env.sample_legal_actions(t).clone()
        if t % 4 == 3:   # raw actions, illegal ones included
            raw = torch.from_numpy(np.stack([gen.integers(0, 2, N), gen.integers(0, 3, N)], 1).astype(np.int8)).cuda()
            a.copy_(raw)
        oa = a[lo:hi].cpu().numpy()
        res = env.step(a)
        ores = orc.step(oa)
        for k in range(6):
            x, o = res[k][lo:hi].cpu().numpy(), ores[k]
            assert np.array_equal(bits(x) if k == 2 else x, bits(o) if k == 2 else o), (kw, t, k)
        nterm += int(ores[3].sum())
    print("ok N=%d steps=%d slice=%d episodes=%d %r (%.1f s)" % (N, steps, hi - lo, nterm, kw, time.time() - t0), flush=True)

if len(sys.argv) > 1:   # a named part (mcts / predict / r02)
    pass
else:
  run(40000, 120, 1000, 3000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=5)
  run(40000, 120, 35000, 37000, opponent_policy="minimax", max_depth=3, rng="mt19937")
  run(300000, 40, 299000, 300000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=6)
  run(9000, 80, 0, 3000, opponent_policy="minimax", max_depth=4, rng="philox", philox_key=7)
  run(5000, 60, 100, 1100, opponent_policy="minimax", max_depth=1, rng="mt19937")
  run(5000, 60, 100, 1100, opponent_policy="minimax", max_depth=2, rng="philox", philox_key=8)
  run(3000, 25, 0, 800, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=9)
  run(1500, 20, 0, 300, opponent_policy="minimax", max_depth=6, rng="mt19937")
  run(2000, 20, 0, 300, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=12, board_size=7)
  run(60000, 150, 20000, 24000, opponent_policy="random", rng="mt19937")
  run(60000, 150, 20000, 24000, opponent_policy="random", rng="philox", philox_key=10)
  run(20000, 60, 5000, 6000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=11, board_size=7)
  run(20000, 40, 5000, 5600, opponent_policy="minimax", max_depth=4, rng="mt19937", board_size=8)
  print("soak passed")

# flat Monte-Carlo: every win count must match the oracle (shared playout generator)
def positions(S, L, n, seed, max_steps=24):
    orc = po.OracleVecEnv(n, board_size=S, cube_layer=L, rng="philox", philox_key=seed, autoreset=True)
    orc.reset(seeds=np.arange(n) + seed)
    gen = np.random.Generator(np.random.PCG64(seed))
    when = gen.integers(0, max_steps, n)
    out, _ = orc.obs(); out = out.copy()
    for t in range(max_steps):
        b = orc.step(orc.sample_legal_actions(t))[0]
        out[when == t] = b[when == t]
    return out, gen.integers(1, 7, n).astype(np.int8)

def roll(N, K, launches, lo, hi, agent="random", agent_max_depth=3, autoreset=True, **kw):
    okw = dict(kw); opp = okw.pop("opponent_policy")
    env = ea.VecEWN(N, opponent_policy=opp, autoreset=autoreset, seed_stride=N, **okw)
    seeds = (np.arange(N, dtype=np.uint64) * 3 + 99).astype(np.uint32)
    env.reset(seeds=seeds)
    orc = po.OracleVecEnv(hi - lo, opponent=opp, autoreset=autoreset, seed_stride=N, lane_offset=lo, **okw)
    ob, od = orc.reset(seeds=seeds[lo:hi])
    traj = env.alloc_rollout(K)
    frozen = np.zeros(hi - lo, bool)
    t0 = time.time(); nterm = 0
    for launch in range(launches):
        env.rollout(K, agent=agent, agent_max_depth=agent_max_depth, traj=traj)
        tj = {k: v[:, lo:hi].cpu().numpy() for k, v in traj.items()}
        for k in range(K):
            acts = orc.random_actions() if agent == "random" else po.predict_minimax(ob, od, agent_max_depth, "hybrid")[0]
            live = ~frozen
            assert np.array_equal(tj["action"][k][live], acts[live]), (kw, launch, k)
            ob, od, r, te, tr, info = orc.step(np.where(live[:, None], acts, 0).astype(np.int8))
            for name, o in (("board", ob), ("dice", od), ("terminated", te), ("truncated", tr), ("info", info)):
                assert np.array_equal(tj[name][k], o), (kw, launch, k, name)
            assert np.array_equal(bits(tj["reward"][k]), bits(r)), (kw, launch, k)
            nterm += int((live & (te != 0)).sum())
            if not autoreset:
                frozen |= te != 0
    print("ok rollout N=%d K=%d launches=%d slice=%d episodes=%d agent=%s %r (%.1f s)" % (N, K, launches, hi - lo, nterm, agent, kw, time.time() - t0), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "mcts":
    for S, n, total in ((5, 300, 1000), (7, 120, 400), (8, 60, 333), (5, 2000, 13)):
        b, d = positions(S, 3, n, 77 + S)
        t0 = time.time()
        acts, wins = ea.predict_mcts(b, d, num_simulations=total, num_env_copies=1, key=S * 7 + total)
        oa, ow = po.predict_mcts(b, d, num_simulations=total, num_env_copies=1, key=S * 7 + total)
        assert np.array_equal(wins.cpu().numpy(), ow) and np.array_equal(acts.cpu().numpy(), oa), (S, n, total)
        w = po.playout_wins(b, 2, total, key=5)
        assert np.array_equal(ea.playout_wins(b, first_player=2, n_sims=total, key=5).cpu().numpy(), w)
        print("ok mcts S=%d positions=%d playouts/root=%d (%.1f s)" % (S, n, total, time.time() - t0), flush=True)
    run(3000, 30, 0, 3000, opponent_policy="mcts", num_simulations=4, num_env_copies=3, rng="philox", philox_key=3)
    print("mcts soak passed")

if len(sys.argv) > 1 and sys.argv[1] == "predict":
    # stateless ExpectiMinimaxAgent.predict, table-driven path against the oracle: action and root value bit patterns
    for S in (5, 6, 7, 8):
        for depth, n in ((1, 20000), (2, 20000), (3, 30000), (4, 20000), (5, 3000), (6, 600)):
            b, d = positions(S, 3, n, 4000 + 10 * S + depth, max_steps=14 if S == 5 else 26)
            t0 = time.time()
            acts, vals = ea.predict_minimax(b, d, depth, "hybrid")
            oa, ov, _ = po.predict_minimax(b, d, depth, "hybrid")
            assert np.array_equal(acts.cpu().numpy(), oa) and np.array_equal(bits(vals.cpu().numpy()), bits(ov)), (S, depth)
            print("ok predict S=%d depth=%d positions=%d (%.1f s)" % (S, depth, n, time.time() - t0), flush=True)
    print("predict soak passed")

if len(sys.argv) > 1 and sys.argv[1] == "d5":
    # the closed-form max_depth 5 / 6 search (ewn_search_d5.hpp): stateless predicts on every board size and (level, count) image,
    # then the step and rollout kernels at one and at two lanes per game
    for S in (5, 6, 7, 8):
        for heur in ("hybrid", "min_dist", "attk"):
            for depth, n in ((5, 40000 if heur == "hybrid" else 8000), (6, 6000 if heur == "hybrid" else 1500)):
                b, d = positions(S, 3, n, 9000 + 10 * S + depth, max_steps=14 if S == 5 else 26)
                t0 = time.time()
                acts, vals = ea.predict_minimax(b, d, depth, heur)
                oa, ov, _ = po.predict_minimax(b, d, depth, heur)
                assert np.array_equal(acts.cpu().numpy(), oa) and np.array_equal(bits(vals.cpu().numpy()), bits(ov)), (S, heur, depth)
                print("ok predict S=%d %s depth=%d positions=%d (%.1f s)" % (S, heur, depth, n, time.time() - t0), flush=True)
    run(65536, 40, 30000, 31500, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=41)          # two lanes per game
    run(140000, 30, 139000, 140000, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=42)       # one lane per game
    run(40000, 30, 0, 800, opponent_policy="minimax", max_depth=6, rng="mt19937")
    run(40000, 30, 39000, 40000, opponent_policy="minimax", max_depth=5, heuristic="min_dist", rng="philox", philox_key=43, board_size=7)
    run(33000, 30, 0, 800, opponent_policy="minimax", max_depth=6, heuristic="attk", rng="philox", philox_key=44, board_size=8)
    run(33000, 30, 0, 800, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=45, board_size=6)

    roll(65536, 12, 2, 20000, 21000, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=46)
    roll(140000, 8, 2, 0, 600, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=47)
    roll(2048, 10, 4, 0, 1024, agent="minimax", agent_max_depth=5, autoreset=False, opponent_policy="minimax", max_depth=5, rng="mt19937")
    roll(2048, 10, 4, 0, 512, agent="minimax", agent_max_depth=6, autoreset=False, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=48, board_size=7)
    print("d5 soak passed")

SHIFT = int(os.environ.get("SOAK_SHIFT", "0"))   # another set of games for a repeated `long` run
if len(sys.argv) > 1 and sys.argv[1] == "long":
    # the round's two new code paths at length: slot-task rollouts (games of a wave drift apart inside a launch) and the closed-form
    # max_depth 5 / 6 search, several launches deep, large slices against the oracle
    roll(262144, 60, 3, 100000, 104000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=61 + SHIFT)
    roll(65536, 100, 3, 60000, 63000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=62 + SHIFT)
    roll(65536, 37, 4, 0, 3000, opponent_policy="minimax", max_depth=4, heuristic="min_dist", rng="philox", philox_key=63 + SHIFT)
    roll(70000, 50, 2, 69000, 70000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=64 + SHIFT, board_size=8)
    roll(70000, 40, 2, 0, 1500, opponent_policy="minimax", max_depth=2, heuristic="attk", rng="philox", philox_key=65 + SHIFT, board_size=6)
    roll(65536, 40, 3, 1000, 2500, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=66 + SHIFT)
    roll(150000, 25, 2, 149000, 150000, opponent_policy="minimax", max_depth=6, rng="philox", philox_key=67 + SHIFT)
    roll(66000, 30, 2, 0, 1000, opponent_policy="minimax", max_depth=5, heuristic="attk", rng="philox", philox_key=68 + SHIFT, board_size=7)
    roll(140000, 40, 1, 0, 3000, autoreset=False, opponent_policy="minimax", max_depth=3, rng="mt19937")
    for S, n in ((5, 200000), (6, 100000), (7, 100000), (8, 100000)):
        b, d = positions(S, 3, n, 12000 + S + 100 * SHIFT, max_steps=14 if S == 5 else 26)
        t0 = time.time()
        acts, vals = ea.predict_minimax(b, d, 5, "hybrid")
        oa, ov, _ = po.predict_minimax(b, d, 5, "hybrid")
        assert np.array_equal(acts.cpu().numpy(), oa) and np.array_equal(bits(vals.cpu().numpy()), bits(ov)), S
        print("ok predict S=%d depth=5 positions=%d (%.1f s)" % (S, n, time.time() - t0), flush=True)
    print("long soak passed")

if len(sys.argv) > 1 and sys.argv[1] == "r02":
    # round 2 paths: shaped env and integer heuristics on the table-driven kernel, K-step rollouts (ewn_step_k)
    run(40000, 60, 1000, 3000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=21, shaped=True, reward=10.0,
        illegal_move_tolerance=3, shaped_refresh_on_reset=True)
    run(20000, 60, 0, 2000, opponent_policy="random", rng="mt19937", shaped=True, reward=10.0, illegal_move_tolerance=5)
    run(20000, 50, 4000, 6000, opponent_policy="minimax", max_depth=3, heuristic="min_dist", rng="philox", philox_key=22)
    run(20000, 50, 4000, 6000, opponent_policy="minimax", max_depth=4, heuristic="attk", rng="mt19937")
    run(6000, 30, 0, 1500, opponent_policy="minimax", max_depth=5, heuristic="attk", rng="philox", philox_key=23)
    run(10000, 40, 0, 2000, opponent_policy="minimax", max_depth=3, heuristic="min_dist", rng="philox", philox_key=24, board_size=7)

    roll(65536, 50, 4, 30000, 32000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=31)
    roll(262144, 25, 2, 200000, 201000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=32)
    roll(20000, 40, 3, 0, 2000, opponent_policy="random", rng="philox", philox_key=33)
    roll(20000, 30, 2, 3000, 4000, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=34, board_size=7)
    roll(4000, 20, 3, 0, 1000, opponent_policy="minimax", max_depth=4, heuristic="attk", rng="philox", philox_key=35, board_size=8)
    roll(2048, 12, 4, 0, 1024, agent="minimax", agent_max_depth=3, autoreset=False, opponent_policy="minimax", max_depth=3, rng="mt19937")
    roll(1024, 10, 4, 0, 512, agent="minimax", agent_max_depth=5, autoreset=False, opponent_policy="random", rng="mt19937")
    roll(512, 10, 3, 0, 256, agent="minimax", agent_max_depth=3, autoreset=False, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=36)
    print("r02 soak passed")

if len(sys.argv) > 1 and sys.argv[1] == "r03":
    # round-3 paths beyond the suite's sizes, through the suite's own lock-step helpers: record-layout rollouts at 262 144 / 1 M lanes,
    # 'two_min_dist' and flat-Monte-Carlo opponents inside ewn_step_k, the policy-driven rollout (logits / values vs torch fp32, every
    # transition vs the oracle, shaped env) over many launches, the MT19937-compat step at 65 536 lanes over many auto-resets (the
    # refill requests now go straight to their global list), and the fused A2C gradient at 65 536 lanes vs torch autograd
    from tests import test_gpu_rollout as tr, test_gpu_policy as tp, test_gpu_a2c_fused as tf
    t0 = time.time()
    for N, lo in ((262144, 200000), (1048576, 1040000)):
        n = tr._rollout_vs_oracle(ea, N, lo, lo + 1024, 25, 4, layout="record", opponent_policy="minimax", max_depth=3, rng="philox", philox_key=31)
        print("ok record rollouts N=%d: %d episodes in the slice (%.1f s)" % (N, n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 140000, 70000, 71024, 30, 3, layout="record", opponent_policy="minimax", max_depth=3, heuristic="two_min_dist", rng="philox", philox_key=32)
    print("ok two_min_dist rollouts: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 66000, 1000, 1768, 12, 3, opponent_policy="minimax", max_depth=4, heuristic="two_min_dist", rng="philox", philox_key=33, board_size=7)
    print("ok two_min_dist 7x7 depth 4: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 20000, 9000, 9512, 10, 3, opponent_policy="mcts", num_simulations=10, num_env_copies=5, rng="philox", philox_key=34, layout="record")
    print("ok MCTS(10 x 5) rollouts: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 3000, 0, 256, 6, 2, opponent_policy="mcts", num_simulations=40, num_env_copies=10, rng="philox", philox_key=35, board_size=7)
    print("ok MCTS 7x7 400 playouts rollouts: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    kw = dict(opponent_policy="minimax", max_depth=3, shaped=True, reward=10.0, illegal_move_reward=-1.0, illegal_move_tolerance=10,
              shaped_refresh_on_reset=True, philox_key=9487)
    n = tp._policy_vs_torch_and_oracle(ea, 65536, 30000, 32000, 5, 24, **kw)
    print("ok policy rollouts 65 536 lanes, 24 launches x 5 steps, 2 000-lane slice: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tp._policy_vs_torch_and_oracle(ea, 262144, 262144 - 1500, 262144, 8, 6, opponent_policy="random", philox_key=36)
    print("ok policy rollouts 262 144 lanes, RandomAgent opponent: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    run(65536, 150, 64000, 65536, opponent_policy="minimax", max_depth=3, rng="mt19937")
    run(65536, 150, 0, 1536, opponent_policy="random", rng="mt19937")
    tf.test_fused_gradient_matches_torch_autograd(ea, 65536, 5, 5, 0.0, "minimax")
    tf.test_fused_gradient_matches_torch_autograd(ea, 131072, 5, 7, 0.01, "minimax")
    print("ok fused A2C gradient vs torch autograd at 65 536 x 5 and 131 072 x 7 samples (%.1f s)" % (time.time() - t0), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "r03b":
    # second session of round 3: the compile-time trajectory instances at the bench's own shape (record + reward: TRJ = 1, both rollout
    # kernels; no trajectory at all: TRJ = 2, final state and totals against the oracle), BASELINE config 2 as records, and the generic
    # K-step kernel (geometries without a table image) beyond the suite's sizes
    from tests import test_gpu_rollout as tr
    t0 = time.time()
    n = tr._rollout_vs_oracle(ea, 65536, 63000, 65048, 50, 4, layout="record", opponent_policy="minimax", max_depth=3, rng="philox", philox_key=2024)
    print("ok 65 536 lanes, record layout, 4 launches x 50 steps, 2 048-lane slice: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 65536, 0, 4096, 50, 3, layout="record", opponent_policy="random", rng="philox", philox_key=2025)
    print("ok config 2 (RandomAgent opponent) as records, 3 x 50 steps, 4 096-lane slice: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 16384, 8000, 10048, 40, 3, layout="record", opponent_policy="minimax", max_depth=3, rng="philox", philox_key=2026)
    print("ok 16 384 lanes (lock-step kernel), record layout: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    for N, lo, kw in ((65536, 20000, dict(max_depth=3)), (140000, 139000, dict(max_depth=4, heuristic="attk")), (66000, 0, dict(max_depth=3, board_size=7))):
        S = kw.get("board_size", 5)
        env = ea.VecEWN(N, opponent_policy="minimax", rng="philox", philox_key=77, autoreset=True, seed_stride=N, **kw)
        seeds = (np.arange(N, dtype=np.uint64) * 3 + 99).astype(np.uint32)
        env.reset(seeds=seeds)
        hi = lo + 1000
        orc = po.OracleVecEnv(hi - lo, opponent="minimax", rng="philox", philox_key=77, autoreset=True, seed_stride=N, lane_offset=lo, **kw)
        ob, od = orc.reset(seeds=seeds[lo:hi])
        tot = env.alloc_totals()
        ret = np.zeros(hi - lo); nep = np.zeros(hi - lo, np.int64); nwin = np.zeros(hi - lo, np.int64)
        for launch in range(3):
            env.rollout(35, totals=tot)                      # no trajectory: the TRJ = 2 instance
            for k in range(35):
                ob, od, r, te, trn, info = orc.step(orc.random_actions())
                ret += r; nep += te != 0; nwin += info == 2
            assert np.array_equal(env.board[lo:hi].cpu().numpy(), ob) and np.array_equal(env.dice[lo:hi].cpu().numpy(), od), (N, launch)
        assert np.array_equal(bits(tot["return_sum"][lo:hi].cpu().numpy()), bits(ret))
        assert np.array_equal(tot["n_episodes"][lo:hi].cpu().numpy(), nep) and np.array_equal(tot["n_wins"][lo:hi].cpu().numpy(), nwin)
        print("ok no-trajectory rollouts N=%d %s: %d episodes in the slice (%.1f s)" % (N, kw, int(nep.sum()), time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 40000, 30000, 30512, 12, 3, layout="record", board_size=7, cube_layer=4, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=5)
    print("ok generic K-step kernel 7x7 cube_layer 4 depth 3, 40 000 lanes: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 70000, 69000, 70000, 20, 3, board_size=10, opponent_policy="random", rng="philox", philox_key=6)
    print("ok generic K-step kernel 10x10 RandomAgent opponent, 70 000 lanes: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    n = tr._rollout_vs_oracle(ea, 20000, 0, 512, 10, 2, board_size=8, cube_layer=5, opponent_policy="minimax", max_depth=2, heuristic="min_dist", rng="philox", philox_key=7)
    print("ok generic K-step kernel 8x8 cube_layer 5 depth 2 min_dist: %d episodes (%.1f s)" % (n, time.time() - t0), flush=True)
    print("r03b soak passed")
