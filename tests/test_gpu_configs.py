"""BASELINE.json configs 4 and 5 at their own shapes (per GPU), through the C ABI:

  config 5: 7x7 EWN, MCTS(400 playouts per root move) opponent, 32 768 lanes per GPU (262 144 over 8)
            -- the split-phase step (k_step<.,1> -> k_mcts_init -> rollout -> k_mcts_pick -> k_step<.,2>) on a 49-cell board;
  config 4: A2C on the shaped training env (envs/training_ewn.py:43-99) with a depth-3 minimax opponent, 65 536 lanes.

A slice of lanes is followed in lock step by the CPU oracle (which mirrors the playout generator, so MCTS is bit-exact);
the rest of the lanes are covered by size-independent properties.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import pyoracle as po  # noqa: E402


@pytest.fixture(scope="module")
def ea():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ewn_gym_amd
    return ewn_gym_amd


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


def cpu(t):
    return t.detach().cpu().numpy()


def _slice_lockstep(ea, N, lo, hi, steps, raw_every=0, **kw):
    """N lanes on the GPU, lanes [lo, hi) followed by the oracle (same seeds, global lane ids, actions)."""
    okw = dict(kw)
    opp = okw.pop("opponent_policy")
    env = ea.VecEWN(N, opponent_policy=opp, autoreset=True, seed_stride=N, **okw)
    seeds = (np.arange(N, dtype=np.uint64) * 13 + 9487).astype(np.uint32)
    env.reset(seeds=seeds)
    orc = po.OracleVecEnv(hi - lo, opponent=opp, autoreset=True, seed_stride=N, lane_offset=lo, **okw)
    orc.reset(seeds=seeds[lo:hi])
    gen = np.random.Generator(np.random.PCG64(N + lo))
    S = kw.get("board_size", 5)
    cn = kw.get("cube_layer", 3) * (kw.get("cube_layer", 3) + 1) // 2
    nterm = 0
    for t in range(steps):
        a = env.sample_legal_actions(t).clone()
        if raw_every and t % raw_every == raw_every - 1:   # raw actions, illegal ones included
            a.copy_(torch.from_numpy(np.stack([gen.integers(0, 2, N), gen.integers(0, 3, N)], 1).astype(np.int8)).cuda())
        oa = cpu(a[lo:hi])
        b, d, r, te, tr, info = env.step(a)
        ores = orc.step(oa)
        for k, x in enumerate((b, d, r, te, tr, info)):
            x, o = cpu(x[lo:hi]), ores[k]
            assert np.array_equal(bits(x) if k == 2 else x, bits(o) if k == 2 else o), (kw, t, k)
        # every lane: board invariants
        bb = b.reshape(N, -1).to(torch.int16)
        assert int(((d < 1) | (d > cn)).sum()) == 0
        assert int((bb.abs() > cn).sum()) == 0
        for c in range(1, cn + 1):
            assert int(((bb == c).sum(1) > 1).sum()) == 0 and int(((bb == -c).sum(1) > 1).sum()) == 0
        assert bool(((info == 2) == (r > 0)).all())
        assert b.shape == (N, S, S)
        nterm += int(ores[3].sum())
    return nterm


def test_config5_7x7_mcts400_at_32768_lanes(ea):
    """BASELINE config 5 per GPU: 7x7, MctsAgent(num_simulations=400, num_env_copies=1) opponent
    (classical_policies/mcts.py:47-69), 32 768 lanes; 64 lanes against the oracle for 4 steps."""
    _slice_lockstep(ea, 32768, 20000, 20064, 4, board_size=7, opponent_policy="mcts", num_simulations=400, num_env_copies=1,
                    rng="philox", philox_key=2024)


def test_config5_7x7_mcts_10x5_and_illegal_actions(ea):
    """The reference's default MctsAgent (10 simulations x 5 env copies) on 7x7 at the same lane count, raw (possibly illegal)
    agent actions every third step, MT19937-compat dice."""
    _slice_lockstep(ea, 32768, 777, 777 + 96, 6, raw_every=3, board_size=7, opponent_policy="mcts", num_simulations=10,
                    num_env_copies=5, rng="mt19937", philox_key=7)


def test_config5_7x7_cube_layer4_generic_rollout(ea):
    """cube_layer 4 (10 cubes a side) takes the generic rollout kernel inside the same split-phase step."""
    _slice_lockstep(ea, 8192, 4000, 4064, 5, board_size=7, cube_layer=4, opponent_policy="mcts", num_simulations=6,
                    num_env_copies=2, rng="philox", philox_key=99)


def test_config4_a2c_on_shaped_env_with_minimax_opponent_65536_lanes(ea):
    """BASELINE config 4 per GPU: A2C on MiniMaxHeuristicEnv semantics (shaped rewards, illegal-move tolerance,
    envs/training_ewn.py:43-99) with the INTENDED depth-3 minimax opponent, 65 536 lanes, n_steps 5.
    Every transition of every rollout of a 96-lane slice -- observation, action taken, reward, done -- is replayed on the
    oracle; parameters must change and the statistics stay finite."""
    from ewn_gym_amd.a2c import A2CTrainer
    N, lo, hi, T, U = 65536, 40000, 40096, 5, 4
    kw = dict(max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0, illegal_move_tolerance=10,
              shaped_refresh_on_reset=True, philox_key=9487)
    env = ea.VecEWN(N, opponent_policy="minimax", autoreset=True, seed_stride=N, **kw)
    seeds = (np.arange(N, dtype=np.uint64) + 9487).astype(np.uint32)
    env.reset(seeds=seeds)
    orc = po.OracleVecEnv(hi - lo, opponent="minimax", autoreset=True, seed_stride=N, lane_offset=lo, **kw)
    ob, od = orc.reset(seeds=seeds[lo:hi])
    tr = A2CTrainer(env, n_steps=T, learning_rate=7e-4, seed=1)
    p0 = [p.detach().clone() for p in tr.model.parameters()]
    illegal = 0
    for u in range(U):                      # update 0 runs eagerly, 1 captures the rollout in a hipGraph, 2.. replay it
        stats = tr.stats_dict(tr.collect_and_update())
        assert all(np.isfinite(v) for v in stats.values()), stats
        boards, dices, acts = cpu(tr._boards[:, lo:hi]), cpu(tr._dices[:, lo:hi]), cpu(tr._acts[:, lo:hi])
        rews, dones = cpu(tr._rews[:, lo:hi]), cpu(tr._dones[:, lo:hi])
        for t in range(T):
            assert np.array_equal(boards[t], ob) and np.array_equal(dices[t], od), (u, t)   # the observation the policy saw
            ob, od, r, te, _, info = orc.step(acts[t])
            assert np.array_equal(rews[t], r.astype(np.float32)), (u, t)
            assert np.array_equal(dones[t], te.astype(np.float32)), (u, t)
            illegal += int(((info == 1) | (info == 5)).sum())
    assert np.array_equal(cpu(env.board[lo:hi]), ob) and np.array_equal(cpu(env.dice[lo:hi]), od)
    ps, tol, _ = orc.aux()
    assert np.array_equal(bits(cpu(env.prev_score[lo:hi])), bits(ps)) and np.array_equal(cpu(env.tolerance[lo:hi]), tol)
    assert illegal > 0                      # an untrained policy does play illegal moves: the tolerance path was exercised
    assert any(not torch.equal(a, b) for a, b in zip(p0, tr.model.parameters()))
    assert tr.num_timesteps == U * T * N


def test_config4_fused_a2c_65536_lanes(ea):
    """BASELINE config 4 per GPU on the fused path (FusedA2CTrainer: ewn_step_k_policy + ewn_a2c_grad + ewn_a2c_apply, what
    train_a2c.py runs by default): 65 536 lanes, shaped env, depth-3 minimax opponent, n_steps 5.  Every transition of every
    rollout of a 96-lane slice -- observation the policy saw, action, reward, flags -- is replayed on the oracle across eager,
    captured and replayed updates; the parameters move, the loss statistics are finite."""
    from ewn_gym_amd.a2c import FusedA2CTrainer
    N, lo, hi, T, U = 65536, 40000, 40096, 5, 4
    kw = dict(max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0, illegal_move_tolerance=10,
              shaped_refresh_on_reset=True, philox_key=9487)
    env = ea.VecEWN(N, opponent_policy="minimax", autoreset=True, seed_stride=N, **kw)
    seeds = (np.arange(N, dtype=np.uint64) + 9487).astype(np.uint32)
    env.reset(seeds=seeds)
    orc = po.OracleVecEnv(hi - lo, opponent="minimax", autoreset=True, seed_stride=N, lane_offset=lo, **kw)
    ob, od = orc.reset(seeds=seeds[lo:hi])
    tr = FusedA2CTrainer(env, n_steps=T, learning_rate=7e-4, seed=1)
    p0 = tr.params.clone()
    illegal = 0
    for u in range(U):                      # update 0 runs eagerly, 1 captures the five launches in a hipGraph, 2.. replay it
        tr.collect_and_update()
        stats = tr.stats_dict()
        assert all(np.isfinite(v) for v in stats.values()), stats
        tj = tr.traj
        for t in range(T):
            assert np.array_equal(cpu(tj["obs_board"][t, lo:hi]), ob) and np.array_equal(cpu(tj["obs_dice"][t, lo:hi]), od), (u, t)
            ob, od, r, te, trn, info = orc.step(cpu(tj["action"][t, lo:hi]))
            assert np.array_equal(bits(cpu(tj["reward"][t, lo:hi])), bits(r)), (u, t)
            assert np.array_equal(cpu(tj["terminated"][t, lo:hi]), te) and np.array_equal(cpu(tj["info"][t, lo:hi]), info), (u, t)
            illegal += int(((info == 1) | (info == 5)).sum())
        assert np.array_equal(cpu(tj["obs_board"][T, lo:hi]), ob)            # s_K: the bootstrap observation
    assert np.array_equal(cpu(env.board[lo:hi]), ob) and np.array_equal(cpu(env.dice[lo:hi]), od)
    ps, tol, _ = orc.aux()
    assert np.array_equal(bits(cpu(env.prev_score[lo:hi])), bits(ps)) and np.array_equal(cpu(env.tolerance[lo:hi]), tol)
    assert illegal > 0
    assert not torch.equal(p0, tr.params) and bool(torch.isfinite(tr.params).all())
    assert tr.num_timesteps == U * T * N
