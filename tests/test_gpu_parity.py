"""HIP engine (through the C ABI, via ewn_gym_amd's ctypes binding) against the CPU oracle
and against the golden vectors captured from the reference.  Bit-exact everywhere
(integer boards/actions/flags; fp64 heuristic values, root values and shaped rewards
compared by bit pattern)."""
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


MSG = {None: 0, "Invalid move for player! End the game.": 1, "You won!": 2,
       "Invalid move for opponent! End the game.": 3, "You lost!": 4}


def msg_code(m):
    if m is not None and m.startswith("Invalid move for player! Tolerance left"):
        return 5
    return MSG[m]


def boards_of(recs, S):
    return np.array([r["board"] for r in recs], np.int8).reshape(-1, S, S)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


def cpu(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- golden vectors

def test_g1_reset(ea, golden):
    g = golden("g1_initial.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        env = ea.VecEWN(len(recs), board_size=S, cube_layer=L, rng="mt19937")
        b, d = env.reset(seeds=[r["seed"] for r in recs])
        assert np.array_equal(cpu(b), boards_of(recs, S))
        assert cpu(d).tolist() == [r["dice"] for r in recs]


def test_g2_legal_actions(ea, golden):
    g = golden("g2_legal.json")
    for (S, L, pl) in sorted({(r["S"], r["L"], r["player"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"], r["player"]) == (S, L, pl)]
        acts, n, cs, cl, win = [cpu(t) for t in ea.legal_actions(boards_of(recs, S), [r["dice"] for r in recs], player=pl, cube_layer=L)]
        for i, r in enumerate(recs):
            assert bool(win[i]) == r["win"]
            if "legal" in r:
                assert acts[i, :n[i]].tolist() == r["legal"], (S, L, pl, i)
                assert (acts[i, n[i]:] == -1).all()
                assert (cs[i], cl[i]) == (r["cube_small"], r["cube_large"])


@pytest.mark.parametrize("h", ["hybrid", "min_dist", "two_min_dist", "attk"])
def test_g4_heuristics(ea, golden, h):
    g = golden("g4_eval.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        out = cpu(ea.evaluate(boards_of(recs, S), h, cube_layer=L))
        exp = np.array([float.fromhex(r[h]) for r in recs])
        assert np.array_equal(bits(out), bits(exp))


def test_g5_expectiminimax(ea, golden):
    g = golden("g5_minimax.json")
    n = 0
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        for key in sorted({k for r in recs for k in r["res"]}):
            sub = [r for r in recs if key in r["res"]]
            d, h = key.split("/")
            acts, vals = ea.predict_minimax(boards_of(sub, S), [r["dice"] for r in sub], int(d), h, cube_layer=L)
            acts, vals = cpu(acts), cpu(vals)
            for i, r in enumerate(sub):
                a0, a1, v = r["res"][key]
                assert acts[i].tolist() == [a0, a1], (S, L, key, i)
                assert float(vals[i]).hex() == float.fromhex(v).hex(), (S, L, key, i)
                n += 1
    assert n > 3000


def _run_group(ea, recs, opp, **kw):
    """Trajectories of one configuration run lane-parallel; finished lanes stay frozen."""
    S, L = recs[0].get("S", 5), recs[0].get("L", 3)
    N = len(recs)
    env = ea.VecEWN(N, board_size=S, cube_layer=L, opponent_policy=opp, rng="mt19937", **kw)
    b, d = env.reset(seeds=[r["seed"] for r in recs])
    b, d = cpu(b), cpu(d)
    for i, r in enumerate(recs):
        assert b[i].reshape(-1).tolist() == r["board0"] and int(d[i]) == r["dice0"]
    T = max(len(r["steps"]) for r in recs)
    for t in range(T):
        acts = np.zeros((N, 2), np.int8)
        for i, r in enumerate(recs):
            if t < len(r["steps"]):
                acts[i] = r["steps"][t]["a"]
        b, d, rew, te, tr, info = [cpu(x) for x in env.step(acts)]
        for i, r in enumerate(recs):
            if t < len(r["steps"]):
                st = r["steps"][t]
                ctx = (r["seed"], r.get("rule"), t)
                assert b[i].reshape(-1).tolist() == st["board"], ctx
                assert int(d[i]) == st["dice"], ctx
                assert float(rew[i]).hex() == float.fromhex(st["r"]).hex(), ctx
                assert (bool(te[i]), bool(tr[i])) == (st["term"], st["trunc"]), ctx
                assert int(info[i]) == msg_code(st["msg"]), ctx
            else:  # frozen lane
                assert bool(te[i]) and float(rew[i]) == 0.0 and b[i].reshape(-1).tolist() == r["steps"][-1]["board"]


def test_g3_trajectories_random_opponent(ea, golden):
    g = golden("g3_traj_random.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        _run_group(ea, [r for r in g if (r["S"], r["L"]) == (S, L)], "random")


def test_g6_trajectories_minimax_opponent(ea, golden):
    g = golden("g6_traj_minimax.json")
    for key in sorted({(r["S"], r["L"], r["depth"], r["heuristic"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"], r["depth"], r["heuristic"]) == key]
        _run_group(ea, recs, "minimax", max_depth=key[2], heuristic=key[3])


def test_g7_shaped_env(ea, golden):
    for grp in golden("g7_shaped.json"):
        env = ea.VecEWN(1, opponent_policy="random", rng="mt19937", shaped=True, illegal_move_tolerance=grp["tol"],
                        reward=1.0, illegal_move_reward=-1.0)
        assert float(cpu(env.prev_score)[0]).hex() == float.fromhex(grp["ctor_prev_score"]).hex()
        for rec in grp["episodes"]:
            b, d = env.reset(seeds=[rec["seed"]])
            assert cpu(b).reshape(-1).tolist() == rec["board0"] and int(cpu(d)[0]) == rec["dice0"]
            for t, st in enumerate(rec["steps"]):
                b, d, r, te, tr, info = [cpu(x) for x in env.step([st["a"]])]
                ctx = (grp["tol"], rec["seed"], t)
                assert b.reshape(-1).tolist() == st["board"], ctx
                assert int(d[0]) == st["dice"], ctx
                assert float(r[0]).hex() == float.fromhex(st["r"]).hex(), ctx
                assert (bool(te[0]), bool(tr[0])) == (st["term"], st["trunc"]), ctx
                assert int(info[0]) == msg_code(st["msg"]), ctx
            assert int(cpu(env.tolerance)[0]) == rec["tol_after"]
            assert float(cpu(env.prev_score)[0]).hex() == float.fromhex(rec["prev_score_after"]).hex()


# ---------------------------------------------------------------- oracle, seeded random lanes

def _lockstep(ea, N, steps, seed0=1000, check_terminal=True, **kw):
    """Same seeds, same actions, HIP engine vs CPU oracle, with auto-reset."""
    okw = dict(kw)
    opp = okw.pop("opponent_policy", "random")
    env = ea.VecEWN(N, opponent_policy=opp, autoreset=True, want_terminal_obs=check_terminal, **okw)
    okw.pop("mt_window", None)
    okw.pop("use_tables", None)
    orc = po.OracleVecEnv(N, opponent=opp, autoreset=True, **okw)
    seeds = np.arange(N, dtype=np.uint32) * 7919 + seed0
    b, d = env.reset(seeds=seeds)
    ob, od = orc.reset(seeds=seeds)
    assert np.array_equal(cpu(b), ob) and np.array_equal(cpu(d), od)
    gen = np.random.Generator(np.random.PCG64(seed0))
    nterm = 0
    for t in range(steps):
        if t % 3 == 2:   # raw actions (illegal ones included) every third step
            acts = np.stack([gen.integers(0, 2, N), gen.integers(0, 3, N)], 1).astype(np.int8)
        else:
            acts = orc.sample_legal_actions(t)
            assert np.array_equal(cpu(env.sample_legal_actions(t)), acts), t
        res = [cpu(x) for x in env.step(acts)]
        ores = orc.step(acts, want_terminal=True)
        names = ("board", "dice", "reward", "terminated", "truncated", "info")
        for k, name in enumerate(names):
            a, o = res[k], ores[k]
            if name == "reward":
                a, o = bits(a), bits(o)
            assert np.array_equal(a, o), (t, name, np.nonzero(np.asarray(a != o).reshape(N, -1).any(1))[0][:5])
        if check_terminal:
            assert np.array_equal(cpu(env.terminal_board), ores[6]), t
            assert np.array_equal(cpu(env.terminal_dice), ores[7]), t
        nterm += int(ores[3].sum())
    return nterm


@pytest.mark.parametrize("rng", ["mt19937", "philox"])
def test_roll_dice_vs_oracle(ea, rng):
    """ewn_roll_dice (envs/ewn.py:90-92) on masked lanes, interleaved with steps: dice and the streams behind them stay the oracle's"""
    N = 2000
    kw = dict(opponent_policy="random", rng=rng, philox_key=77)
    env = ea.VecEWN(N, autoreset=False, **kw)
    orc = po.OracleVecEnv(N, opponent="random", autoreset=False, rng=rng, philox_key=77)
    seeds = np.arange(N, dtype=np.uint32) * 13 + 5
    env.reset(seeds=seeds)
    orc.reset(seeds=seeds)
    gen = np.random.Generator(np.random.PCG64(3))
    for t in range(25):
        mask = (gen.integers(0, 3, N) > 0).astype(np.uint8) if t % 2 else None
        d = cpu(env.roll_dice(mask))
        assert np.array_equal(d, orc.roll_dice(mask)), t
        acts = orc.sample_legal_actions(t)
        res = [cpu(x) for x in env.step(acts)]
        ores = orc.step(acts)
        for a, o in zip(res, ores):
            assert np.array_equal(a, o), t
    assert cpu(env.done).sum() > N // 2     # finished lanes kept their dice through the later calls


@pytest.mark.parametrize("rng", ["mt19937", "philox"])
def test_step_random_opponent_vs_oracle(ea, rng):
    assert _lockstep(ea, 3000, 40, rng=rng, philox_key=0x1234ABCD5678) > 3000


@pytest.mark.parametrize("depth,heur", [(1, "hybrid"), (2, "attk"), (3, "hybrid"), (3, "two_min_dist"), (3, "min_dist")])
def test_step_minimax_opponent_vs_oracle(ea, depth, heur):
    assert _lockstep(ea, 1500, 30, opponent_policy="minimax", max_depth=depth, heuristic=heur, rng="mt19937") > 1000


@pytest.mark.parametrize("N,steps,kw", [
    (900, 25, dict(max_depth=3, rng="mt19937")), (900, 20, dict(max_depth=4, rng="philox", philox_key=9)), (900, 20, dict(max_depth=2, rng="philox")),
    (40000, 10, dict(max_depth=3, rng="philox", philox_key=3)), (132000, 5, dict(max_depth=3, rng="philox", philox_key=4)),
    (160, 10, dict(max_depth=5, rng="mt19937")), (160, 8, dict(max_depth=6, rng="philox", philox_key=6)), (33000, 4, dict(max_depth=5, rng="philox", philox_key=7)),
    (700, 20, dict(max_depth=3, rng="philox", board_size=7)), (500, 16, dict(max_depth=4, rng="mt19937", board_size=8)),
    (600, 25, dict(max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_tolerance=3)),
], ids=lambda v: str(v) if isinstance(v, int) else "-".join("%s=%s" % kv for kv in sorted(v.items())))
def test_two_min_dist_on_the_fused_step_kernel(ea, N, steps, kw):
    """'two_min_dist' (envs/minimax_ewn.py:133-178) on k_step_d3's own instances: four, two and one lane(s) per game, every depth class,
    both dice kinds, larger boards, the shaped env -- lock step with the oracle."""
    import ctypes as C
    from ewn_gym_amd import _lib
    probe = ea.VecEWN(N, opponent_policy="minimax", heuristic="two_min_dist", autoreset=True, **kw)
    assert _lib.load().ewn_lanes_per_game(C.byref(probe.cfg), 0) > 0   # the table-driven kernel, not the generic one
    del probe
    _lockstep(ea, N, steps, opponent_policy="minimax", heuristic="two_min_dist", **kw)


def test_step_minimax_depth4_5_vs_oracle(ea):
    _lockstep(ea, 200, 12, opponent_policy="minimax", max_depth=4, rng="philox", philox_key=5)
    _lockstep(ea, 64, 10, opponent_policy="minimax", max_depth=5, rng="mt19937")


@pytest.mark.parametrize("S,L", [(6, 3), (7, 3), (7, 4), (7, 5), (8, 3)])
def test_step_other_board_sizes_vs_oracle(ea, S, L):
    _lockstep(ea, 700, 50, board_size=S, cube_layer=L, rng="mt19937")
    if L == 3 or L == 4:
        _lockstep(ea, 300, 20, board_size=S, cube_layer=L, opponent_policy="minimax", max_depth=3, rng="philox")


def test_step_small_cube_layers(ea):
    _lockstep(ea, 500, 30, board_size=5, cube_layer=2, rng="mt19937")
    _lockstep(ea, 500, 30, board_size=4, cube_layer=1, rng="mt19937")
    with pytest.raises(AssertionError):   # assert cube_layer < board_size - 1, envs/ewn.py:47
        ea.VecEWN(4, board_size=5, cube_layer=4)
    with pytest.raises(ea.EwnError):      # dice loop 1..6 needs cube_num >= 6 (IndexError upstream)
        ea.VecEWN(4, board_size=5, cube_layer=2, opponent_policy="minimax")


def test_shaped_env_vs_oracle(ea):
    _lockstep(ea, 2000, 40, shaped=True, illegal_move_tolerance=4, reward=10.0, illegal_move_reward=-0.5, rng="mt19937")
    _lockstep(ea, 500, 40, shaped=True, shaped_refresh_on_reset=True, opponent_policy="minimax", max_depth=2, rng="philox")


def test_mt_window_overflow_closed_form(ea):
    """A 16-word window forces the memory-free closed-form path for later draws."""
    _lockstep(ea, 600, 45, rng="mt19937", mt_window=16)
    _lockstep(ea, 300, 60, board_size=7, cube_layer=5, rng="mt19937", mt_window=16)


@pytest.mark.parametrize("kw", [dict(opponent_policy="minimax", max_depth=3, rng="philox"),
                                dict(opponent_policy="random", rng="mt19937"),
                                dict(opponent_policy="minimax", max_depth=2, rng="philox", board_size=7),
                                dict(opponent_policy="minimax", max_depth=3, rng="mt19937", use_tables=False)])
def test_fused_random_agent_output(ea, kw):
    """ewn_step_out.random_action: RandomAgent.predict on the post-step observation, fused into the step kernel and
    fed back as the next action (the buffer aliases `actions`), lock-step against the oracle's restatement."""
    N = 1536
    okw = dict(kw)
    opp = okw.pop("opponent_policy")
    okw.pop("use_tables", None)
    env = ea.VecEWN(N, autoreset=True, want_random_action=True, lane_offset=77, philox_key=99, **kw)
    orc = po.OracleVecEnv(N, opponent=opp, autoreset=True, lane_offset=77, philox_key=99, **okw)
    seeds = np.arange(N, dtype=np.uint32) + 31
    env.reset(seeds=seeds)
    orc.reset(seeds=seeds)
    acts = orc.random_actions()
    buf = env.random_action
    buf.copy_(torch.from_numpy(acts))
    S = kw.get("board_size", 5)
    for t in range(25):
        res = [cpu(x) for x in env.step(buf)]     # reads buf, writes the next random legal action into buf
        ores = orc.step(acts)
        for a, o in zip(res, ores):
            assert np.array_equal(a, o), t
        acts = orc.random_actions()
        got = cpu(buf)
        assert np.array_equal(got, acts), (t, np.nonzero((got != acts).any(1))[0][:5])
        la, n, _, _, _ = po.legal_actions(ores[0], ores[1], player=1)
        ok = ((la[:, :, 0] == acts[:, None, 0]) & (la[:, :, 1] == acts[:, None, 1])).any(1)
        assert ok.all()


@pytest.mark.parametrize("use_tables", [True, False])
def test_mt_window_rotation_survives_starvation_and_checkpoint(ea, use_tables):
    """MT kind: (a) wipe every lane's READY/X/Y bookkeeping so the next auto-resets find no prepared window (the
    rebuild-in-place path), (b) checkpoint / restore mid-run -- dice must keep matching the oracle's numpy stream."""
    N = 1024
    env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="mt19937", autoreset=True, use_tables=use_tables)
    orc = po.OracleVecEnv(N, opponent="minimax", max_depth=3, rng="mt19937", autoreset=True)
    seeds = np.arange(N, dtype=np.uint32) * 3 + 11
    env.reset(seeds=seeds)
    orc.reset(seeds=seeds)
    sd = None
    for t in range(40):
        if t == 6:
            env.rng_state.view(-1)[3:4 * N:4] &= 0x31          # keep overflow bit + current slot, drop READY / X / Y
        if t == 15:
            sd = env.state_dict()
            saved_orc_step = t
        a = orc.sample_legal_actions(t)
        res = [cpu(x) for x in env.step(a)]
        ores = orc.step(a)
        for x, o in zip(res, ores):
            assert np.array_equal(x, o), t
    # restore the checkpoint into a fresh engine and replay the same steps against a fresh oracle run
    env2 = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="mt19937", autoreset=True, use_tables=use_tables)
    env2.load_state_dict(sd)
    orc2 = po.OracleVecEnv(N, opponent="minimax", max_depth=3, rng="mt19937", autoreset=True)
    orc2.reset(seeds=seeds)
    for t in range(saved_orc_step):
        orc2.step(orc2.sample_legal_actions(t))
    for t in range(saved_orc_step, 40):
        a = orc2.sample_legal_actions(t)
        res = [cpu(x) for x in env2.step(a)]
        ores = orc2.step(a)
        for x, o in zip(res, ores):
            assert np.array_equal(x, o), t


def test_frozen_lanes_without_autoreset(ea):
    N = 512
    env = ea.VecEWN(N, rng="mt19937")
    orc = po.OracleVecEnv(N, rng="mt19937")
    seeds = np.arange(N) + 5
    env.reset(seeds=seeds)
    orc.reset(seeds=seeds)
    for t in range(25):
        acts = orc.sample_legal_actions(t)
        res = [cpu(x) for x in env.step(acts)]
        ores = orc.step(acts)
        for a, o in zip(res, ores):
            assert np.array_equal(a, o), t
    assert cpu(env.done).all()
    mask = (np.arange(N) % 3 == 0)
    b, d = env.reset(seeds=seeds + 1, mask=mask)
    ob, od = orc.reset(seeds=seeds + 1, mask=mask)
    assert np.array_equal(cpu(b), ob) and np.array_equal(cpu(d), od)
    assert np.array_equal(cpu(env.done) == 0, mask)


# ---------------------------------------------------------------- stateless policies

def _random_positions(S, L, n, seed, max_steps=14):
    """Reachable non-terminal positions from every phase of the game (opening to few-cube endgames)."""
    orc = po.OracleVecEnv(n, board_size=S, cube_layer=L, rng="philox", philox_key=seed, autoreset=True)
    orc.reset(seeds=np.arange(n) + seed)
    gen = np.random.Generator(np.random.PCG64(seed))
    when = gen.integers(0, max_steps, n)
    out, _ = orc.obs()
    out = out.copy()
    for t in range(max_steps):
        b = orc.step(orc.sample_legal_actions(t))[0]
        out[when == t] = b[when == t]
    return out, gen.integers(1, 7, n).astype(np.int8)


@pytest.mark.parametrize("S,L,depth,heur,n", [(5, 3, 3, "hybrid", 6000), (5, 3, 2, "min_dist", 1500), (5, 3, 4, "hybrid", 300),
                                             (5, 3, 5, "hybrid", 60), (7, 3, 3, "hybrid", 1500), (7, 4, 3, "attk", 500),
                                             (6, 3, 3, "two_min_dist", 500), (8, 5, 2, "hybrid", 300), (5, 3, 6, "hybrid", 8)])
def test_predict_minimax_vs_oracle(ea, S, L, depth, heur, n):
    b, d = _random_positions(S, L, n, 77 + depth)
    acts, vals = ea.predict_minimax(b, d, depth, heur, cube_layer=L)
    oa, ov, _ = po.predict_minimax(b, d, depth, heur, cube_layer=L)
    assert np.array_equal(cpu(acts), oa)
    assert np.array_equal(bits(cpu(vals)), bits(ov))


@pytest.mark.parametrize("S", [5, 6, 7, 8])
def test_depth3_specialised_kernel_equals_generic_and_oracle(ea, S):
    """The table-driven depth-3 kernel (ewn_fast.hpp) vs the template-recursive one vs the oracle."""
    n = 20000 if S == 5 else 4000
    b, d = _random_positions(S, 3, n, 1234 + S, max_steps=14 if S == 5 else 24)
    fa, fv = ea.predict_minimax(b, d, 3, "hybrid", use_tables=True)
    ga, gv = ea.predict_minimax(b, d, 3, "hybrid", use_tables=False)
    oa, ov, _ = po.predict_minimax(b, d, 3, "hybrid")
    assert np.array_equal(cpu(fa), oa) and np.array_equal(cpu(ga), oa)
    assert np.array_equal(bits(cpu(fv)), bits(ov)) and np.array_equal(bits(cpu(gv)), bits(ov))


def test_step_generic_depth3_kernel_still_covered(ea):
    _lockstep(ea, 1000, 20, opponent_policy="minimax", max_depth=3, rng="philox", use_tables=False)


def test_pruned_search_differs_from_full_width_somewhere(ea, golden):
    """alpha-beta is passed through chance nodes, which changes ~1 % of the moves.
    The engine must reproduce the pruned answer, so the goldens must contain such cases."""
    b, d = _random_positions(5, 3, 6000, 4242)
    acts, vals = ea.predict_minimax(b, d, 3, "hybrid")
    oa, ov, leaves = po.predict_minimax(b, d, 3, "hybrid")
    assert np.array_equal(cpu(acts), oa)
    assert leaves.max() <= 216 and leaves.min() >= 1


def test_predict_mcts_bit_exact_vs_oracle(ea):
    b, d = _random_positions(5, 3, 96, 9)
    ids = np.arange(96, dtype=np.uint32) * 977 + 3
    acts, wins = ea.predict_mcts(b, d, num_simulations=6, num_env_copies=5, key=0xFEEDFACE12345678, obs_id=ids)
    oa, ow = po.predict_mcts(b, d, num_simulations=6, num_env_copies=5, key=0xFEEDFACE12345678, obs_id=ids)
    assert np.array_equal(cpu(wins), ow)
    assert np.array_equal(cpu(acts), oa)
    b, d = _random_positions(7, 3, 32, 10)
    acts, wins = ea.predict_mcts(b, d, num_simulations=8, num_env_copies=1, key=3)
    oa, ow = po.predict_mcts(b, d, num_simulations=8, num_env_copies=1, key=3)
    assert np.array_equal(cpu(wins), ow) and np.array_equal(cpu(acts), oa)


@pytest.mark.parametrize("S,L,n,total", [(5, 3, 24, 7), (6, 3, 40, 20), (6, 3, 33, 100), (8, 3, 21, 130), (5, 3, 50, 400), (5, 3, 7, 1),
                                         (7, 4, 24, 9), (7, 5, 16, 5), (8, 5, 12, 11)])
def test_predict_mcts_every_geometry_and_group_size(ea, S, L, n, total):
    """cube_layer <= 3 runs the byte-per-cube playout (ewn_playout.hpp) with 8..64 lanes per root move, larger layers the
    generic one; both must reproduce the oracle's win counts exactly, late-game positions (few cubes) included."""
    b, d = _random_positions(S, L, n, 100 + S + L, max_steps=30)
    d = np.minimum(d, L * (L + 1) // 2).astype(np.int8)
    acts, wins = ea.predict_mcts(b, d, num_simulations=total, num_env_copies=1, key=S * 1000 + total, cube_layer=L)
    oa, ow = po.predict_mcts(b, d, num_simulations=total, num_env_copies=1, key=S * 1000 + total, cube_layer=L)
    assert np.array_equal(cpu(wins), ow) and np.array_equal(cpu(acts), oa)


@pytest.mark.parametrize("S,L,n_sims,first", [(5, 3, 100, 2), (5, 3, 37, 1), (7, 3, 64, 2), (8, 3, 9, 1), (7, 4, 20, 2)])
def test_playout_wins_bit_exact_vs_oracle(ea, S, L, n_sims, first):
    """MinimaxEnv.simulate (envs/minimax_ewn.py:215-238) on reachable positions AND on finished games (no playout is played)."""
    b, _ = _random_positions(S, L, 48, 500 + S, max_steps=30)
    done = b[:6].copy()
    done[0][done[0] == 1] = 0
    done[0, S - 1, S - 1] = 1      # TOP_LEFT already home
    done[1][done[1] == -1] = 0
    done[1, 0, 0] = -1             # BOTTOM_RIGHT already home
    done[2][done[2] < 0] = 0       # no negative cube left
    done[3][done[3] > 0] = 0       # no positive cube left
    b = np.concatenate([b, done[:4]])
    wins = ea.playout_wins(b, first_player=first, n_sims=n_sims, key=77 + S, cube_layer=L)
    assert np.array_equal(cpu(wins), po.playout_wins(b, first, n_sims, key=77 + S, cube_layer=L))


def test_g9_mcts_statistics_vs_reference(ea, golden):
    g = golden("g9_mcts.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        nsim = 2000
        _, wins = ea.predict_mcts(boards_of(recs, S), [r["dice"] for r in recs], num_simulations=nsim, num_env_copies=1,
                                  key=99, cube_layer=L)
        wins = cpu(wins)
        for i, r in enumerate(recs):
            for j, w in enumerate(r["wins"]):
                p = (w + wins[i, j]) / (r["n"] + nsim)
                sigma = max(1e-9, (p * (1 - p) * (1 / r["n"] + 1 / nsim)) ** 0.5)
                assert abs(w / r["n"] - wins[i, j] / nsim) <= 5 * sigma + 1e-9, (S, i, j)


def test_step_mcts_opponent_vs_oracle(ea):
    _lockstep(ea, 192, 14, opponent_policy="mcts", num_simulations=3, num_env_copies=2, rng="philox", philox_key=11)
    _lockstep(ea, 64, 10, opponent_policy="mcts", num_simulations=2, num_env_copies=2, rng="mt19937", philox_key=12)


# ---------------------------------------------------------------- full size: properties

def test_full_size_properties_and_sharding(ea):
    """BASELINE size (65 536 lanes, depth-3 opponent): size-independent checks --
    determinism, invariance to how lanes are sharded, board invariants, reward/flag
    consistency, and a sampled slice checked against the oracle."""
    N, T = 65536, 12
    kw = dict(opponent_policy="minimax", max_depth=3, rng="philox", philox_key=2024, autoreset=True, seed_stride=N)
    seeds = (np.arange(N, dtype=np.uint64) + 9487).astype(np.uint32)

    def run(lo, hi):
        env = ea.VecEWN(hi - lo, lane_offset=lo, **kw)
        env.reset(seeds=seeds[lo:hi])
        out = []
        for t in range(T):
            a = env.sample_legal_actions(t).clone()
            res = env.step(a)
            out.append([x.clone() for x in res] + [a])
        return out

    full = run(0, N)
    again = run(0, N)
    halves = [run(0, N // 2), run(N // 2, N)]
    for t in range(T):
        for k in range(7):
            assert torch.equal(full[t][k], again[t][k])
            assert torch.equal(full[t][k], torch.cat([halves[0][t][k], halves[1][t][k]]))
        b, d, r, te, tr, info, a = full[t]
        bb = b.reshape(N, -1).to(torch.int16)
        for c in range(1, 7):
            assert int(((bb == c).sum(1) > 1).sum()) == 0 and int(((bb == -c).sum(1) > 1).sum()) == 0
        assert int(((d < 1) | (d > 6)).sum()) == 0
        assert bool(((r != 0) == (te == 1)).all())          # legal agents: reward only on terminal transitions
        assert bool(((info == 2) == (r > 0)).all()) and bool(((info == 4) == (r < 0)).all())
        assert int(tr.sum()) == 0
    # sampled slice against the oracle
    lo, hi = 30000, 30512
    orc = po.OracleVecEnv(hi - lo, opponent="minimax", max_depth=3, rng="philox", philox_key=2024, autoreset=True,
                          seed_stride=N, lane_offset=lo)
    orc.reset(seeds=seeds[lo:hi])
    for t in range(T):
        a = orc.sample_legal_actions(t)
        assert np.array_equal(cpu(full[t][6][lo:hi]), a)
        ores = orc.step(a)
        for k in range(6):
            x, o = cpu(full[t][k][lo:hi]), ores[k]
            assert np.array_equal(bits(x) if k == 2 else x, bits(o) if k == 2 else o), (t, k)


def test_one_and_two_lanes_per_game_variants_at_large_lane_counts(ea):
    """>= 131 072 lanes select the T = 1 instance of the lean step kernel (one lane per game), 32 768..131 071 the T = 2 one; a
    slice against the oracle, 7x7 included (64-bit occupancy masks)."""
    for S, N, T in ((5, 262144, 8), (7, 100000, 6), (7, 131072, 4)):
        kw = dict(board_size=S, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=77, autoreset=True, seed_stride=N)
        seeds = (np.arange(N, dtype=np.uint64) * 3 + 5).astype(np.uint32)
        env = ea.VecEWN(N, **kw)
        env.reset(seeds=seeds)
        lo, hi = 70001, 70001 + 384
        orc = po.OracleVecEnv(hi - lo, board_size=S, opponent="minimax", max_depth=3, rng="philox", philox_key=77, autoreset=True,
                              seed_stride=N, lane_offset=lo)
        orc.reset(seeds=seeds[lo:hi])
        for t in range(T):
            a = env.sample_legal_actions(t).clone()
            oa = orc.sample_legal_actions(t)
            assert np.array_equal(cpu(a[lo:hi]), oa)
            res = env.step(a)
            ores = orc.step(oa)
            for k in range(6):
                x, o = cpu(res[k][lo:hi]), ores[k]
                assert np.array_equal(bits(x) if k == 2 else x, bits(o) if k == 2 else o), (S, t, k)


def test_ragged_and_empty_inputs(ea):
    """Edge shapes: no observations, one lane, lane counts that do not fill a block or a wave, out-of-range dice in queries."""
    e = np.zeros((0, 5, 5), np.int8)
    assert tuple(ea.predict_minimax(e, np.zeros(0, np.int8), 3)[0].shape) == (0, 2)
    assert tuple(ea.predict_mcts(e, np.zeros(0, np.int8), num_simulations=4, num_env_copies=1)[0].shape) == (0, 2)
    assert tuple(ea.playout_wins(e, first_player=1, n_sims=8).shape) == (0,)
    assert tuple(ea.evaluate(e).shape) == (0,)
    for N in (1, 63, 257, 1000):
        for opp, kw in (("random", {}), ("minimax", {"max_depth": 3}), ("mcts", {"num_simulations": 2, "num_env_copies": 2})):
            _lockstep(ea, N, 6, seed0=N, opponent_policy=opp, rng="philox", philox_key=N, **kw)
    # a dice value the game cannot produce (0, 7) must not fault: the stateless policies answer (-1, -1) for it
    b, _ = _random_positions(5, 3, 64, 321)
    for dv in (0, 7):
        acts, _ = ea.predict_minimax(b, np.full(64, dv, np.int8), 3)
        assert (cpu(acts) == -1).all()


def test_mt19937_windows_at_scale(ea):
    """MT19937-compat dice with the T = 2 kernel instance and one refill block per step block: enough steps for every lane to
    rotate through its three windows several times; a slice against the oracle (numpy's stream)."""
    N, T = 40000, 45
    kw = dict(opponent_policy="minimax", max_depth=3, rng="mt19937", autoreset=True, seed_stride=N)
    seeds = (np.arange(N, dtype=np.uint64) * 5 + 11).astype(np.uint32)
    env = ea.VecEWN(N, **kw)
    env.reset(seeds=seeds)
    lo, hi = 33333, 33333 + 320
    orc = po.OracleVecEnv(hi - lo, opponent="minimax", max_depth=3, rng="mt19937", autoreset=True, seed_stride=N, lane_offset=lo)
    orc.reset(seeds=seeds[lo:hi])
    nterm = 0
    for t in range(T):
        a = env.sample_legal_actions(t).clone()
        oa = orc.sample_legal_actions(t)
        assert np.array_equal(cpu(a[lo:hi]), oa)
        res = env.step(a)
        ores = orc.step(oa)
        for k in range(6):
            x, o = cpu(res[k][lo:hi]), ores[k]
            assert np.array_equal(bits(x) if k == 2 else x, bits(o) if k == 2 else o), (t, k)
        nterm += int(ores[3].sum())
    assert nterm > 4 * (hi - lo)   # every lane of the slice has gone through several episodes


def test_g12_maximum_board_size(ea, golden):
    """8x8 (the largest supported board) against vectors produced by the reference itself: heuristics, depth 1-3 searches
    (depth 3 'hybrid' through the table-driven kernel), trajectories vs Random and vs minimax(3)."""
    g = golden("g12_maxboard.json")
    S, L = g["S"], g["L"]
    boards = np.array([r["board"] for r in g["eval"]], np.int8).reshape(-1, S, S)
    for h in ("hybrid", "min_dist", "two_min_dist", "attk"):
        vals = cpu(ea.evaluate(boards, h, cube_layer=L))
        assert [float(v).hex() for v in vals] == [float.fromhex(r[h]).hex() for r in g["eval"]], h
    recs = g["minimax"]
    for key in sorted({k for r in recs for k in r["res"]}):
        d, h = key.split("/")
        acts, vals = ea.predict_minimax(np.array([r["board"] for r in recs], np.int8).reshape(-1, S, S), [r["dice"] for r in recs],
                                        int(d), h, cube_layer=L)
        acts, vals = cpu(acts), cpu(vals)
        for i, r in enumerate(recs):
            a0, a1, v = r["res"][key]
            assert acts[i].tolist() == [a0, a1] and float(vals[i]).hex() == float.fromhex(v).hex(), (key, i)
    _run_group(ea, [r for r in g["traj"] if r["opp"] == "random"], "random")
    _run_group(ea, [r for r in g["traj"] if r["opp"] == "minimax"], "minimax", max_depth=3, heuristic="hybrid")


def test_g13_cube_layers_4_and_5(ea, golden):
    """cube_layer 4 / 5 (10 / 15 cubes a side) against vectors produced by the reference: searches and minimax-opponent trajectories"""
    for grp in golden("g13_layers.json"):
        S, L, recs = grp["S"], grp["L"], grp["minimax"]
        for key in sorted({k for r in recs for k in r["res"]}):
            d, h = key.split("/")
            acts, vals = ea.predict_minimax(np.array([r["board"] for r in recs], np.int8).reshape(-1, S, S), [r["dice"] for r in recs],
                                            int(d), h, cube_layer=L)
            acts, vals = cpu(acts), cpu(vals)
            for i, r in enumerate(recs):
                a0, a1, v = r["res"][key]
                assert acts[i].tolist() == [a0, a1] and float(vals[i]).hex() == float.fromhex(v).hex(), (S, L, key, i)
        _run_group(ea, grp["traj"], "minimax", max_depth=2, heuristic="hybrid")


def test_no_kernel_writes_outside_its_buffers(ea, monkeypatch):
    """Every tensor VecEWN hands to the C ABI is carved out of a larger allocation with 4 KB of 0xA5 on either side; after stepping
    (auto-reset, both RNG kinds, every opponent, every lanes-per-game choice of the lean kernel) the guards must be intact.
    (Added after a refill queue sized for 4 lanes per game let a one-lane-per-game launch write 8 KB past it.)"""
    from ewn_gym_amd import vec_env
    G = 4096
    guarded = []
    real_zeros = torch.zeros

    def guarded_zeros(shape, dtype=torch.float32, device=None):
        if device is None or torch.device(device).type != "cuda":
            return real_zeros(shape, dtype=dtype, device=device)
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        pad = (nbytes + 511) // 512 * 512
        buf = torch.full((G + pad + G,), 0xA5, dtype=torch.uint8, device=device)
        buf[G:G + pad] = 0
        guarded.append((buf, pad))
        return buf[G:G + nbytes].view(dtype).view(shape)

    cases = [dict(n=64, opponent_policy="minimax", max_depth=5, rng="mt19937"), dict(n=64, opponent_policy="random", rng="mt19937"),
             dict(n=300, opponent_policy="minimax", max_depth=3, rng="mt19937"), dict(n=3000, opponent_policy="random", rng="mt19937"),
             dict(n=257, opponent_policy="minimax", max_depth=3, rng="philox"), dict(n=40000, opponent_policy="minimax", max_depth=3, rng="mt19937"),
             dict(n=100, opponent_policy="mcts", num_simulations=3, num_env_copies=2, rng="philox"),
             dict(n=70, opponent_policy="minimax", max_depth=2, heuristic="attk", rng="mt19937", board_size=7, cube_layer=4)]
    # the cached table images are handed to the C ABI too (and were the buffer the r01 overrun landed in): rebuild them
    # through the guarded allocator; their guards are checked after every case below and the cache is dropped at the end
    saved_tables = dict(vec_env._TABLES)
    vec_env._TABLES.clear()
    monkeypatch.setattr(vec_env.torch, "zeros", guarded_zeros)
    try:
        for (S, L) in ((5, 3), (7, 3), (6, 3), (8, 3)):
            assert vec_env.search_tables(S, L, torch.device("cuda")) is not None
    finally:
        monkeypatch.setattr(vec_env.torch, "zeros", real_zeros)
    table_guards = list(guarded)
    guarded.clear()
    assert len(table_guards) == 4

    def tables_intact():
        return all(bool((buf[:G] == 0xA5).all()) and bool((buf[G + pad:] == 0xA5).all()) for buf, pad in table_guards)

    for kw in cases:
        kw = dict(kw)
        n = kw.pop("n")
        monkeypatch.setattr(vec_env.torch, "zeros", guarded_zeros)
        try:
            env = ea.VecEWN(n, autoreset=True, want_terminal_obs=True, want_random_action=True, **kw)
        finally:
            monkeypatch.setattr(vec_env.torch, "zeros", real_zeros)
        env.reset(seeds=np.arange(n) + 3)
        for t in range(25):
            env.step(env.sample_legal_actions(t))
        torch.cuda.synchronize()
        for buf, pad in guarded:
            assert bool((buf[:G] == 0xA5).all()) and bool((buf[G + pad:] == 0xA5).all()), kw
        assert tables_intact(), kw
        guarded.clear()
    # ewn_step_k: state, trajectory ([K][N] columns) and per-lane totals, every lanes-per-game choice, partial last blocks
    for kw in (dict(n=1000, opponent_policy="minimax", max_depth=3, rng="philox"), dict(n=257, opponent_policy="random", rng="philox"),
               dict(n=40000, opponent_policy="minimax", max_depth=3, rng="philox"), dict(n=140000, opponent_policy="minimax", max_depth=2, rng="philox"),
               dict(n=300, opponent_policy="minimax", max_depth=5, rng="mt19937", autoreset=False, agent="minimax"),
               dict(n=513, opponent_policy="minimax", max_depth=4, rng="philox", board_size=8, agent="minimax")):
        kw = dict(kw)
        n, agent, autoreset = kw.pop("n"), kw.pop("agent", "random"), kw.pop("autoreset", True)
        monkeypatch.setattr(vec_env.torch, "zeros", guarded_zeros)
        try:
            env = ea.VecEWN(n, autoreset=autoreset, **kw)
            traj, tot = env.alloc_rollout(7), env.alloc_totals()
        finally:
            monkeypatch.setattr(vec_env.torch, "zeros", real_zeros)
        env.reset(seeds=np.arange(n) + 3)
        for _ in range(3):
            env.rollout(7, agent=agent, agent_max_depth=3, traj=traj, totals=tot)
            env.rollout(5, agent=agent, agent_max_depth=3)
        torch.cuda.synchronize()
        for buf, pad in guarded:
            assert bool((buf[:G] == 0xA5).all()) and bool((buf[G + pad:] == 0xA5).all()), kw
        assert tables_intact(), kw
        guarded.clear()
    # the stateless queries allocate their outputs with torch.zeros too
    for (S, L, M) in ((5, 3, 1), (5, 3, 63), (7, 3, 1000), (8, 5, 77), (6, 4, 130)):
        b, d = _random_positions(S, L, M, 900 + M, max_steps=20)
        d = np.minimum(d, 6).astype(np.int8)
        monkeypatch.setattr(vec_env.torch, "zeros", guarded_zeros)
        try:
            ea.predict_minimax(b, d, 3, "hybrid", cube_layer=L)
            ea.predict_minimax(b, d, 2, "attk", cube_layer=L)
            ea.predict_mcts(b, d, num_simulations=7, num_env_copies=3, key=M, cube_layer=L)
            ea.playout_wins(b, first_player=2, n_sims=37, key=M, cube_layer=L)
            ea.evaluate(b, "hybrid", cube_layer=L)
            ea.predict_random(b, d, key=M, step=3, cube_layer=L) if hasattr(ea, "predict_random") else None
        finally:
            monkeypatch.setattr(vec_env.torch, "zeros", real_zeros)
        torch.cuda.synchronize()
        for buf, pad in guarded:
            assert bool((buf[:G] == 0xA5).all()) and bool((buf[G + pad:] == 0xA5).all()), (S, L, M)
        assert tables_intact(), (S, L, M)
        guarded.clear()
    vec_env._TABLES.clear()
    vec_env._TABLES.update(saved_tables)


@pytest.mark.parametrize("heur", ["min_dist", "attk", "two_min_dist"])
@pytest.mark.parametrize("S,depth,n", [(5, 1, 1500), (5, 2, 1500), (5, 3, 6000), (5, 4, 400), (5, 5, 60), (7, 3, 1500), (8, 4, 200), (6, 6, 6)])
def test_integer_heuristics_on_the_table_driven_kernel(ea, heur, S, depth, n):
    """'min_dist' and 'attk' (envs/minimax_ewn.py:88-131, 180-213) are functions of each side's (level, count) like 'hybrid':
    the table-driven search runs them from their own table images; 'two_min_dist' (:133-178) from an image indexed by each side's
    sum of its two smallest distances (stateless predict and the one-thread-per-game step); against the template recursion and
    the oracle, late-game positions (one or two cubes a side) included."""
    b, d = _random_positions(S, 3, n, 4000 + depth + S, max_steps=14 if S == 5 else 24)
    fa, fv = ea.predict_minimax(b, d, depth, heur, use_tables=True)
    oa, ov, _ = po.predict_minimax(b, d, depth, heur)
    assert np.array_equal(cpu(fa), oa) and np.array_equal(bits(cpu(fv)), bits(ov))
    if depth <= 4:
        ga, gv = ea.predict_minimax(b, d, depth, heur, use_tables=False)
        assert np.array_equal(cpu(ga), oa) and np.array_equal(bits(cpu(gv)), bits(ov))


def test_generic_kernels_stay_covered(ea):
    """use_tables=False keeps the template-recursive / bitboard kernels under test for the configurations the table-driven
    kernel has taken over (shaped env, integer heuristics), and the lean kernel is run with them at its other lanes-per-game
    choices (40 000 lanes: two per game)."""
    _lockstep(ea, 800, 30, shaped=True, illegal_move_tolerance=3, reward=10.0, illegal_move_reward=-0.25, rng="mt19937", use_tables=False)
    _lockstep(ea, 600, 25, shaped=True, shaped_refresh_on_reset=True, opponent_policy="minimax", max_depth=3, rng="philox", use_tables=False)
    _lockstep(ea, 500, 20, opponent_policy="minimax", max_depth=3, heuristic="min_dist", rng="philox", use_tables=False)
    _lockstep(ea, 500, 20, opponent_policy="minimax", max_depth=4, heuristic="attk", rng="mt19937")
    _lockstep(ea, 300, 20, board_size=7, opponent_policy="minimax", max_depth=3, heuristic="min_dist", rng="philox", philox_key=8)
    _lockstep(ea, 128, 10, opponent_policy="minimax", max_depth=5, heuristic="attk", rng="philox", philox_key=9)
    _lockstep(ea, 2000, 40, opponent_policy="minimax", max_depth=3, heuristic="two_min_dist", rng="philox", philox_key=12)   # table path of the one-thread-per-game step
    _lockstep(ea, 300, 30, board_size=7, opponent_policy="minimax", max_depth=4, heuristic="two_min_dist", rng="mt19937", shaped=True)
    _lockstep(ea, 40000, 8, shaped=True, illegal_move_tolerance=2, reward=10.0, opponent_policy="minimax", max_depth=3, rng="philox",
              philox_key=10, check_terminal=False)


@pytest.mark.parametrize("S,L,depth,n", [(5, 3, 1, 64), (5, 3, 2, 48), (5, 3, 3, 24), (7, 4, 2, 16), (6, 3, 4, 3), (5, 3, 5, 2)])
def test_sim_winrate_as_a_search_leaf(ea, S, L, depth, n):
    """ExpectiMinimaxAgent(heuristic='sim_winrate') (classical_policies/minimax.py:22-23 -> envs/minimax_ewn.py:36-37, 215-238):
    every leaf is 100 random playouts whose first mover follows the reference's current_player chain.  Bit-exact against the
    oracle's literal restatement (same generator); parity with the reference itself is statistical (unseeded Python random)."""
    b, d = _random_positions(S, L, n, 600 + depth, max_steps=20)
    d = np.minimum(d, 6).astype(np.int8)
    ids = np.arange(n, dtype=np.uint32) * 31 + 5
    acts, vals = ea.predict_minimax(b, d, depth, "sim_winrate", cube_layer=L, key=0xABCDEF12345, obs_id=ids)
    oa, ov = po.predict_minimax_sim(b, d, depth, key=0xABCDEF12345, obs_id=ids, cube_layer=L)
    assert np.array_equal(cpu(acts), oa)
    assert np.array_equal(bits(cpu(vals)), bits(ov))
    v = cpu(vals)
    assert ((v >= 0.0) & (v <= 1.0)).all()          # a win rate, never the +-10 of the board heuristics
    if depth == 1:                                  # value = max over root moves of a playout win rate: multiples of 1/100
        assert np.allclose(v * 100, np.round(v * 100))
    with pytest.raises(ea.EwnError):
        ea.predict_minimax(b, d, 7, "sim_winrate", cube_layer=L)   # max_depth > 6: not built for any heuristic


def test_step_with_sim_winrate_opponent(ea):
    """the env's minimax opponent with heuristic='sim_winrate': split-phase step (agent half, search kernel, opponent half)"""
    _lockstep(ea, 96, 8, opponent_policy="minimax", max_depth=2, heuristic="sim_winrate", rng="philox", philox_key=21)
    _lockstep(ea, 40, 6, opponent_policy="minimax", max_depth=1, heuristic="sim_winrate", rng="mt19937", philox_key=22)


def test_g9b_mcts_playout_policy_chi_square_vs_reference(ea, golden):
    """The HIP playouts against the reference's (oracle/gen_golden_mcts.py: 72 positions, 5000 / 1500 reference playouts per root
    move), 20 000 HIP playouts per root move: aggregate chi-square over all cells at p > 1e-3 and no systematic bias."""
    from scipy.stats import chi2
    from test_oracle_golden import _mcts_chi2
    g = golden("g9b_mcts_large.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        nsim = 20000
        _, wins = ea.predict_mcts(boards_of(recs, S), [r["dice"] for r in recs], num_simulations=nsim, num_env_copies=1, key=99, cube_layer=L)
        wins = cpu(wins)
        c2, cells, bias = _mcts_chi2(recs, lambda i, j: int(wins[i, j]), nsim)
        assert chi2.sf(c2, cells) > 1e-3, (S, c2, cells)
        assert abs(bias) < 3.5, (S, bias)


def test_philox_streams_do_not_repeat_when_the_seed_wraps(ea):
    """The episode seed advances by seed_stride per auto-reset and is 32 bits wide (np.random.seed's range).  The Philox kind
    counts the wraps in the RNG header and keys the counter and every per-episode hash with it (without it a lane
    replays another lane's dice after 2^32 / stride episodes).  Lanes started just below 2^32 wrap within a few episodes: lock-step
    against the oracle through the wrap (single steps, the fused RandomAgent output and K-step rollouts), and the episode after
    the wrap must NOT be the episode a fresh engine plays from the same 32-bit seed."""
    N = 600
    seeds = ((1 << 32) - 1500 + np.arange(N, dtype=np.uint64) * 3).astype(np.uint32)
    kw = dict(opponent_policy="random", rng="philox", philox_key=77, autoreset=True, seed_stride=N)
    env = ea.VecEWN(N, want_random_action=True, **kw)
    orc = po.OracleVecEnv(N, opponent="random", rng="philox", philox_key=77, autoreset=True, seed_stride=N)
    env.reset(seeds=seeds)
    orc.reset(seeds=seeds)
    buf = env.random_action
    acts = orc.random_actions()
    buf.copy_(torch.from_numpy(acts))
    for t in range(45):
        res = [cpu(x) for x in env.step(buf)]
        ores = orc.step(acts)
        for a, o in zip(res, ores):
            assert np.array_equal(a, o), t
        acts = orc.random_actions()
        assert np.array_equal(cpu(buf), acts), t
    hdr = cpu(env.rng_state.view(-1)[:4 * N]).reshape(N, 4).view(np.uint32)
    wrapped = hdr[:, 3] > 0
    assert wrapped.sum() > N // 2                      # most lanes are past 2^32 by now
    traj = env.alloc_rollout(20)
    env.rollout(20, traj=traj)
    for k in range(20):
        a = orc.random_actions()
        assert np.array_equal(cpu(traj["action"][k]), a), k
        ob, od, r, te, _, _ = orc.step(a)
        assert np.array_equal(cpu(traj["board"][k]), ob) and np.array_equal(cpu(traj["dice"][k]), od), k
    # same 32-bit seed, different high word: a different dice stream
    lanes = np.nonzero(wrapped)[0][:64]
    cur_seed = cpu(env.rng_state.view(-1)[:4 * N]).reshape(N, 4).view(np.uint32)[lanes, 0]
    # word 0 of an episode's stream is philox(ctr = {0, seed, high word, 'ENV1'}): the high word changes the block
    from oracle.pyoracle import philox
    key = np.array([77, 0], np.uint32)
    differ = sum(int(not np.array_equal(philox([0, s, 0, 0x454E5631], key), philox([0, s, 1, 0x454E5631], key))) for s in cur_seed[:16])
    assert differ == 16


@pytest.mark.parametrize("S,L", [(9, 3), (9, 5), (10, 4), (11, 5)])
def test_boards_larger_than_8x8(ea, S, L):
    """9x9 .. 11x11 (the reference only asserts cube_layer < board_size - 1): 81..121 cells fit no 64-bit mask, the generic kernels
    run on a mask-free state (7-bit positions).  Lock-step against the oracle with every opponent kind, the stateless queries,
    the heuristics; 12x12 is refused."""
    _lockstep(ea, 400, 60, board_size=S, cube_layer=L, rng="mt19937")
    _lockstep(ea, 200, 30, board_size=S, cube_layer=L, rng="philox", philox_key=S, shaped=True, illegal_move_tolerance=3)
    if L >= 3:
        _lockstep(ea, 160, 16, board_size=S, cube_layer=L, opponent_policy="minimax", max_depth=2, rng="philox", philox_key=S + 1)
        _lockstep(ea, 64, 8, board_size=S, cube_layer=L, opponent_policy="minimax", max_depth=3, heuristic="two_min_dist", rng="mt19937")
        _lockstep(ea, 48, 8, board_size=S, cube_layer=L, opponent_policy="mcts", num_simulations=3, num_env_copies=2, rng="philox", philox_key=S + 2)
        b, d = _random_positions(S, L, 300, 50 + S, max_steps=40)
        d = np.minimum(d, 6).astype(np.int8)
        for depth, heur in ((1, "hybrid"), (2, "min_dist"), (3, "hybrid"), (3, "attk"), (2, "two_min_dist")):
            acts, vals = ea.predict_minimax(b, d, depth, heur, cube_layer=L)
            oa, ov, _ = po.predict_minimax(b, d, depth, heur, cube_layer=L)
            assert np.array_equal(cpu(acts), oa) and np.array_equal(bits(cpu(vals)), bits(ov)), (depth, heur)
        for h in ("hybrid", "min_dist", "two_min_dist", "attk"):
            assert np.array_equal(bits(cpu(ea.evaluate(b, h, cube_layer=L))), bits(po.evaluate(b, h, cube_layer=L))), h
        acts, wins = ea.predict_mcts(b[:40], d[:40], num_simulations=9, num_env_copies=1, key=S, cube_layer=L)
        oa, ow = po.predict_mcts(b[:40], d[:40], num_simulations=9, num_env_copies=1, key=S, cube_layer=L)
        assert np.array_equal(cpu(wins), ow) and np.array_equal(cpu(acts), oa)
        w = ea.playout_wins(b[:40], first_player=2, n_sims=17, key=3, cube_layer=L)
        assert np.array_equal(cpu(w), po.playout_wins(b[:40], 2, 17, key=3, cube_layer=L))
    for player in (1, 2):
        pb, pd = _random_positions(S, L, 200, 7 + S + player, max_steps=30)
        pd = np.minimum(pd, L * (L + 1) // 2).astype(np.int8)
        got = [cpu(t) for t in ea.legal_actions(pb, pd, player=player, cube_layer=L)]
        exp = po.legal_actions(pb, pd, player=player, cube_layer=L)
        for k, (a, o) in enumerate(zip(got, exp)):
            if k == 0:
                a = np.where(a < 0, 0, a)   # the engine pads the list with -1, the oracle with 0
                o = np.where(np.arange(6)[None, :, None] < exp[1][:, None, None], o, 0)
            assert np.array_equal(a, o), (player, k)
    with pytest.raises(ea.EwnError):
        ea.VecEWN(4, board_size=12, cube_layer=3)
