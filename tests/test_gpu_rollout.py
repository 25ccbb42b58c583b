"""ewn_step_k (K env steps per launch, the agent played by the engine) against the CPU oracle, step for step:
every column of the trajectory, the carried-over state between launches, the per-lane totals -- bit-exact."""
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


SLOT_N = 66000   # two lanes per game x 66 000 games >= 131 072 lanes: the launcher picks k_rollout_slots (one root cube per loop iteration)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


def cpu(t):
    return t.detach().cpu().numpy()


def _rollout_vs_oracle(ea, N, lo, hi, K, launches, agent="random", agent_max_depth=3, autoreset=True, board_column=True, layout="columns", **kw):
    okw = dict(kw)
    opp = okw.pop("opponent_policy")
    env = ea.VecEWN(N, opponent_policy=opp, autoreset=autoreset, seed_stride=N, **okw)
    assert env.supports_rollout(agent, agent_max_depth)
    seeds = (np.arange(N, dtype=np.uint64) * 7 + 1234).astype(np.uint32)
    env.reset(seeds=seeds)
    S, L = kw.get("board_size", 5), 3
    orc = po.OracleVecEnv(hi - lo, opponent=opp, autoreset=autoreset, seed_stride=N, lane_offset=lo, **okw)
    ob, od = orc.reset(seeds=seeds[lo:hi])
    traj = env.alloc_rollout(K, board=board_column, layout=layout)
    board_column = board_column or layout == "record"
    totals = env.alloc_totals()
    frozen = np.zeros(hi - lo, bool)
    ret = np.zeros(hi - lo)
    nst = np.zeros(hi - lo, np.int64)
    nep = np.zeros(hi - lo, np.int64)
    nwin = np.zeros(hi - lo, np.int64)
    for launch in range(launches):
        env.rollout(K, agent=agent, agent_max_depth=agent_max_depth, traj=traj, totals=totals)
        tj = {k: cpu(v[:, lo:hi]) for k, v in traj.items()}
        for k in range(K):
            if agent == "random":
                acts = orc.random_actions()
            elif agent == "sample":
                acts = orc.sample_actions()
            else:
                acts = po.predict_minimax(ob, od, agent_max_depth, "hybrid", cube_layer=L)[0]
            live = ~frozen
            ctx = (kw, launch, k)
            assert np.array_equal(tj["action"][k][live], acts[live]), ctx
            ob, od, r, te, tr, info = orc.step(np.where(live[:, None], acts, 0).astype(np.int8))
            if board_column:
                assert np.array_equal(tj["board"][k], ob), ctx
            assert np.array_equal(tj["dice"][k], od), ctx
            assert np.array_equal(bits(tj["reward"][k]), bits(r)), ctx
            assert np.array_equal(tj["terminated"][k], te) and np.array_equal(tj["truncated"][k], tr), ctx
            assert np.array_equal(tj["info"][k], info), ctx
            ret += np.where(live, r, 0.0)
            nst += live
            nep += live & (te != 0)
            nwin += live & (info == 2)
            if not autoreset:
                frozen |= te != 0
        # the state written back at the end of the launch is the oracle's
        assert np.array_equal(cpu(env.board[lo:hi]), ob) and np.array_equal(cpu(env.dice[lo:hi]), od), (kw, launch)
        assert np.array_equal(cpu(env.done[lo:hi]) != 0, frozen), (kw, launch)
    assert np.array_equal(bits(cpu(totals["return_sum"][lo:hi])), bits(ret))
    assert np.array_equal(cpu(totals["n_steps"][lo:hi]), nst) and np.array_equal(cpu(totals["n_episodes"][lo:hi]), nep)
    assert np.array_equal(cpu(totals["n_wins"][lo:hi]), nwin)
    return int(nep.sum())


@pytest.mark.parametrize("N,lo", [(1500, 0), (40000, 30000), (140000, 131000)])   # two lanes per game (small and mid-size), one lane per game
def test_rollout_random_agent_depth3_opponent(ea, N, lo):
    n = _rollout_vs_oracle(ea, N, lo, lo + 384, 9, 3, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=2024)
    assert n > 384     # every lane of the slice went through auto-resets inside the launches


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_rollout_other_table_depths(ea, depth):
    _rollout_vs_oracle(ea, 700, 100, 500, 6, 2, opponent_policy="minimax", max_depth=depth, rng="philox", philox_key=depth)


@pytest.mark.parametrize("heur,depth", [("min_dist", 3), ("attk", 2), ("attk", 4)])
def test_rollout_integer_heuristic_opponents(ea, heur, depth):
    _rollout_vs_oracle(ea, 700, 100, 500, 6, 2, opponent_policy="minimax", max_depth=depth, heuristic=heur, rng="philox", philox_key=depth)


@pytest.mark.parametrize("N,lo,kw", [
    (700, 100, dict(max_depth=3)), (700, 0, dict(max_depth=2, board_size=7)), (SLOT_N, 30000, dict(max_depth=3)), (SLOT_N, 100, dict(max_depth=4, board_size=6)),
    (140000, 139000, dict(max_depth=3)), (300, 0, dict(max_depth=5)), (900, 0, dict(max_depth=3, rng="mt19937", autoreset=False)),
], ids=lambda v: str(v) if not isinstance(v, dict) else "-".join("%s=%s" % kv for kv in sorted(v.items())))
def test_rollout_two_min_dist_opponent(ea, N, lo, kw):
    """'two_min_dist' (envs/minimax_ewn.py:133-178) inside the K-step kernels: its table image has its own instances of the lock-step
    and of the slot-task kernel (max_depth 1-4; max_depth 5 / 6 keep the reference's loops, lock-step, one lane per game)"""
    kw = dict(kw)
    rng = kw.pop("rng", "philox")
    autoreset = kw.pop("autoreset", True)
    n = 96 if kw["max_depth"] >= 5 else 256
    _rollout_vs_oracle(ea, N, lo, lo + n, 4 if kw["max_depth"] >= 5 else 7, 2, autoreset=autoreset, opponent_policy="minimax", heuristic="two_min_dist", rng=rng,
                       philox_key=500 + kw["max_depth"], layout="record" if N == SLOT_N else "columns", **kw)


@pytest.mark.parametrize("N,lo,n,K,kw", [
    (600, 100, 300, 5, dict(num_simulations=10, num_env_copies=5)),                      # MCTS(10 x 5): the reference's default, eval_pairs.py:16
    (300, 0, 200, 4, dict(num_simulations=3, num_env_copies=4, board_size=7)),
    (700, 500, 200, 4, dict(num_simulations=2, num_env_copies=3, agent="sample")),
    (260, 0, 260, 6, dict(num_simulations=4, num_env_copies=2, rng="mt19937", autoreset=False)),
    (300, 40, 128, 3, dict(num_simulations=40, num_env_copies=10, layout="record")),     # 400 playouts per root move: groups of 64 lanes
    (520, 256, 264, 4, dict(num_simulations=5, num_env_copies=5, board_size=6, layout="record")),
], ids=lambda v: str(v) if not isinstance(v, dict) else "-".join("%s=%s" % kv for kv in sorted(v.items())))
def test_rollout_mcts_opponent(ea, N, lo, n, K, kw):
    """The flat Monte-Carlo opponent (classical_policies/mcts.py:21-106) inside ewn_step_k: agent half, playouts and opponent half as
    the phases of one loop body (k_rollout_mcts), step for step against the oracle (which mirrors the playout generator)"""
    kw = dict(kw)
    agent = kw.pop("agent", "random")
    rng = kw.pop("rng", "philox")
    autoreset = kw.pop("autoreset", True)
    layout = kw.pop("layout", "columns")
    _rollout_vs_oracle(ea, N, lo, lo + n, K, 2, agent=agent, autoreset=autoreset, opponent_policy="mcts", rng=rng, philox_key=4242, layout=layout, **kw)


def test_rollout_depth5_opponent(ea):
    _rollout_vs_oracle(ea, 200, 0, 96, 5, 2, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=55)


@pytest.mark.parametrize("S", [6, 7, 8])
def test_rollout_other_board_sizes(ea, S):
    _rollout_vs_oracle(ea, 1000, 300, 600, 8, 2, board_size=S, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=S)
    _rollout_vs_oracle(ea, 40000, 20000, 20128, 5, 2, board_size=S, opponent_policy="random", rng="philox", philox_key=S + 1, board_column=False)


@pytest.mark.parametrize("kw", [
    dict(max_depth=1), dict(max_depth=2), dict(max_depth=4), dict(max_depth=3, heuristic="min_dist"), dict(max_depth=4, heuristic="attk"),
    dict(max_depth=5), dict(max_depth=6, heuristic="attk"), dict(max_depth=3, board_size=6), dict(max_depth=3, board_size=7),
    dict(max_depth=4, board_size=8), dict(max_depth=5, board_size=7),
], ids=lambda kw: "-".join("%s=%s" % kv for kv in sorted(kw.items())))
def test_rollout_slot_task_kernel_every_search(ea, kw):
    """The slot-task rollout kernel (games of a wave drift apart by a few env steps inside a launch) against the oracle: every
    table-driven depth class, heuristic image and board size, state carried over three launches, trajectory rows included."""
    lo = 40000
    _rollout_vs_oracle(ea, SLOT_N, lo, lo + (96 if kw["max_depth"] >= 5 else 256), 7, 3, opponent_policy="minimax", rng="philox",
                       philox_key=1000 + kw["max_depth"], **kw)


@pytest.mark.parametrize("N,lo,kw", [(1200, 0, dict(max_depth=3)), (SLOT_N, 30000, dict(max_depth=3)), (140000, 139000, dict(max_depth=2)),
                                     (SLOT_N, 100, dict(max_depth=5)), (3000, 1000, dict(opponent="random"))],
                         ids=lambda v: str(v) if not isinstance(v, dict) else "-".join("%s=%s" % kv for kv in sorted(v.items())))
def test_rollout_action_space_sample_agent(ea, N, lo, kw):
    """EWN_AGENT_SAMPLE: env.action_space.sample() as the agent -- a draw that leaves the board is an illegal move, which ends the
    episode (envs/ewn.py:443-449) and, with auto-reset, starts the next (about one lane in ten per step): both rollout kernels."""
    kw = dict(kw)
    opp = kw.pop("opponent", "minimax")
    n = _rollout_vs_oracle(ea, N, lo, lo + 200, 8, 3, agent="sample", opponent_policy=opp, rng="philox", philox_key=314, **kw)
    assert n > 100         # episodes ended (and restarted) inside the launches


@pytest.mark.parametrize("N,lo,kw", [
    (SLOT_N, 30000, dict(max_depth=3)), (SLOT_N, 100, dict(max_depth=3, board_size=6)), (SLOT_N, 65000, dict(max_depth=4, board_size=7)),
    (SLOT_N, 2000, dict(max_depth=5, board_size=8)), (140000, 139000, dict(max_depth=3)), (140000, 64, dict(max_depth=2, board_size=7)),
    (1500, 0, dict(max_depth=3)), (40000, 30000, dict(max_depth=3, board_size=6)), (900, 100, dict(max_depth=5, board_size=8)),
    (3000, 1000, dict(opponent="random")), (3000, 0, dict(opponent="random", board_size=7)),
], ids=lambda v: str(v) if not isinstance(v, dict) else "-".join("%s=%s" % kv for kv in sorted(v.items())))
def test_rollout_record_layout(ea, N, lo, kw):
    """ewn_rollout_out.record: one 16-byte aligned record per lane-step (board | dice | action | flags), the board bytes kept
    current in LDS move by move (slot-task kernel) or re-encoded per step (lock-step kernel): every byte against the oracle,
    both kernels, one and two lanes per game, 5x5 .. 8x8, auto-resets inside the launches."""
    kw = dict(kw)
    opp = kw.pop("opponent", "minimax")
    n = _rollout_vs_oracle(ea, N, lo, lo + (96 if kw.get("max_depth", 3) >= 5 else 200), 9, 3, layout="record", opponent_policy=opp,
                           rng="philox", philox_key=99, **kw)
    assert n > 100


def test_rollout_record_padding_is_zero_and_columns_agree(ea):
    """record and separate columns asked for together describe the same trajectory; padding bytes stay zero; frozen lanes (no
    auto-reset) and the action_space.sample() agent (illegal moves) included"""
    from ewn_gym_amd.vec_env import _ptr  # noqa: F401
    for N, S, agent, autoreset in ((SLOT_N, 5, "sample", True), (2000, 7, "random", False), (SLOT_N, 6, "random", False)):
        env = ea.VecEWN(N, board_size=S, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=5, autoreset=autoreset, seed_stride=N)
        env.reset(seeds=(np.arange(N, dtype=np.uint64) + 11).astype(np.uint32))
        K = 12
        cols = env.alloc_rollout(K)
        rec = env.alloc_rollout(K, layout="record")
        both = dict(cols)
        both["record"] = rec["record"]
        # hand both the columns and the record to one call (rollout() drops the columns when it sees a record: call the ABI directly)
        import ctypes as C
        from ewn_gym_amd._lib import AGENT, EwnRolloutOut
        from ewn_gym_amd.vec_env import _ptr, _stream
        out = EwnRolloutOut(_ptr(cols["board"]), _ptr(cols["dice"]), _ptr(cols["action"]), _ptr(cols["reward"]), _ptr(cols["terminated"]),
                            _ptr(cols["truncated"]), _ptr(cols["info"]), None, None, None, None, _ptr(rec["record"]))
        assert env.lib.ewn_step_k(C.byref(env.cfg), C.byref(env._st), K, AGENT[agent], 3, C.byref(out), _stream()) == 0
        for key in ("board", "dice", "action", "terminated", "truncated", "info"):
            assert torch.equal(rec[key], cols[key]), (N, S, key)
        assert int(rec["record"][:, :, S * S + 6:].max()) == 0
        assert int(cols["terminated"].sum()) > 0


def test_rollout_slot_task_kernel_short_launches(ea):
    """K = 1, 2, 3: a launch ends after every game has played exactly K steps, however many of them needed a second iteration"""
    for K in (1, 2, 3):
        _rollout_vs_oracle(ea, SLOT_N + 37, 65900, 66037, K, 6, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=70 + K)


def test_rollout_slot_task_kernel_frozen_lanes_and_numpy_dice(ea):
    """... without auto-reset (lanes freeze as their episode ends, and go on writing their rows) on the MT19937-compat dice, with
    and without the board column; one lane per game at 140 000 lanes."""
    _rollout_vs_oracle(ea, SLOT_N, 100, 400, 9, 4, autoreset=False, opponent_policy="minimax", max_depth=3, rng="mt19937")
    _rollout_vs_oracle(ea, SLOT_N, 65000, 65300, 9, 2, autoreset=False, opponent_policy="minimax", max_depth=5, rng="mt19937", board_column=False)
    _rollout_vs_oracle(ea, 140000, 139700, 140000, 11, 3, autoreset=False, opponent_policy="minimax", max_depth=3, rng="mt19937")
    _rollout_vs_oracle(ea, 140000, 0, 200, 6, 2, opponent_policy="minimax", max_depth=5, rng="philox", philox_key=5, board_column=False)


@pytest.mark.parametrize("kw", [
    dict(board_size=7, cube_layer=4, opponent_policy="minimax", max_depth=3), dict(board_size=7, cube_layer=5, opponent_policy="minimax", max_depth=2, heuristic="two_min_dist"),
    dict(board_size=9, cube_layer=3, opponent_policy="minimax", max_depth=3), dict(board_size=11, cube_layer=3, opponent_policy="random"),
    dict(board_size=6, cube_layer=4, opponent_policy="random"), dict(board_size=7, cube_layer=4, opponent_policy="minimax", max_depth=1, heuristic="attk", agent="sample"),
], ids=lambda kw: "-".join("%s=%s" % kv for kv in sorted(kw.items())))
def test_rollout_geometries_without_a_table_image(ea, kw):
    """cube_layer 4 / 5 and boards of 9x9 .. 11x11 have no table image: ewn_step_k plays them on the generic K-step kernel
    (one thread per game, the rules and the recursion of the generic step kernel) -- every column, both layouts, the carried-over
    state and the totals against the oracle; auto-reset, frozen lanes with numpy-compatible dice"""
    kw = dict(kw)
    agent = kw.pop("agent", "random")
    N = 2500
    _rollout_vs_oracle(ea, N, 1200, 1500, 7, 3, agent=agent, rng="philox", philox_key=77, **kw)
    _rollout_vs_oracle(ea, N, 0, 300, 7, 2, agent=agent, layout="record", rng="philox", philox_key=78, **kw)
    _rollout_vs_oracle(ea, N, 2200, 2500, 9, 3, agent=agent, autoreset=False, rng="mt19937", **kw)


def test_bound_rollout_call_is_the_same_launch(ea):
    """VecEWN.bind_rollout: the pre-marshalled call leaves the same state and trajectory as rollout()"""
    outs = []
    for bound in (False, True):
        env = ea.VecEWN(SLOT_N, opponent_policy="minimax", max_depth=3, rng="philox", philox_key=8, autoreset=True, seed_stride=SLOT_N)
        env.reset(seeds=(np.arange(SLOT_N, dtype=np.uint64) + 5).astype(np.uint32))
        traj = env.alloc_rollout(6, layout="record")
        call = env.bind_rollout(6, traj=traj) if bound else (lambda: env.rollout(6, traj=traj))
        call(); call()
        outs.append((env.board.clone(), env.dice.clone(), traj["record"].clone(), traj["reward"].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_config2_random_opponent_at_its_own_shape(ea):
    """BASELINE config 2 at its own shape (5x5, 65 536 lanes, RandomAgent opponent): ewn_step_k against the oracle on two slices,
    both trajectory layouts (the launcher's kernel choice depends on the lane count)"""
    _rollout_vs_oracle(ea, 65536, 0, 300, 10, 3, opponent_policy="random", rng="philox", philox_key=2024)
    _rollout_vs_oracle(ea, 65536, 65236, 65536, 10, 3, opponent_policy="random", rng="philox", philox_key=2024, layout="record")


def test_rollout_random_opponent_and_mt19937_without_autoreset(ea):
    _rollout_vs_oracle(ea, 3000, 1000, 1600, 10, 3, opponent_policy="random", rng="philox", philox_key=3)
    # numpy-compatible dice, one episode per lane (the evaluation scripts' shape): lanes freeze as they finish
    _rollout_vs_oracle(ea, 1024, 0, 512, 8, 4, autoreset=False, opponent_policy="random", rng="mt19937")
    _rollout_vs_oracle(ea, 1024, 0, 512, 8, 3, autoreset=False, opponent_policy="minimax", max_depth=3, rng="mt19937")
    _rollout_vs_oracle(ea, 256, 0, 128, 6, 3, autoreset=False, opponent_policy="minimax", max_depth=5, rng="mt19937")


@pytest.mark.parametrize("agent_depth,opp,opp_depth,rng", [(3, "minimax", 3, "mt19937"), (2, "random", 3, "philox"), (4, "minimax", 3, "philox"),
                                                          (3, "minimax", 4, "mt19937"), (5, "minimax", 3, "mt19937"), (3, "minimax", 5, "philox"),
                                                          (5, "random", 3, "mt19937")])
def test_rollout_minimax_agent(ea, agent_depth, opp, opp_depth, rng):
    """ExpectiMinimaxAgent as the agent (eval_minimax.py's loop on the device): the agent's search runs on the flipped position;
    depth classes 1-4 and 5-6 on either side, same or different table images."""
    n = 160 if max(agent_depth, opp_depth) >= 5 else 512
    _rollout_vs_oracle(ea, n, 0, min(n, 256), 6, 3, agent="minimax", agent_max_depth=agent_depth, autoreset=False,
                       opponent_policy=opp, max_depth=opp_depth, rng=rng, philox_key=agent_depth * 10 + opp_depth)


def test_rollout_minimax_agent_with_autoreset_on_7x7(ea):
    _rollout_vs_oracle(ea, 300, 0, 200, 7, 2, agent="minimax", agent_max_depth=3, board_size=7, opponent_policy="minimax", max_depth=2,
                       rng="philox", philox_key=77)


def test_rollout_equals_the_step_loop_at_full_size(ea):
    """65 536 lanes: K steps in one launch == K ewn_step launches with the fused RandomAgent action fed back (state, RNG headers
    and every step's outputs); the first action of the loop is the oracle-checked hash pick taken from a 1-step rollout."""
    N, K = 65536, 12
    kw = dict(opponent_policy="minimax", max_depth=3, rng="philox", philox_key=2024, autoreset=True, seed_stride=N)
    seeds = (np.arange(N, dtype=np.uint64) + 9487).astype(np.uint32)
    a = ea.VecEWN(N, **kw)
    b = ea.VecEWN(N, want_random_action=True, **kw)
    a.reset(seeds=seeds)
    b.reset(seeds=seeds)
    traj = a.alloc_rollout(K)
    a.rollout(K, traj=traj)
    acts = traj["action"][0].clone()          # what RandomAgent played at step 0
    buf = b.random_action
    buf.copy_(acts)
    for k in range(K):
        assert torch.equal(traj["action"][k], buf)
        bo, di, r, te, tr, info = b.step(buf)
        assert torch.equal(traj["board"][k], bo) and torch.equal(traj["dice"][k], di), k
        assert torch.equal(traj["reward"][k], r) and torch.equal(traj["terminated"][k], te) and torch.equal(traj["info"][k], info), k
    assert torch.equal(a.board, b.board) and torch.equal(a.dice, b.dice) and torch.equal(a.rng_state, b.rng_state)
    # shard invariance: two half-size engines with global lane ids give the same trajectory
    h0 = ea.VecEWN(N // 2, lane_offset=0, **kw)
    h1 = ea.VecEWN(N // 2, lane_offset=N // 2, **kw)
    h0.reset(seeds=seeds[:N // 2])
    h1.reset(seeds=seeds[N // 2:])
    t0, t1 = h0.alloc_rollout(K), h1.alloc_rollout(K)
    h0.rollout(K, traj=t0)
    h1.rollout(K, traj=t1)
    for key in traj:
        assert torch.equal(traj[key], torch.cat([t0[key], t1[key]], 1)), key


def test_tournament_through_rollouts_matches_the_reference_goldens(golden):
    """eval_minimax.py:16-50 with the whole loop on the device: per-episode scores and lengths identical to the reference's"""
    from ewn_gym_amd.tournament import evaluate
    for rec in golden("g10_eval_loop.json"):
        opp = {"kind": rec["opp"]}
        if rec["opp_depth"]:
            opp["max_depth"] = rec["opp_depth"]
        r = evaluate({"kind": "minimax", "max_depth": rec["agent_depth"]}, opp, num=len(rec["scores"]), rng="mt19937")
        assert r["engine"] == "ewn_step_k"
        assert r["scores"].cpu().tolist() == rec["scores"]
        assert r["lengths"].cpu().tolist() == rec["lengths"]
        slow = evaluate({"kind": "minimax", "max_depth": rec["agent_depth"]}, opp, num=len(rec["scores"]), rng="mt19937", use_rollout=False)
        assert slow["engine"] == "ewn_step" and torch.equal(slow["scores"], r["scores"]) and torch.equal(slow["lengths"], r["lengths"])
