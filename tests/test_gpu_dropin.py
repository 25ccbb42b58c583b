"""The drop-in packages (`envs`, `classical_policies`, `constants`) driven the way the
reference's own scripts drive them (eval_minimax.py:16-50: reset(seed=k), predict, step),
against the golden trajectories captured from the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _replay(env, rec):
    obs, info = env.reset(seed=rec["seed"])
    assert info == {} and obs["board"].reshape(-1).tolist() == rec["board0"] and obs["dice_roll"] == rec["dice0"]
    assert obs["board"] is env.board and obs["board"].dtype == np.int16   # live alias, int16: envs/ewn.py:49,493
    for t, st in enumerate(rec["steps"]):
        obs, reward, terminated, truncated, info = env.step(np.array(st["a"]))
        ctx = (rec["seed"], t)
        assert obs["board"].reshape(-1).tolist() == st["board"], ctx
        assert obs["dice_roll"] == st["dice"], ctx
        assert float(reward).hex() == float.fromhex(st["r"]).hex(), ctx
        assert (terminated, truncated) == (st["term"], st["trunc"]), ctx
        assert info.get("message") == st["msg"], ctx


def test_env_random_opponent_trajectories(golden):
    import envs
    from constants import ClassicalPolicy
    g = golden("g3_traj_random.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.random)
        assert type(env.opponent_policy).__name__ == "RandomAgent"
        for rec in [r for r in g if (r["S"], r["L"]) == (S, L)][:24]:
            _replay(env, rec)


def test_env_minimax_opponent_trajectories(golden):
    import envs
    from constants import ClassicalPolicy
    g = golden("g6_traj_minimax.json")
    for key in sorted({(r["S"], r["L"], r["depth"], r["heuristic"]) for r in g}):
        S, L, depth, heur = key
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L, opponent_policy=ClassicalPolicy.minimax,
                                            max_depth=depth, heuristic=heur)
        for rec in [r for r in g if (r["S"], r["L"], r["depth"], r["heuristic"]) == key][:6]:
            _replay(env, rec)
    with pytest.raises(TypeError):   # max_depth is a required argument of ExpectiMinimaxAgent upstream too
        envs.EinsteinWuerfeltNichtEnv(opponent_policy=ClassicalPolicy.minimax)


def test_training_env_with_reference_quirks(golden):
    import envs
    from constants import ClassicalPolicy
    for grp in golden("g7_shaped.json"):
        env = envs.MiniMaxHeuristicEnv(board_size=5, cube_layer=3, illegal_move_tolerance=grp["tol"],
                                       opponent_policy=ClassicalPolicy.minimax, goal_reward=10., seed=123)
        assert type(env.opponent_policy).__name__ == grp["opp_class"]          # D1: RandomAgent despite the argument
        assert float(env.reward).hex() == grp["ctor_reward"]
        assert float(env.prev_score).hex() == float.fromhex(grp["ctor_prev_score"]).hex()
        for rec in grp["episodes"]:
            _replay(env, rec)
            assert env.illegal_move_tolerance == rec["tol_after"]
            assert float(env.prev_score).hex() == float.fromhex(rec["prev_score_after"]).hex()


def test_intended_behaviour_without_quirks():
    import envs
    from constants import ClassicalPolicy
    env = envs.MiniMaxHeuristicEnv(opponent_policy=ClassicalPolicy.minimax, max_depth=3, goal_reward=10., seed=5,
                                   reference_quirks=False)
    assert type(env.opponent_policy).__name__ == "ExpectiMinimaxAgent" and env.reward == 10.0
    obs, _ = env.reset(seed=5)
    done, n = False, 0
    while not done and n < 100:
        a = env.get_legal_actions(env.current_player)[0]
        obs, r, done, trunc, info = env.step(a)
        n += 1
    assert done and abs(r) == 10.0


def test_policies_predict_like_the_reference(golden):
    from classical_policies import ExpectiMinimaxAgent, MctsAgent, RandomAgent, MiniMaxPolicy
    assert MiniMaxPolicy is ExpectiMinimaxAgent
    g = [r for r in golden("g5_minimax.json") if r["S"] == 5]
    agents = {}
    for r in g[:40]:
        board = np.array(r["board"], np.int16).reshape(5, 5)
        for key, (a0, a1, v) in r["res"].items():
            d, h = key.split("/")
            ag = agents.setdefault(key, ExpectiMinimaxAgent(int(d), 3, 5, heuristic=h))
            action, state = ag.predict({"board": board, "dice_roll": r["dice"]}, deterministic=True)
            assert state is None and isinstance(action, list) and action == [a0, a1]
            val, act = ag.expectiminimax_root({"board": board, "dice_roll": r["dice"]})
            assert val.hex() == float.fromhex(v).hex()
    board = np.array(g[0]["board"], np.int16).reshape(5, 5)
    m = MctsAgent(3, 5, num_simulations=10, num_env_copies=5)
    action, state = m.predict({"board": board, "dice_roll": g[0]["dice"]})
    assert state is None and isinstance(action, np.ndarray) and action.shape == (2,)
    import envs
    env = envs.EinsteinWuerfeltNichtEnv()
    ra = RandomAgent(env)
    obs, _ = env.reset(seed=3)
    for _ in range(5):
        action, _ = ra.predict(obs)
        assert action.tolist() in env.get_legal_actions(env.current_player)


def test_env_queries(golden):
    import envs
    from constants import Player
    env = envs.MinimaxEnv()
    g = [r for r in golden("g2_legal.json") if (r["S"], r["L"]) == (5, 3)][:120]
    for r in g:
        env.board[:] = np.array(r["board"]).reshape(5, 5)
        env.set_dice_roll(r["dice"])
        assert env.check_win() == r["win"]
        if "legal" in r:
            pl = Player(r["player"])
            assert env.get_legal_actions(pl) == r["legal"]
            idx = env.find_cube_to_move(True, pl)
            assert (idx + 1 if pl == Player.TOP_LEFT else -idx) == r["cube_large"]
            cp = env.cube_pos
            for (i, j), c in np.ndenumerate(env.board):
                if c != 0:
                    k = c - 1 if c > 0 else c
                    assert tuple(cp[k]) == (i, j)
    g4 = [r for r in golden("g4_eval.json") if (r["S"], r["L"]) == (5, 3)][:60]
    for r in g4:
        env.board[:] = np.array(r["board"]).reshape(5, 5)
        for h in ("hybrid", "min_dist", "two_min_dist", "attk"):
            assert float(env.evaluate(h)).hex() == float.fromhex(r[h]).hex()
    with pytest.raises(ValueError):
        env.get_legal_actions(Player.CHANCE)
    with pytest.raises(AssertionError):
        envs.EinsteinWuerfeltNichtEnv(board_size=5, cube_layer=4)


def test_ansi_render(capsys):
    import envs
    env = envs.EinsteinWuerfeltNichtEnv(render_mode="ansi")
    env.reset(seed=0)
    env.render()
    out = capsys.readouterr().out
    assert out.startswith("dice:\n5\nboard:\n")


def test_sb3_vec_env_adapter_contract():
    """The VecEnv surface SB3's A2C drives (reset -> dict of numpy, step -> obs, rewards, dones, infos with
    terminal_observation on auto-reset); checked against the single-game drop-in env on the same seeds."""
    import envs
    import ewn_gym_amd as ea
    from ewn_gym_amd.sb3_adapter import EWNVecEnv
    N = 64
    ve = EWNVecEnv(ea.VecEWN(N, opponent_policy="random", rng="mt19937", autoreset=True, want_terminal_obs=True, seed_stride=1000))
    ve.seed(100)
    obs = ve.reset()
    assert obs["board"].shape == (N, 5, 5) and obs["board"].dtype == np.int16 and obs["dice_roll"].dtype == np.int64
    single = envs.EinsteinWuerfeltNichtEnv()
    o1, _ = single.reset(seed=103)
    assert np.array_equal(o1["board"], obs["board"][3]) and o1["dice_roll"] == obs["dice_roll"][3]
    seen_terminal = 0
    for t in range(30):
        acts = np.stack([np.zeros(N, np.int64), np.full(N, t % 3)], 1)
        before = obs
        obs, rewards, dones, infos = ve.step(acts)
        assert rewards.dtype == np.float32 and dones.dtype == bool and len(infos) == N
        for i in np.nonzero(dones)[0]:
            assert "terminal_observation" in infos[i] and infos[i]["TimeLimit.truncated"] is False
            assert infos[i]["terminal_observation"]["board"].shape == (5, 5)
            assert np.array_equal(obs["board"][i], before["board"][i] * 0 + single.reset(seed=0)[0]["board"])  # fresh episode
            seen_terminal += 1
        for i in np.nonzero(~dones)[0]:
            assert "terminal_observation" not in infos[i]
    assert seen_terminal > N
    with pytest.raises(ea.EwnError):
        EWNVecEnv(ea.VecEWN(4))


def test_make_and_undo_simulated_action(golden):
    """envs/ewn.py:377-434 through the drop-in env (the move itself is the ewn_apply_action kernel)."""
    import envs
    from constants import Player
    env = envs.MinimaxEnv()
    g = golden("g11_simulated_action.json")
    for r in g["moves"][::3]:
        before = np.array(r["board"]).reshape(5, 5)
        env.board[:] = before
        env.set_dice_roll(r["dice"])
        env.history = []
        env.make_simulated_action(Player(r["player"]), r["action"])
        assert env.board.reshape(-1).tolist() == r["after"]
        assert (env.history[-1] is not None) == r["legal"]
        env.undo_simulated_action()
        assert np.array_equal(env.board, before) and env.history == []


def test_apply_action_batched(golden):
    import ewn_gym_amd as ea
    g = golden("g11_simulated_action.json")["moves"]
    for pl in (1, 2):
        recs = [r for r in g if r["player"] == pl]
        nb, valid = ea.apply_action(np.array([r["board"] for r in recs], np.int8).reshape(-1, 5, 5), [r["dice"] for r in recs],
                                    [r["action"] for r in recs], player=pl)
        assert nb.cpu().numpy().reshape(len(recs), -1).tolist() == [r["after"] for r in recs]
        assert valid.cpu().numpy().astype(bool).tolist() == [r["legal"] for r in recs]


def test_sim_winrate_heuristic_statistics(golden):
    """MinimaxEnv.simulate / evaluate('sim_winrate') (envs/minimax_ewn.py:215-238): statistical parity, 5 sigma."""
    import ewn_gym_amd as ea
    g = golden("g11_simulated_action.json")["simulate"]
    boards = np.array([r["board"] for r in g], np.int8).reshape(-1, 5, 5)
    n = 4000
    wins = ea.playout_wins(boards, first_player=2, n_sims=n, key=5).cpu().numpy()   # current_player TOP_LEFT -> BOTTOM_RIGHT moves first
    for r, w in zip(g, wins):
        p = (r["winrate"] * r["n"] + w) / (r["n"] + n)
        sigma = max(1e-9, (p * (1 - p) * (1 / r["n"] + 1 / n)) ** 0.5)
        assert abs(r["winrate"] - w / n) <= 5 * sigma + 1e-9
    import envs
    env = envs.MinimaxEnv()
    env.board[:] = boards[0]
    v = env.evaluate("sim_winrate")
    assert 0.0 <= v <= 1.0


def test_rgb_array_render_and_ctor_side_effect():
    """render_mode='rgb_array' (envs/ewn.py:503-569) drawn with PIL: frame geometry and colours of the reference's surface;
    and the reference's ctor side effect on the global numpy stream: building a search policy re-seeds it
    to 9487 and consumes one dice draw."""
    import envs
    from classical_policies import ExpectiMinimaxAgent, MctsAgent
    env = envs.EinsteinWuerfeltNichtEnv(render_mode="rgb_array")
    env.reset(seed=3)
    frame = env.render()
    assert frame.shape == (700 + 28, 700, 3) and frame.dtype == np.uint8
    lw = 700 // 5
    assert tuple(frame[lw * 2 + lw // 2, lw * 2 + lw // 2]) == (211, 179, 104)          # cell (2, 2) is empty: board colour
    assert tuple(frame[lw // 2, lw // 2 - 45]) == (255, 255, 255)                        # TOP_LEFT cube 1 at (0, 0): white disc
    assert tuple(frame[4 * lw + lw // 2, 4 * lw + lw // 2 - 45]) == (0, 0, 0)            # BOTTOM_RIGHT cube at (4, 4): black disc
    assert tuple(frame[lw, 300]) == (0, 0, 0)                                            # a grid line
    for make in (lambda: ExpectiMinimaxAgent(max_depth=2, cube_layer=3, board_size=5), lambda: MctsAgent(cube_layer=3, board_size=5)):
        np.random.seed(1)
        make()
        got = np.random.randint(0, 1 << 30, 4).tolist()
        np.random.seed(9487)
        np.random.randint(1, 7)
        assert got == np.random.randint(0, 1 << 30, 4).tolist()


def test_mt_stream_exhaustion_is_reported():
    """MT19937-compat dice are numpy's only for the first 454 draws of an episode; past that the lane is flagged and
    VecEWN.check_rng() raises instead of silently playing wrong dice."""
    import ewn_gym_amd as ea
    env = ea.VecEWN(64, rng="mt19937", mt_window=16)
    env.reset(seeds=np.arange(64))
    env.check_rng()
    env.rng_state.view(-1)[1:4 * 64:4] = 453          # draw index of every lane: the next draws are numbers 453, 454, ...
    for t in range(2):
        env.step(env.sample_legal_actions(t))
    assert bool(env.rng_overflow().any())
    with pytest.raises(ea.EwnError, match="MT19937"):
        env.check_rng()
    ok = ea.VecEWN(64, rng="philox")
    ok.reset(seeds=np.arange(64))
    ok.step(ok.sample_legal_actions(0))
    ok.check_rng()


def test_sim_winrate_agent_through_the_drop_in_classes():
    """ExpectiMinimaxAgent(heuristic='sim_winrate') and MinimaxEnv.evaluate('sim_winrate') exist and behave: a depth-1 agent
    (max over root moves of 100-playout win rates) beats RandomAgent clearly more often than it loses."""
    import envs
    from classical_policies import ExpectiMinimaxAgent
    from constants import ClassicalPolicy
    agent = ExpectiMinimaxAgent(max_depth=1, cube_layer=3, board_size=5, heuristic="sim_winrate", seed=3)
    env = envs.EinsteinWuerfeltNichtEnv(opponent_policy=ClassicalPolicy.random)
    wins = 0
    for ep in range(24):
        obs, _ = env.reset(seed=ep)
        for _ in range(60):
            action, _ = agent.predict(obs)
            assert action in env.get_legal_actions(env.current_player)
            obs, reward, terminated, truncated, info = env.step(action)
            if terminated:
                wins += reward > 0
                break
    assert wins >= 16, wins
    m = envs.MinimaxEnv()
    v = m.evaluate("sim_winrate")
    assert 0.0 <= v <= 1.0


def test_roll_dice_is_numpys_next_draw():
    """envs/ewn.py:90-92: roll_dice() = np.random.randint(1, cube_num + 1) on the stream reset(seed) seeded -- the drop-in draws it on
    the device from the env's own MT19937-compatible stream; numpy itself (legacy seed + randint) says what the values must be"""
    import envs
    for S, L in ((5, 3), (7, 4)):
        env = envs.EinsteinWuerfeltNichtEnv(board_size=S, cube_layer=L)
        for seed in (0, 7, 9487, 2 ** 32 - 1):
            obs, _ = env.reset(seed=seed)
            np.random.seed(seed)
            exp = [int(np.random.randint(1, env.cube_num + 1)) for _ in range(9)]
            got = [obs["dice_roll"]]
            for _ in range(8):
                env.roll_dice()
                got.append(env.dice_roll)
            assert got == exp, (S, L, seed)
