"""The CPU oracle (oracle/ewn_oracle.c) against golden vectors captured from the
unmodified reference (oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po

MSG = {None: 0, "Invalid move for player! End the game.": 1, "You won!": 2,
       "Invalid move for opponent! End the game.": 3, "You lost!": 4}


def msg_code(m):
    if m is not None and m.startswith("Invalid move for player! Tolerance left"):
        return 5
    return MSG[m]


def boards_of(recs, S):
    return np.array([r["board"] for r in recs], np.int8).reshape(-1, S, S)


def test_g1_initial_boards_and_first_dice(golden):
    g = golden("g1_initial.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        env = po.OracleVecEnv(len(recs), board_size=S, cube_layer=L, rng="mt19937")
        b, d = env.reset(seeds=[r["seed"] for r in recs])
        assert np.array_equal(b, boards_of(recs, S))
        assert d.tolist() == [r["dice"] for r in recs]


def test_g2_legal_actions_cube_selection_win(golden):
    g = golden("g2_legal.json")
    for (S, L, pl) in sorted({(r["S"], r["L"], r["player"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"], r["player"]) == (S, L, pl)]
        acts, n, cs, cl, win = po.legal_actions(boards_of(recs, S), [r["dice"] for r in recs], player=pl, cube_layer=L)
        for i, r in enumerate(recs):
            assert bool(win[i]) == r["win"]
            if "legal" in r:
                assert acts[i, :n[i]].tolist() == r["legal"], (S, L, pl, i)
                assert (cs[i], cl[i]) == (r["cube_small"], r["cube_large"])


@pytest.mark.parametrize("h", ["hybrid", "min_dist", "two_min_dist", "attk"])
def test_g4_heuristics_bit_exact(golden, h):
    g = golden("g4_eval.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        out = po.evaluate(boards_of(recs, S), h, cube_layer=L)
        exp = np.array([float.fromhex(r[h]) for r in recs])
        assert np.array_equal(out.view(np.uint64), exp.view(np.uint64))


def test_g5_expectiminimax_action_and_root_value(golden):
    g = golden("g5_minimax.json")
    n = 0
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        for key in sorted({k for r in recs for k in r["res"]}):
            sub = [r for r in recs if key in r["res"]]
            d, h = key.split("/")
            acts, vals, _ = po.predict_minimax(boards_of(sub, S), [r["dice"] for r in sub], int(d), h, cube_layer=L)
            for i, r in enumerate(sub):
                a0, a1, v = r["res"][key]
                assert acts[i].tolist() == [a0, a1], (S, L, key, i)
                assert vals[i].hex() == float.fromhex(v).hex(), (S, L, key, i)
                n += 1
    assert n > 3000


def _check_traj(rec, opp, **kw):
    S, L = rec.get("S", 5), rec.get("L", 3)
    env = po.OracleVecEnv(1, board_size=S, cube_layer=L, opponent=opp, rng="mt19937", **kw)
    b, d = env.reset(seeds=[rec["seed"]])
    assert b.reshape(-1).tolist() == rec["board0"] and int(d[0]) == rec["dice0"]
    for t, st in enumerate(rec["steps"]):
        b, d, r, te, tr, info = env.step([st["a"]])
        ctx = (rec["seed"], rec.get("rule"), t)
        assert b.reshape(-1).tolist() == st["board"], ctx
        assert int(d[0]) == st["dice"], ctx
        assert float(r[0]).hex() == float.fromhex(st["r"]).hex(), ctx
        assert (bool(te[0]), bool(tr[0])) == (st["term"], st["trunc"]), ctx
        assert int(info[0]) == msg_code(st["msg"]), ctx


def test_g3_trajectories_random_opponent(golden):
    g = golden("g3_traj_random.json")
    for rec in g:
        _check_traj(rec, "random")
    assert sum(len(r["steps"]) for r in g) > 2000


def test_g6_trajectories_minimax_opponent(golden):
    for rec in golden("g6_traj_minimax.json"):
        _check_traj(rec, "minimax", max_depth=rec["depth"], heuristic=rec["heuristic"])


def test_g7_shaped_env_with_reference_quirks(golden):
    for grp in golden("g7_shaped.json"):
        # App. D1: MinimaxEnv drops opponent_policy / reward / seed -> RandomAgent, reward 1.0
        assert grp["opp_class"] == "RandomAgent" and float.fromhex(grp["ctor_reward"]) == 1.0
        env = po.OracleVecEnv(1, opponent="random", rng="mt19937", shaped=True, illegal_move_tolerance=grp["tol"],
                              reward=1.0, illegal_move_reward=-1.0)
        ps, tol, _ = env.aux()
        assert ps[0].hex() == float.fromhex(grp["ctor_prev_score"]).hex()
        for rec in grp["episodes"]:
            b, d = env.reset(seeds=[rec["seed"]])
            assert b.reshape(-1).tolist() == rec["board0"] and int(d[0]) == rec["dice0"]
            for t, st in enumerate(rec["steps"]):
                b, d, r, te, tr, info = env.step([st["a"]])
                ctx = (grp["tol"], rec["seed"], t)
                assert b.reshape(-1).tolist() == st["board"], ctx
                assert int(d[0]) == st["dice"], ctx
                assert float(r[0]).hex() == float.fromhex(st["r"]).hex(), ctx
                assert (bool(te[0]), bool(tr[0])) == (st["term"], st["trunc"]), ctx
                assert int(info[0]) == msg_code(st["msg"]), ctx
            ps, tol, _ = env.aux()
            assert int(tol[0]) == rec["tol_after"]
            assert ps[0].hex() == float.fromhex(rec["prev_score_after"]).hex()


def test_g8_numpy_legacy_rng_stream(golden):
    g = golden("g8_rng.json")
    for raw in g["raw"]:
        assert po.mt_outputs(raw["seed"], len(raw["u32"])).tolist() == raw["u32"]
        lo = [c[0] for c in raw["randint"]]
        hi = [c[1] for c in raw["randint"]]
        assert po.np_randint_seq(raw["seed"], lo, hi).tolist() == [c[2] for c in raw["randint"]]
    for ep in g["episodes"]:
        lo = [c[0] for c in ep["calls"]]
        hi = [c[1] for c in ep["calls"]]
        assert po.np_randint_seq(ep["seed"], lo, hi).tolist() == [c[2] for c in ep["calls"]]


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10
    assert po.philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert po.philox([0xffffffff] * 4, [0xffffffff] * 2).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert po.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_g9_flat_monte_carlo_statistics(golden):
    """The reference's rollouts use a never-seeded Python `random`: statistical parity only.
    Each root move's win rate must agree within 5 sigma of the two-sample binomial error."""
    g = golden("g9_mcts.json")
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        nsim = 400
        acts, wins = po.predict_mcts(boards_of(recs, S), [r["dice"] for r in recs], num_simulations=nsim,
                                     num_env_copies=1, key=77, cube_layer=L)
        for i, r in enumerate(recs):
            for j, w in enumerate(r["wins"]):
                p_ref, p_or = w / r["n"], wins[i, j] / nsim
                p = (w + wins[i, j]) / (r["n"] + nsim)
                sigma = max(1e-9, (p * (1 - p) * (1 / r["n"] + 1 / nsim)) ** 0.5)
                assert abs(p_ref - p_or) <= 5 * sigma + 1e-9, (S, i, j, p_ref, p_or)
            assert (wins[i, len(r["wins"]):] == -1).all()


def test_g12_maximum_board_size(golden):
    """8x8, the largest board this build supports: heuristics, depth 1-3 search results and full trajectories produced by the
    reference (oracle/gen_golden_maxboard.py)."""
    g = golden("g12_maxboard.json")
    S, L = g["S"], g["L"]
    boards = np.array([r["board"] for r in g["eval"]], np.int8).reshape(-1, S, S)
    for h in ("hybrid", "min_dist", "two_min_dist", "attk"):
        vals = po.evaluate(boards, h, cube_layer=L)
        assert [float(v).hex() for v in vals] == [float.fromhex(r[h]).hex() for r in g["eval"]], h
    recs = g["minimax"]
    n = 0
    for key in sorted({k for r in recs for k in r["res"]}):
        d, h = key.split("/")
        acts, vals, _ = po.predict_minimax(np.array([r["board"] for r in recs], np.int8).reshape(-1, S, S), [r["dice"] for r in recs],
                                           int(d), h, cube_layer=L)
        for i, r in enumerate(recs):
            a0, a1, v = r["res"][key]
            assert acts[i].tolist() == [a0, a1] and vals[i].hex() == float.fromhex(v).hex(), (key, i)
            n += 1
    assert n >= 400
    for rec in g["traj"]:
        if rec["opp"] == "random":
            _check_traj(rec, "random")
        else:
            _check_traj(rec, "minimax", max_depth=rec["depth"], heuristic=rec["heuristic"])


def test_g13_cube_layers_4_and_5(golden):
    """10 and 15 cubes a side (two-word position state in the HIP code; the search's dice loop still runs 1..6 as upstream):
    search results and minimax-opponent trajectories produced by the reference (oracle/gen_golden_layers.py)."""
    n = 0
    for grp in golden("g13_layers.json"):
        S, L, recs = grp["S"], grp["L"], grp["minimax"]
        for key in sorted({k for r in recs for k in r["res"]}):
            d, h = key.split("/")
            acts, vals, _ = po.predict_minimax(np.array([r["board"] for r in recs], np.int8).reshape(-1, S, S), [r["dice"] for r in recs],
                                               int(d), h, cube_layer=L)
            for i, r in enumerate(recs):
                a0, a1, v = r["res"][key]
                assert acts[i].tolist() == [a0, a1] and vals[i].hex() == float.fromhex(v).hex(), (S, L, key, i)
                n += 1
        for rec in grp["traj"]:
            _check_traj(rec, "minimax", max_depth=rec["depth"], heuristic=rec["heuristic"])
    assert n >= 300


def _mcts_chi2(recs, wins_of, nsim):
    """Aggregate test of homogeneity over all (position, root move) cells: z = (p_ref - p_ours) / sigma_pooled per cell,
    chi2 = sum z^2 with one degree of freedom per cell; also the mean z (a systematic bias of the playout policy moves every
    cell the same way).  Cells whose pooled rate is 0 or 1 carry no information and are left out."""
    z = []
    for i, r in enumerate(recs):
        for j, w in enumerate(r["wins"]):
            p = (w + wins_of(i, j)) / (r["n"] + nsim)
            if p <= 0.0 or p >= 1.0:
                assert w / r["n"] == wins_of(i, j) / nsim
                continue
            z.append((w / r["n"] - wins_of(i, j) / nsim) / (p * (1 - p) * (1 / r["n"] + 1 / nsim)) ** 0.5)
    z = np.array(z)
    return float((z ** 2).sum()), len(z), float(z.mean() * len(z) ** 0.5)


def test_g9b_mcts_playout_policy_chi_square_vs_reference(golden):
    """72 positions x ~3.3 root moves x 5000 (5x5) / 1500 (7x7) playouts of the reference's MctsAgent.simulate
    (classical_policies/mcts.py:21-45, oracle/gen_golden_mcts.py) against the same number of playouts of the mirrored
    generator: aggregate chi-square at p > 1e-3 and no systematic bias (|sum z / sqrt(cells)| < 3.5).  A playout policy that is
    off by one percent in win probability fails this (z per cell ~ 1, chi2 ~ 2 x cells); the per-cell 5-sigma check of G9 did not."""
    from scipy.stats import chi2
    g = golden("g9b_mcts_large.json")
    assert len(g) >= 60 and sum(len(r["wins"]) for r in g) >= 180
    for (S, L) in sorted({(r["S"], r["L"]) for r in g}):
        recs = [r for r in g if (r["S"], r["L"]) == (S, L)]
        nsim = recs[0]["n"]
        acts, wins = po.predict_mcts(boards_of(recs, S), [r["dice"] for r in recs], num_simulations=nsim, num_env_copies=1, key=4242, cube_layer=L)
        for i, r in enumerate(recs):
            assert (wins[i, :len(r["wins"])] >= 0).all() and (wins[i, len(r["wins"]):] == -1).all()
        c2, cells, bias = _mcts_chi2(recs, lambda i, j: int(wins[i, j]), nsim)
        assert chi2.sf(c2, cells) > 1e-3, (S, c2, cells)
        assert abs(bias) < 3.5, (S, bias)


def test_playout_generator_dice_and_move_index_are_jointly_uniform():
    """One 32-bit draw per ply yields both the dice (top of a 24-bit product) and the move index (its low 24 bits times the number
    of legal moves): they come from ONE word, so their JOINT distribution is what has to be uniform, for every list length
    n = 1..6, within a stream and across the streams of consecutive playouts; plus no serial correlation of successive dice."""
    from scipy.stats import chi2
    ds, fs = [], []
    for x in range(400):                      # 400 playout streams of one observation, as the kernels seed them
        d, f = po.prng_draws(12345, x, 3000, key=0xC0FFEE)
        ds.append(d)
        fs.append(f)
    d, f = np.concatenate(ds).astype(np.int64), np.concatenate(fs).astype(np.int64)
    N = len(d)
    assert d.min() == 0 and d.max() == 5 and f.max() < (1 << 24)
    for n in range(1, 7):
        pick = (f * n) >> 24
        assert pick.min() == 0 and pick.max() == n - 1
        counts = np.bincount(d * n + pick, minlength=6 * n).astype(float)
        e = N / (6 * n)
        assert chi2.sf(((counts - e) ** 2 / e).sum(), 6 * n - 1) > 1e-4, n
    # successive plies of a stream: (dice_t, dice_t+1) uniform on 36 cells; first plies across streams uniform as well
    for arr in ds[:50]:
        a = arr.astype(np.int64)
        c = np.bincount(a[:-1] * 6 + a[1:], minlength=36).astype(float)
        assert chi2.sf(((c - (len(a) - 1) / 36) ** 2 / ((len(a) - 1) / 36)).sum(), 35) > 1e-5
    first = np.array([po.prng_draws(777, x, 1, key=5)[0][0] for x in range(6000)], dtype=np.int64)
    c = np.bincount(first, minlength=6).astype(float)
    assert chi2.sf(((c - 1000) ** 2 / 1000).sum(), 5) > 1e-4
