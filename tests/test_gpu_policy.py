"""ewn_step_k_policy: K env steps per launch with the actor-critic of train.py:35-63 as the agent (the rollout collector of the
trainer as one kernel).  The network's outputs against a plain fp32 torch forward of the same parameters on the same
observations; the sampling against the recorded noise and statistically; every transition against the CPU oracle."""
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


def make_model(S, seed, head_gain=3.0):
    from ewn_gym_amd.a2c import ActorCritic
    torch.manual_seed(seed)
    m = ActorCritic(S, 6).cuda()
    with torch.no_grad():          # SB3's 0.01-gain action head gives near-uniform policies: spread the logits so that sampling is tested
        m.action_net.weight.mul_(head_gain / 0.01)
        m.action_net.bias.uniform_(-0.5, 0.5)
        m.value_net.bias.fill_(0.25)
        for seq in (m.pi, m.vf):
            for lin in seq:
                if hasattr(lin, "bias"):
                    lin.bias.uniform_(-0.3, 0.3)
    return m


def gumbel_actions(logits, u):
    g = -torch.log(-torch.log(u))
    z = logits + g
    return torch.stack([z[..., :2].argmax(-1), z[..., 2:].argmax(-1)], -1).to(torch.int8)


def _policy_vs_torch_and_oracle(ea, N, lo, hi, K, launches, S=5, layout="record", want_value=True, deterministic=False, autoreset=True, **kw):
    okw = dict(kw)
    opp = okw.pop("opponent_policy")
    env = ea.VecEWN(N, board_size=S, opponent_policy=opp, autoreset=autoreset, seed_stride=N, rng="philox", **okw)
    assert env.supports_policy_rollout()
    seeds = (np.arange(N, dtype=np.uint64) * 5 + 77).astype(np.uint32)
    env.reset(seeds=seeds)
    orc = po.OracleVecEnv(hi - lo, board_size=S, opponent=opp, autoreset=autoreset, seed_stride=N, lane_offset=lo, rng="philox", **okw)
    ob, od = orc.reset(seeds=seeds[lo:hi])
    model = make_model(S, 5)
    params = model.flat_parameters()
    assert params.numel() == env.policy_param_count()
    rec = layout == "record"
    traj = env.alloc_rollout(K, layout=layout, initial_obs=rec)
    totals = env.alloc_totals()
    logits = torch.zeros((K, N, 5), dtype=torch.float32, device="cuda")
    value = torch.zeros((K, N), dtype=torch.float32, device="cuda") if want_value else None
    noise = torch.zeros((K, N, 5), dtype=torch.float32, device="cuda")
    frozen = np.zeros(hi - lo, bool)
    nep, flips, total = 0, 0, 0
    counts = np.zeros(5)
    probs = np.zeros(5)
    for launch in range(launches):
        b0, d0 = env.board.clone(), env.dice.clone()
        env.rollout_policy(K, params, traj=traj, totals=totals, deterministic=deterministic, noise_key=99, logits=logits, value=value, noise=noise)
        torch.cuda.synchronize()
        for k in range(K):
            # the observation the policy saw at step k: the state before the launch, then the previous step's row
            if rec:
                bo, di = traj["obs_board"][k], traj["obs_dice"][k]
                if k == 0:
                    assert torch.equal(bo, b0) and torch.equal(di, d0)
            else:
                bo, di = (b0, d0) if k == 0 else (traj["board"][k - 1], traj["dice"][k - 1])
            with torch.no_grad():
                l0, l1, v = model(bo, di)
            ref = torch.cat([l0, l1], 1)
            assert torch.allclose(logits[k], ref, atol=1e-5, rtol=0), (launch, k, float((logits[k] - ref).abs().max()))
            if want_value:
                assert torch.allclose(value[k], v, atol=1e-5, rtol=0), (launch, k, float((value[k] - v).abs().max()))
            # the noise is the engine's hash stream at that step, the action its Gumbel-max (argmax when deterministic)
            assert np.array_equal(cpu(noise[k, lo:hi]), orc.policy_noise(99)), (launch, k)
            act = traj["action"][k]
            if deterministic:
                exp = torch.stack([logits[k][:, :2].argmax(1), logits[k][:, 2:].argmax(1)], 1).to(torch.int8)
                assert torch.equal(act, exp)
            else:
                exp = gumbel_actions(logits[k], noise[k])
                bad = int((act != exp).any(1).sum())
                flips += bad
                total += N
                a = cpu(act.to(torch.int64))
                p = cpu(torch.cat([torch.softmax(logits[k][:, :2], 1), torch.softmax(logits[k][:, 2:], 1)], 1).double())
                counts += np.array([(a[:, 0] == 0).sum(), (a[:, 0] == 1).sum(), (a[:, 1] == 0).sum(), (a[:, 1] == 1).sum(), (a[:, 1] == 2).sum()])
                probs += p.sum(0)
            # every transition against the oracle, the recorded action fed in
            acts = cpu(act[lo:hi])
            live = ~frozen
            ob, od, r, te, tr, info = orc.step(np.where(live[:, None], acts, 0).astype(np.int8))
            ctx = (kw, launch, k)
            assert np.array_equal(cpu(traj["board"][k, lo:hi]), ob), ctx
            assert np.array_equal(cpu(traj["dice"][k, lo:hi]), od), ctx
            assert np.array_equal(bits(cpu(traj["reward"][k, lo:hi])), bits(r)), ctx
            assert np.array_equal(cpu(traj["terminated"][k, lo:hi]), te) and np.array_equal(cpu(traj["truncated"][k, lo:hi]), tr), ctx
            assert np.array_equal(cpu(traj["info"][k, lo:hi]), info), ctx
            nep += int((live & (te != 0)).sum())
            if not autoreset:
                frozen |= te != 0
        assert np.array_equal(cpu(env.board[lo:hi]), ob) and np.array_equal(cpu(env.dice[lo:hi]), od)
        assert np.array_equal(cpu(env.done[lo:hi]) != 0, frozen)
        if kw.get("shaped"):
            ps, tol, _ = orc.aux()
            assert np.array_equal(bits(cpu(env.prev_score[lo:hi])), bits(ps)) and np.array_equal(cpu(env.tolerance[lo:hi]), tol)
    if not deterministic:
        assert flips <= max(2, total // 20000), (flips, total)     # only near-ties of fp32 log rounding may differ
        chi2 = float((((counts - probs) ** 2) / np.maximum(probs, 1)).sum())
        assert chi2 < 40.0, (chi2, counts, probs)                   # 3 degrees of freedom; a policy that ignored its logits is off by thousands
    return nep


def test_policy_rollout_shaped_env_minimax_opponent(ea):
    """config 4's collector: MiniMaxHeuristicEnv semantics (shaped reward, tolerance), depth-3 minimax opponent, records with the
    initial observation, value requested; 3 000 games = 11 full blocks of 256 and a partial one"""
    kw = dict(opponent_policy="minimax", max_depth=3, shaped=True, reward=10.0, illegal_move_reward=-1.0, illegal_move_tolerance=4,
              shaped_refresh_on_reset=True, philox_key=9487)
    assert _policy_vs_torch_and_oracle(ea, 3000, 2600, 3000, 6, 3, **kw) > 100


@pytest.mark.parametrize("kw", [
    dict(opponent_policy="random"), dict(opponent_policy="minimax", max_depth=1), dict(opponent_policy="minimax", max_depth=4, heuristic="attk"),
    dict(opponent_policy="minimax", max_depth=2, shaped=True, reward=10.0, illegal_move_tolerance=10, shaped_refresh_on_reset=False),
], ids=lambda kw: "-".join("%s=%s" % kv for kv in sorted(kw.items())))
def test_policy_rollout_other_opponents(ea, kw):
    _policy_vs_torch_and_oracle(ea, 700, 100, 500, 5, 2, philox_key=3, **kw)


def test_policy_rollout_columns_no_value_deterministic_and_frozen_lanes(ea):
    _policy_vs_torch_and_oracle(ea, 600, 0, 300, 5, 2, layout="columns", want_value=False, opponent_policy="minimax", max_depth=3, philox_key=8)
    _policy_vs_torch_and_oracle(ea, 600, 200, 600, 6, 2, deterministic=True, opponent_policy="minimax", max_depth=3, philox_key=9)
    _policy_vs_torch_and_oracle(ea, 520, 0, 520, 9, 3, autoreset=False, opponent_policy="minimax", max_depth=3, philox_key=10)


def test_policy_rollout_7x7(ea):
    _policy_vs_torch_and_oracle(ea, 400, 100, 400, 5, 2, S=7, opponent_policy="minimax", max_depth=3, philox_key=12)
    _policy_vs_torch_and_oracle(ea, 400, 0, 200, 4, 2, S=7, opponent_policy="random", shaped=True, reward=10.0, philox_key=13)


def test_policy_rollout_at_config4_shape(ea):
    """65 536 lanes, shaped env, depth-3 opponent: a slice against torch and the oracle"""
    kw = dict(opponent_policy="minimax", max_depth=3, shaped=True, reward=10.0, illegal_move_reward=-1.0, illegal_move_tolerance=10,
              shaped_refresh_on_reset=True, philox_key=9487)
    _policy_vs_torch_and_oracle(ea, 65536, 65236, 65536, 5, 2, **kw)


def test_policy_rollout_is_unsupported_where_documented(ea):
    env = ea.VecEWN(64, opponent_policy="minimax", max_depth=5, rng="philox")
    assert not env.supports_policy_rollout()
    env = ea.VecEWN(64, opponent_policy="minimax", max_depth=3, rng="mt19937")
    assert not env.supports_policy_rollout()
    env = ea.VecEWN(64, board_size=6, opponent_policy="random", rng="philox")
    assert not env.supports_policy_rollout()
