"""Tournament (eval_*.py / eval_pairs.py counterpart) and A2C trainer on the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_eval_loop_bit_exact_vs_reference(golden):
    """eval_minimax.py:16-50 with a deterministic agent: per-episode final reward and length are identical."""
    from ewn_gym_amd.tournament import evaluate
    for rec in golden("g10_eval_loop.json"):
        opp = {"kind": rec["opp"]}
        if rec["opp_depth"]:
            opp["max_depth"] = rec["opp_depth"]
        r = evaluate({"kind": "minimax", "max_depth": rec["agent_depth"]}, opp, num=len(rec["scores"]), rng="mt19937")
        assert r["scores"].cpu().tolist() == rec["scores"]
        assert r["lengths"].cpu().tolist() == rec["lengths"]
        assert r["wins"] == sum(s > 0 for s in rec["scores"])


def test_win_rates_in_the_published_bands():
    """assets/p22.png (read off the bars, +-3 %): Minimax(5) vs Random ~0.92, Random vs Minimax(5) ~0.11,
    Minimax(5) vs Minimax(5) ~0.55.  512 episodes: +-0.07 covers chart-reading error plus 3 sigma."""
    from ewn_gym_amd.tournament import evaluate, wilson
    mm5 = {"kind": "minimax", "max_depth": 5}
    r1 = evaluate(mm5, {"kind": "random"}, num=512)
    r2 = evaluate({"kind": "random"}, mm5, num=512)
    r3 = evaluate(mm5, mm5, num=256)
    assert abs(r1["win_rate"] - 0.92) < 0.07, r1["win_rate"]
    assert abs(r2["win_rate"] - 0.11) < 0.07, r2["win_rate"]
    assert abs(r3["win_rate"] - 0.55) < 0.12, r3["win_rate"]
    lo, hi = wilson(460, 512)
    assert lo < 460 / 512 < hi and hi - lo < 0.06


def test_mcts_agent_beats_random_clearly():
    from ewn_gym_amd.tournament import evaluate
    r = evaluate({"kind": "mcts", "num_simulations": 10, "num_env_copies": 5}, {"kind": "random"}, num=256, rng="philox")
    assert r["win_rate"] > 0.75, r["win_rate"]


def test_a2c_learns_to_avoid_illegal_moves():
    import ewn_gym_amd as ea
    from ewn_gym_amd.a2c import A2CTrainer
    N = 4096
    env = ea.VecEWN(N, opponent_policy="random", rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0,
                    illegal_move_tolerance=10, autoreset=True, shaped_refresh_on_reset=True, philox_key=1)
    env.reset(seeds=torch.arange(N, dtype=torch.int32))
    tr = A2CTrainer(env, n_steps=5, learning_rate=1e-3, seed=0)
    p0 = [p.detach().clone() for p in tr.model.parameters()]

    def illegal_rate(steps=12):
        """fraction of env transitions the env itself flags as illegal agent moves (info codes 1 and 5)"""
        bad = tot = 0
        for _ in range(steps):
            a, _ = tr.model.act(env.board, env.dice, deterministic=False, generator=tr.gen)
            info = env.step(a)[5]
            bad += int(((info == 1) | (info == 5)).sum().item())
            tot += N
        return bad / tot

    before = illegal_rate()
    stats = None
    for _ in range(300):
        stats = tr.collect_and_update()
    after = illegal_rate()
    stats = tr.stats_dict(stats)
    assert all(np.isfinite(v) for v in stats.values())
    assert any(not torch.equal(a, b) for a, b in zip(p0, tr.model.parameters()))
    assert tr.num_timesteps == 300 * 5 * N
    assert after < 0.6 * before, (before, after)


def test_ppo_learns_to_avoid_illegal_moves():
    """train.py's PPO sub-command on the device-resident rollout (ewn_gym_amd/ppo.py; SB3 defaults, parity unpinned)"""
    import ewn_gym_amd as ea
    from ewn_gym_amd.ppo import PPOTrainer
    N = 4096
    env = ea.VecEWN(N, opponent_policy="random", rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0,
                    illegal_move_tolerance=10, autoreset=True, shaped_refresh_on_reset=True, philox_key=2)
    env.reset(seeds=torch.arange(N, dtype=torch.int32))
    tr = PPOTrainer(env, n_steps=8, n_epochs=4, learning_rate=1e-3, seed=0)

    def illegal_rate(steps=12):
        bad = tot = 0
        for _ in range(steps):
            a, _ = tr.model.act(env.board, env.dice, deterministic=False, generator=tr.gen)
            info = env.step(a)[5]
            bad += int(((info == 1) | (info == 5)).sum().item())
            tot += N
        return bad / tot

    before = illegal_rate()
    stats = None
    for _ in range(40):
        stats = tr.collect_and_update()
    after = illegal_rate()
    stats = tr.stats_dict(stats)
    assert all(np.isfinite(v) for v in stats.values())
    assert tr.num_timesteps == 40 * 8 * N
    assert after < 0.6 * before, (before, after)


def test_checkpoint_resume(tmp_path):
    """train.py:137-139 / --checkpoint: a saved trainer resumes with the same parameters, optimiser state and step count"""
    import ewn_gym_amd as ea
    from ewn_gym_amd.a2c import A2CTrainer
    N = 1024

    def make():
        env = ea.VecEWN(N, opponent_policy="minimax", max_depth=2, rng="philox", shaped=True, reward=10.0, autoreset=True, philox_key=5,
                        shaped_refresh_on_reset=True)
        env.reset(seeds=torch.arange(N, dtype=torch.int32))
        return A2CTrainer(env, n_steps=4, learning_rate=1e-3, seed=3)

    a = make()
    for _ in range(5):
        a.collect_and_update()
    path = str(tmp_path / "ckpt.pt")
    a.save(path)
    b = make()
    b.load(path)
    assert b.num_timesteps == a.num_timesteps == 5 * 4 * N
    for p, q in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(p, q)
    sa, sb = a.opt.state_dict()["state"], b.opt.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k]["square_avg"], sb[k]["square_avg"]) for k in sa)
