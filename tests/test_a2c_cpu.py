"""Trainer counterpart, CPU side: return computation, feature layout, and the one
gradient all-reduce of the multi-GPU path exercised with two gloo ranks."""
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ewn_gym_amd.a2c import ActorCritic, all_reduce_gradients, n_step_returns


def test_n_step_returns_match_a_plain_loop():
    g = torch.Generator().manual_seed(0)
    T, N, gamma = 7, 5, 0.9
    r = torch.randn(T, N, generator=g)
    v = torch.randn(T, N, generator=g)
    d = (torch.rand(T, N, generator=g) < 0.3).float()
    lv = torch.randn(N, generator=g)
    adv, ret = n_step_returns(r, v, d, lv, gamma, 1.0)
    for n in range(N):
        for t in range(T):
            G, disc = 0.0, 1.0
            k = t
            while True:
                G += disc * r[k, n].item()
                if d[k, n] == 1:
                    break
                disc *= gamma
                k += 1
                if k == T:
                    G += disc * lv[n].item()
                    break
            assert abs(ret[t, n].item() - G) < 1e-5
    assert torch.allclose(adv, ret - v)


def test_features_and_parameter_budget():
    m = ActorCritic(5, 6)
    board = torch.zeros(3, 5, 5, dtype=torch.int8)
    board[0, 0, 0] = 4
    board[1, 4, 4] = -2
    x = m.features(board, torch.tensor([1, 6, 3], dtype=torch.int8))
    assert x.shape == (3, 25 + 7)
    assert x[0, 0] == 4 and x[1, 24] == -2
    assert x[:, 25:].argmax(1).tolist() == [0, 5, 2] and x[:, 25:].sum(1).tolist() == [1, 1, 1]
    n = sum(p.numel() for p in m.parameters())
    assert 12000 < n < 14000          # ~13 k fp32 = 52 KB: one all-reduce bucket
    a, v = m.act(board, torch.tensor([1, 6, 3], dtype=torch.int8), deterministic=True)
    assert a.shape == (3, 2) and a.dtype == torch.int8 and bool((a[:, 0] < 2).all()) and bool((a[:, 1] < 3).all())


def _worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(0)
    m = ActorCritic(5, 6)
    g = torch.Generator().manual_seed(100 + rank)
    board = torch.randint(-6, 7, (16, 5, 5), generator=g).to(torch.int8)
    dice = torch.randint(1, 7, (16,), generator=g).to(torch.int8)
    acts = torch.stack([torch.randint(0, 2, (16,), generator=g), torch.randint(0, 3, (16,), generator=g)], 1)
    logp, ent, val = m.evaluate_actions(board, dice, acts)
    (logp.mean() + val.pow(2).mean()).backward()
    local = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    all_reduce_gradients(list(m.parameters()))
    red = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    q.put((rank, local.tolist(), red.tolist()))   # by value: a tensor travels as a shared-memory handle that dies with this process
    dist.barrier()
    dist.destroy_process_group()


def test_single_bucket_gradient_all_reduce_with_two_gloo_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (torch.tensor(l), torch.tensor(g))) for r, l, g in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean = (res[0][0] + res[1][0]) / 2
    assert torch.allclose(res[0][1], mean, atol=1e-7) and torch.allclose(res[1][1], mean, atol=1e-7)


def test_split_k_linear_matches_nn_linear():
    """The batched-partial-product weight gradient (a2c.SplitKLinear) is the same gradient as nn.Linear's, up to fp32 summation order."""
    from ewn_gym_amd.a2c import SplitKLinear
    torch.manual_seed(0)
    a = SplitKLinear(32, 64)
    b = torch.nn.Linear(32, 64)
    b.load_state_dict(a.state_dict())
    x = torch.randn(8192, 32, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    ya, yb = a(x), b(x2)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(ya, yb) and torch.equal(x.grad, x2.grad)
    assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-3)
    small = torch.randn(16, 32)
    assert torch.equal(a(small), b(small))


class _FakeEnv:
    """CPU stand-in with VecEWN's attribute surface (the real engine needs a GPU): fixed observation, zero reward"""

    def __init__(self, n, lane_offset):
        import types
        self.N, self.S, self.cube_num = n, 5, 6
        self.cfg = types.SimpleNamespace(lane_offset=lane_offset)
        self.board = torch.zeros((n, 5, 5), dtype=torch.int8)
        self.dice = torch.ones(n, dtype=torch.int8)
        self._r = torch.zeros(n, dtype=torch.float64)
        self._t = torch.zeros(n, dtype=torch.uint8)

    def step(self, a):
        return self.board, self.dice, self._r, self._t, self._t, self._t


def test_ranks_share_parameters_but_not_sampling_noise():
    """Every rank seeds the same parameters (then broadcast), but the exploration noise is keyed by the rank's first global
    lane: two shards of one job must not draw the same Gumbel noise (perfectly correlated exploration)."""
    from ewn_gym_amd.a2c import A2CTrainer
    a = A2CTrainer(_FakeEnv(64, 0), n_steps=3, seed=5, use_graph=False)
    b = A2CTrainer(_FakeEnv(64, 64), n_steps=3, seed=5, use_graph=False)
    a2 = A2CTrainer(_FakeEnv(64, 0), n_steps=3, seed=5, use_graph=False)
    for p, q in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(p, q)
    for t in (a, b, a2):
        t._rollout()
    assert torch.equal(a._acts, a2._acts)          # same shard, same seed: reproducible
    assert not torch.equal(a._acts, b._acts)       # other shard: other noise


class _NoisyEnv(_FakeEnv):
    """... with observations, rewards and episode ends that vary, so that advantages and log-probabilities are not degenerate"""

    def __init__(self, n, lane_offset):
        super().__init__(n, lane_offset)
        self.g = torch.Generator().manual_seed(7 + lane_offset)

    def step(self, a):
        self.board = torch.randint(-6, 7, (self.N, 5, 5), generator=self.g).to(torch.int8)
        self.dice = torch.randint(1, 7, (self.N,), generator=self.g).to(torch.int8)
        r = torch.randn(self.N, generator=self.g, dtype=torch.float64)
        t = (torch.rand(self.N, generator=self.g) < 0.2).to(torch.uint8)
        return self.board, self.dice, r, t, self._t, self._t


def test_ppo_first_step_is_the_a2c_policy_gradient_and_clipping_bites_later():
    """With one epoch, ONE minibatch (the whole buffer) and no advantage normalisation the probability ratio is 1 everywhere at the
    first optimiser step, so PPO's clipped surrogate has A2C's policy gradient (train.py:39-49: SB3 PPO; maths only, parity unpinned).
    After several epochs on the same buffer the ratio leaves [1 - clip, 1 + clip] for some samples and their gradient vanishes."""
    from ewn_gym_amd.a2c import n_step_returns
    from ewn_gym_amd.ppo import PPOTrainer
    T, N = 4, 96
    tr = PPOTrainer(_NoisyEnv(N, 0), n_steps=T, batch_size=T * N, n_epochs=1, normalize_advantage=False, seed=11, use_graph=False,
                    learning_rate=1e-2)
    tr._rollout()
    boards, dices, acts = tr._boards.reshape(T * N, 5, 5), tr._dices.reshape(T * N), tr._acts.reshape(T * N, 2)
    with torch.no_grad():
        _, _, lv = tr.model(tr.env.board, tr.env.dice)
        adv, ret = n_step_returns(tr._rews, tr._vals, tr._dones, lv, tr.gamma, tr.gae_lambda)
        old_logp, _, _ = tr.model.evaluate_actions(boards, dices, acts)
    adv, ret = adv.reshape(-1), ret.reshape(-1)
    loss, pl, vl, en, cf = tr.ppo_loss(boards, dices, acts, old_logp, adv, ret)
    assert float(cf) == 0.0
    tr.model.zero_grad()
    loss.backward()
    g_ppo = torch.cat([p.grad.reshape(-1) for p in tr.model.parameters()]).clone()
    logp, ent, value = tr.model.evaluate_actions(boards, dices, acts)
    a2c = -(adv * logp).mean() + tr.vf_coef * torch.nn.functional.mse_loss(ret, value)
    tr.model.zero_grad()
    a2c.backward()
    g_a2c = torch.cat([p.grad.reshape(-1) for p in tr.model.parameters()])
    assert torch.allclose(g_ppo, g_a2c, rtol=1e-4, atol=1e-6)
    # many passes over one buffer: the policy moves, ratios leave the trust region, the clip fraction becomes positive
    tr2 = PPOTrainer(_NoisyEnv(N, 0), n_steps=T, batch_size=T * N // 2, n_epochs=30, seed=11, use_graph=False, learning_rate=3e-2)
    p0 = [p.detach().clone() for p in tr2.model.parameters()]
    st = tr2.collect_and_update()
    assert torch.isfinite(st).all() and tr2.num_timesteps == T * N
    assert any(not torch.equal(a, b) for a, b in zip(p0, tr2.model.parameters()))
    with torch.no_grad():
        new_logp, _, _ = tr2.model.evaluate_actions(tr2._boards.reshape(T * N, 5, 5), tr2._dices.reshape(T * N), tr2._acts.reshape(T * N, 2))
    assert float((new_logp.exp().sum())) > 0.0


def test_ppo_rejects_a_minibatch_larger_than_the_buffer():
    import pytest
    from ewn_gym_amd.ppo import PPOTrainer
    with pytest.raises(ValueError):
        PPOTrainer(_FakeEnv(8, 0), n_steps=2, batch_size=17, use_graph=False)


def test_checkpoint_round_trip_and_algorithm_check(tmp_path):
    """--checkpoint (train.py:137-139): model, optimiser, step counter, best score and the sampling generator come back; a
    checkpoint of the other algorithm is refused with a clear error instead of mis-loading its optimiser state"""
    import pytest
    from ewn_gym_amd.a2c import A2CTrainer
    from ewn_gym_amd.ppo import PPOTrainer
    t = A2CTrainer(_FakeEnv(8, 0), n_steps=2, seed=3, device="cpu", use_graph=False)
    t.best_score, t.num_timesteps = 0.4, 123
    torch.rand(5, generator=t.gen)
    path = str(tmp_path / "ck.pt")
    t.save(path)
    nxt = torch.rand(4, generator=t.gen)
    u = A2CTrainer(_FakeEnv(8, 0), n_steps=2, seed=99, device="cpu", use_graph=False)
    u.load(path)
    assert u.best_score == 0.4 and u.num_timesteps == 123
    assert torch.equal(torch.rand(4, generator=u.gen), nxt)
    for a, b in zip(t.model.parameters(), u.model.parameters()):
        assert torch.equal(a, b)
    p = PPOTrainer(_FakeEnv(8, 0), n_steps=2, seed=3, device="cpu", use_graph=False)
    with pytest.raises(ValueError, match="A2C trainer"):
        p.load(path)
