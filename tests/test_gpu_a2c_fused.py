"""The fused A2C update (ewn_a2c_grad / ewn_a2c_apply) against torch autograd of the same loss on the same trajectory, and the
fused trainer end to end.  SB3 itself is absent (parity unpinned): the reference here is a plain fp32 torch implementation of
SB3's documented A2C loss (ewn_gym_amd.a2c: n-step returns with gae_lambda 1, policy gradient + vf_coef * MSE + ent_coef * -entropy,
clip_grad_norm_, RMSprop)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ea():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ewn_gym_amd
    return ewn_gym_amd


def _torch_loss(model, traj, K, gamma, vf_coef, ent_coef):
    from ewn_gym_amd.a2c import n_step_returns
    obs_b, obs_d = traj["obs_board"], traj["obs_dice"]
    N = obs_b.shape[1]
    S = obs_b.shape[2]
    with torch.no_grad():
        vals = torch.stack([model(obs_b[t], obs_d[t])[2] for t in range(K)])
        last = model(obs_b[K], obs_d[K])[2]
        adv, ret = n_step_returns(traj["reward"].float(), vals, traj["terminated"].float(), last, gamma, 1.0)
    logp, ent, value = model.evaluate_actions(obs_b[:K].reshape(K * N, S, S), obs_d[:K].reshape(K * N), traj["action"].reshape(K * N, 2))
    pl = -(adv.reshape(-1) * logp).mean()
    vl = torch.nn.functional.mse_loss(ret.reshape(-1), value)
    en = ent.mean()
    return pl + vf_coef * vl - ent_coef * en, pl, vl, en


@pytest.mark.parametrize("N,S,K,ent_coef,opp", [(3000, 5, 5, 0.0, "minimax"), (1000, 5, 3, 0.01, "random"), (40000, 5, 5, 0.0, "minimax"),
                                                (700, 7, 4, 0.02, "minimax")])
def test_fused_gradient_matches_torch_autograd(ea, N, S, K, ent_coef, opp):
    from ewn_gym_amd._lib import EwnA2cHyper, check
    from ewn_gym_amd.vec_env import _ptr, _stream
    from tests.test_gpu_policy import make_model
    env = ea.VecEWN(N, board_size=S, opponent_policy=opp, max_depth=2, rng="philox", shaped=True, reward=10.0, illegal_move_tolerance=5,
                    shaped_refresh_on_reset=True, autoreset=True, seed_stride=N, philox_key=21)
    env.reset(seeds=(np.arange(N, dtype=np.uint64) + 3).astype(np.uint32))
    model = make_model(S, 11, head_gain=1.0)
    params = model.flat_parameters()
    traj = env.alloc_rollout(K, layout="record", initial_obs=True)
    for _ in range(3):            # a few launches so that the trajectories hold terminal steps, resets and tolerance penalties
        env.rollout_policy(K, params, traj=traj, noise_key=5)
    gamma, vf_coef = 0.97, 0.5
    hp = EwnA2cHyper(gamma, vf_coef, ent_coef, 0.5, 7e-4, 0.99, 1e-5, 1)
    nscr = check(env.lib.ewn_a2c_scratch_bytes(C.byref(env.cfg), K))
    scratch = torch.zeros(int(nscr), dtype=torch.uint8, device="cuda")
    grad = torch.zeros(params.numel() + 8, dtype=torch.float32, device="cuda")
    check(env.lib.ewn_a2c_grad(C.byref(env.cfg), K, _ptr(traj["record"]), _ptr(traj["reward"]), _ptr(params), C.byref(hp), _ptr(grad),
                               _ptr(scratch), _stream()), "ewn_a2c_grad")
    loss, pl, vl, en = _torch_loss(model, traj, K, gamma, vf_coef, ent_coef)
    model.zero_grad()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    pl, vl, en = pl.detach(), vl.detach(), en.detach()
    got = grad[:-8]
    rel = float((got - ref).norm() / ref.norm())
    assert rel < 2e-4, (rel, float(ref.norm()))
    assert torch.allclose(got, ref, rtol=5e-3, atol=2e-5 * float(ref.abs().max())), float((got - ref).abs().max())
    # every parameter block on its own (a mis-indexed block would hide in the global norm)
    off = 0
    for name, p in model.named_parameters():
        a, b = got[off:off + p.numel()], ref[off:off + p.numel()]
        off += p.numel()
        if float(b.norm()) > 0:
            assert float((a - b).norm() / b.norm()) < 2e-3, name
    st = grad[-8:].tolist()
    n = K * N
    assert abs(st[0] / n - float(pl)) < 1e-4 * max(1.0, abs(float(pl))) and abs(st[5] / n - float(vl)) < 1e-4 * max(1.0, float(vl))
    assert abs(st[2] / n - float(en)) < 1e-4
    # bit-reproducible: the same call again gives the same bits
    grad2 = torch.zeros_like(grad)
    check(env.lib.ewn_a2c_grad(C.byref(env.cfg), K, _ptr(traj["record"]), _ptr(traj["reward"]), _ptr(params), C.byref(hp), _ptr(grad2),
                               _ptr(scratch), _stream()), "ewn_a2c_grad")
    assert torch.equal(grad, grad2)


def test_fused_apply_is_clip_plus_rmsprop(ea):
    from ewn_gym_amd._lib import EwnA2cHyper, check
    from ewn_gym_amd.vec_env import _ptr, _stream
    env = ea.VecEWN(64, opponent_policy="random", rng="philox")
    P = env.policy_param_count()
    g = torch.Generator(device="cuda").manual_seed(0)
    for scale, world in ((3.0, 1), (1e-3, 1), (2.0, 4)):
        params = torch.randn(P, device="cuda", generator=g)
        grad = torch.randn(P + 8, device="cuda", generator=g) * scale
        ref_p = torch.nn.Parameter(params.clone())
        opt = torch.optim.RMSprop([ref_p], lr=7e-4, alpha=0.99, eps=1e-5)
        sq = torch.zeros(P, device="cuda")
        norm = torch.zeros(1, device="cuda")
        hp = EwnA2cHyper(0.99, 0.5, 0.0, 0.5, 7e-4, 0.99, 1e-5, world)
        for it in range(3):
            ref_p.grad = grad[:P].clone() / world
            tn = torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
            opt.step()
            check(env.lib.ewn_a2c_apply(C.byref(env.cfg), _ptr(params), _ptr(sq), _ptr(grad), C.byref(hp), _ptr(norm), _stream()))
            assert abs(float(norm) - float(tn)) < 1e-4 * float(tn)
            assert torch.allclose(params, ref_p.data, rtol=1e-5, atol=1e-6), (scale, it)


def test_fused_trainer_learns_and_matches_the_torch_trainer_step(ea):
    """One update of FusedA2CTrainer == the same update done by torch (autograd + clip + RMSprop) on the trajectory it collected;
    a short run beats the illegal-move habit of a fresh policy (mean reward rises)."""
    from ewn_gym_amd.a2c import ActorCritic, FusedA2CTrainer
    N = 8192
    env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0,
                    illegal_move_tolerance=10, shaped_refresh_on_reset=True, autoreset=True, seed_stride=N, philox_key=9487)
    env.reset(seeds=(np.arange(N, dtype=np.uint64) + 9487).astype(np.uint32))
    tr = FusedA2CTrainer(env, n_steps=5, learning_rate=7e-4, seed=1, use_graph=True)
    before = tr.params.clone()
    tr.collect_and_update()        # eager
    torch.cuda.synchronize()
    ref = ActorCritic(5, 6).cuda()
    ref.load_flat_parameters(before)
    loss, _, _, _ = _torch_loss(ref, tr.traj, 5, 0.99, 0.5, 0.0)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
    opt = torch.optim.RMSprop(ref.parameters(), lr=7e-4, alpha=0.99, eps=1e-5)
    opt.step()
    assert torch.allclose(tr.params, ref.flat_parameters(), rtol=1e-4, atol=2e-6), float((tr.params - ref.flat_parameters()).abs().max())
    # the module views the flat vector: the torch-side policy sees the update
    assert torch.equal(tr.model.flat_parameters(), tr.params)
    first = None
    for it in range(150):          # graph replays from the second update on
        tr.collect_and_update()
        if it == 1:
            first = tr.stats_dict()
    last = tr.stats_dict()
    assert np.isfinite(last["loss"]) and last["grad_norm"] > 0
    assert last["mean_reward"] > first["mean_reward"] + 0.05, (first, last)


def test_bf16x3_forward_and_gradient_are_fp32_accurate(ea):
    """The network runs on the bf16 matrix pipe with every operand split into three bf16 parts (csrc/ewn_mlp3.hpp).  The claim is fp32
    accuracy: against the same model evaluated in FLOAT64 the engine's logits and gradient must be as close as plain fp32 torch is
    (measured, tools/a2c_accuracy.py: logits 6.9e-7 vs 1.0e-6, gradient 5.6e-8 vs 6.3e-8 relative)."""
    import copy
    from ewn_gym_amd._lib import EwnA2cHyper, check
    from ewn_gym_amd.a2c import n_step_returns
    from ewn_gym_amd.vec_env import _ptr, _stream
    from tests.test_gpu_policy import make_model
    N, S, K = 6000, 5, 5
    env = ea.VecEWN(N, board_size=S, opponent_policy="minimax", max_depth=2, rng="philox", shaped=True, reward=10.0, illegal_move_tolerance=5,
                    shaped_refresh_on_reset=True, autoreset=True, seed_stride=N, philox_key=23)
    env.reset(seeds=(np.arange(N, dtype=np.uint64) + 5).astype(np.uint32))
    model = make_model(S, 13, head_gain=1.0)
    params = model.flat_parameters()
    traj = env.alloc_rollout(K, layout="record", initial_obs=True)
    logits = torch.zeros((K, N, 5), dtype=torch.float32, device="cuda")
    for _ in range(3):
        env.rollout_policy(K, params, traj=traj, noise_key=5, logits=logits)
    gamma, vf_coef, ent_coef = 0.97, 0.5, 0.01
    hp = EwnA2cHyper(gamma, vf_coef, ent_coef, 0.5, 7e-4, 0.99, 1e-5, 1)
    scratch = torch.zeros(int(check(env.lib.ewn_a2c_scratch_bytes(C.byref(env.cfg), K))), dtype=torch.uint8, device="cuda")
    grad = torch.zeros(params.numel() + 8, dtype=torch.float32, device="cuda")
    check(env.lib.ewn_a2c_grad(C.byref(env.cfg), K, _ptr(traj["record"]), _ptr(traj["reward"]), _ptr(params), C.byref(hp), _ptr(grad),
                               _ptr(scratch), _stream()), "ewn_a2c_grad")
    m64 = copy.deepcopy(model).double()
    obs_b, obs_d = traj["obs_board"], traj["obs_dice"]
    with torch.no_grad():
        l64 = torch.stack([torch.cat(m64(obs_b[t], obs_d[t])[:2], 1) for t in range(K)])
        vals = torch.stack([m64(obs_b[t], obs_d[t])[2] for t in range(K)])
        adv, ret = n_step_returns(traj["reward"].double(), vals, traj["terminated"].double(), m64(obs_b[K], obs_d[K])[2], gamma, 1.0)
    logp, ent, value = m64.evaluate_actions(obs_b[:K].reshape(K * N, S, S), obs_d[:K].reshape(K * N), traj["action"].reshape(K * N, 2))
    loss = -(adv.reshape(-1) * logp).mean() + vf_coef * torch.nn.functional.mse_loss(ret.reshape(-1), value) - ent_coef * ent.mean()
    loss.backward()
    g64 = torch.cat([p.grad.reshape(-1) for p in m64.parameters()])
    assert float((logits.double() - l64).abs().max()) < 3e-6
    assert float((grad[:-8].double() - g64).norm() / g64.norm()) < 5e-7
