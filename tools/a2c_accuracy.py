"""How close to exact arithmetic are the policy forward pass and the A2C gradient?  Reference: the same torch model in FLOAT64.
   python tools/a2c_accuracy.py            (run once per EWN_A2C_KERNEL=3|2|1: the kernel kind is read once per process)
Prints: max |logit - logit64| of ewn_step_k_policy, the relative l2 error of ewn_a2c_grad's gradient against the float64 gradient, and
the same two numbers for plain fp32 torch (what "fp32 accuracy" means on this problem)."""
import copy
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ewn_gym_amd as ea  # noqa: E402
from ewn_gym_amd._lib import EwnA2cHyper, check  # noqa: E402
from ewn_gym_amd.vec_env import _ptr, _stream  # noqa: E402
from tests.test_gpu_policy import make_model  # noqa: E402
from ewn_gym_amd.a2c import n_step_returns  # noqa: E402

N, S, K = 40000, 5, 5
env = ea.VecEWN(N, board_size=S, opponent_policy="minimax", max_depth=2, rng="philox", shaped=True, reward=10.0, illegal_move_tolerance=5,
                shaped_refresh_on_reset=True, autoreset=True, seed_stride=N, philox_key=21)
env.reset(seeds=(np.arange(N, dtype=np.uint64) + 3).astype(np.uint32))
model = make_model(S, 11, head_gain=1.0)
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


def _loss(m, t, dt):
    obs_b, obs_d = t["obs_board"], t["obs_dice"]
    with torch.no_grad():
        vals = torch.stack([m(obs_b[k], obs_d[k])[2] for k in range(K)])
        last = m(obs_b[K], obs_d[K])[2]
        adv, ret = n_step_returns(t["reward"].to(dt), vals, t["terminated"].to(dt), last, gamma, 1.0)
    logp, ent, value = m.evaluate_actions(obs_b[:K].reshape(K * N, S, S), obs_d[:K].reshape(K * N), t["action"].reshape(K * N, 2))
    return -(adv.reshape(-1) * logp).mean() + vf_coef * torch.nn.functional.mse_loss(ret.reshape(-1), value) - ent_coef * ent.mean()


def grad_of(m, t, dt):
    loss = _loss(m, t, dt)
    m.zero_grad()
    loss.backward()
    return torch.cat([p.grad.reshape(-1) for p in m.parameters()])


m64 = copy.deepcopy(model).double()
g64 = grad_of(m64, traj, torch.float64)
g32 = grad_of(model, traj, torch.float32)
with torch.no_grad():
    l64 = torch.stack([torch.cat(m64(traj["obs_board"][t], traj["obs_dice"][t])[:2], 1) for t in range(K)])
    l32 = torch.stack([torch.cat(model(traj["obs_board"][t], traj["obs_dice"][t])[:2], 1) for t in range(K)])
kind = os.environ.get("EWN_A2C_KERNEL", "3")
print("EWN_A2C_KERNEL=%s  (rollout forward: bf16 x 3)" % kind)
print("  logits   max |engine - f64| %.3e   max |torch fp32 - f64| %.3e" % (float((logits.double() - l64).abs().max()), float((l32.double() - l64).abs().max())))
print("  gradient |engine - f64| / |f64| %.3e   |torch fp32 - f64| / |f64| %.3e" %
      (float((grad[:-8].double() - g64).norm() / g64.norm()), float((g32.double() - g64).norm() / g64.norm())))
