"""A2C on the vectorised env: the counterpart of train.py:35-159.

The reference trains with stable_baselines3's A2C ("MultiInputPolicy", tanh, RMSprop) over
SubprocVecEnv workers; SB3 is not vendored, not pinned and not installed here, so its
arithmetic is **parity unpinned** -- this module follows SB3's documented defaults, not a
byte-level contract:
  * features: board flattened to S*S floats ++ one-hot(dice_roll) of width cube_num+1
    (the Discrete(cube_num+1, start=1) observation space, envs/ewn.py:66-68);
  * separate 64-64 tanh networks for policy and value, orthogonal init (gain sqrt2 / 0.01 / 1);
  * MultiDiscrete([2,3]) action head = two independent categoricals;
  * n-step returns with GAE(lambda=1), no advantage normalisation, vf_coef 0.5, ent_coef 0,
    max_grad_norm 0.5, RMSprop(alpha 0.99, eps 1e-5), lr 7e-4 unless given (train.py passes its own).
Observations never leave the GPU: the policy consumes VecEWN.board / .dice directly.

Multi-GPU: one process per GPU, each owning its own lanes; the only collective on the
training path is ONE all-reduce (RCCL under torch.distributed's "nccl" backend) of the
flattened gradient (~13 k fp32 = 52 KB: latency-bound on xGMI, a single bucket by design).
"""
import math

import torch
import torch.nn as nn


class ActorCritic(nn.Module):
    def __init__(self, board_size=5, cube_num=6, hidden=64):
        super().__init__()
        self.S, self.cube_num = board_size, cube_num
        feat = board_size * board_size + cube_num + 1
        self.pi = nn.Sequential(nn.Linear(feat, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.vf = nn.Sequential(nn.Linear(feat, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.action_net = nn.Linear(hidden, 5)   # logits of MultiDiscrete([2, 3])
        self.value_net = nn.Linear(hidden, 1)
        for seq in (self.pi, self.vf):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=math.sqrt(2))
                    nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.action_net.weight, gain=0.01)
        nn.init.zeros_(self.action_net.bias)
        nn.init.orthogonal_(self.value_net.weight, gain=1.0)
        nn.init.zeros_(self.value_net.bias)

    def features(self, board, dice):
        """board int8 [N,S,S], dice int8 [N] (1..cube_num) -> float32 [N, S*S + cube_num + 1]"""
        b = board.reshape(board.shape[0], -1).to(torch.float32)
        oh = torch.nn.functional.one_hot((dice.to(torch.int64) - 1).clamp_(0, self.cube_num), self.cube_num + 1).to(torch.float32)
        return torch.cat([b, oh], dim=1)

    def forward(self, board, dice):
        x = self.features(board, dice)
        logits = self.action_net(self.pi(x))
        value = self.value_net(self.vf(x)).squeeze(-1)
        return logits[:, :2], logits[:, 2:], value

    @torch.no_grad()
    def act(self, board, dice, deterministic=False, generator=None):
        l0, l1, value = self(board, dice)
        if deterministic:
            a0, a1 = l0.argmax(1), l1.argmax(1)
        else:
            a0 = torch.multinomial(torch.softmax(l0, 1), 1, generator=generator).squeeze(1)
            a1 = torch.multinomial(torch.softmax(l1, 1), 1, generator=generator).squeeze(1)
        return torch.stack([a0, a1], 1).to(torch.int8), value

    def evaluate_actions(self, board, dice, actions):
        l0, l1, value = self(board, dice)
        lp0, lp1 = torch.log_softmax(l0, 1), torch.log_softmax(l1, 1)
        a = actions.to(torch.int64)
        logp = lp0.gather(1, a[:, :1]).squeeze(1) + lp1.gather(1, a[:, 1:2]).squeeze(1)
        ent = -(lp0.exp() * lp0).sum(1) - (lp1.exp() * lp1).sum(1)
        return logp, ent, value


def n_step_returns(rewards, values, dones, last_value, gamma=0.99, gae_lambda=1.0):
    """SB3 RolloutBuffer.compute_returns_and_advantage.  rewards/values/dones: [T, N]; dones[t] = episode ended AT step t.
    Returns (advantages, returns)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_value)
    for t in reversed(range(T)):
        next_value = last_value if t == T - 1 else values[t + 1]
        nonterminal = 1.0 - dones[t]
        delta = rewards[t] + gamma * next_value * nonterminal - values[t]
        last = delta + gamma * gae_lambda * nonterminal * last
        adv[t] = last
    return adv, adv + values


def all_reduce_gradients(params, world_size=None):
    """ONE collective per update: flatten every gradient into a single bucket, all-reduce (sum), divide, scatter back."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size() if world_size is None else world_size
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class A2CTrainer:
    def __init__(self, env, n_steps=5, learning_rate=7e-4, gamma=0.99, gae_lambda=1.0, ent_coef=0.0, vf_coef=0.5,
                 max_grad_norm=0.5, seed=None, hidden=64, device=None):
        self.env = env
        self.device = torch.device(device) if device is not None else env.board.device
        if seed is not None:
            torch.manual_seed(seed)
        self.model = ActorCritic(env.S, env.cube_num, hidden).to(self.device)
        self._sync_parameters()
        self.opt = torch.optim.RMSprop(self.model.parameters(), lr=learning_rate, alpha=0.99, eps=1e-5)
        self.n_steps, self.gamma, self.gae_lambda = n_steps, gamma, gae_lambda
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed + 7919 * getattr(env, "lane_offset", 0))
        self.num_timesteps = 0

    def _sync_parameters(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)

    def collect_and_update(self):
        env, T, N = self.env, self.n_steps, self.env.N
        boards, dices, acts, rews, dones, vals = [], [], [], [], [], []
        for _ in range(T):
            b, d = env.board.clone(), env.dice.clone()
            a, v = self.model.act(b, d, generator=self.gen)
            _, _, r, term, _, _ = env.step(a)
            boards.append(b); dices.append(d); acts.append(a); vals.append(v)
            rews.append(r.to(torch.float32)); dones.append((term != 0).to(torch.float32))
        with torch.no_grad():
            _, _, last_value = self.model(env.board, env.dice)
        rews, dones, vals = torch.stack(rews), torch.stack(dones), torch.stack(vals)
        adv, ret = n_step_returns(rews, vals, dones, last_value, self.gamma, self.gae_lambda)
        logp, ent, value = self.model.evaluate_actions(torch.cat(boards), torch.cat(dices), torch.cat(acts))
        policy_loss = -(adv.reshape(-1) * logp).mean()
        value_loss = torch.nn.functional.mse_loss(ret.reshape(-1), value)
        entropy_loss = -ent.mean()
        loss = policy_loss + self.ent_coef * entropy_loss + self.vf_coef * value_loss
        self.opt.zero_grad(set_to_none=False)
        loss.backward()
        all_reduce_gradients(list(self.model.parameters()))
        nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.opt.step()
        self.num_timesteps += T * N
        return {"loss": float(loss.item()), "policy_loss": float(policy_loss.item()), "value_loss": float(value_loss.item()),
                "entropy": float(-entropy_loss.item()), "mean_reward": float(rews.mean().item()),
                "episodes": int(dones.sum().item())}

    def learn(self, total_timesteps):
        stats = None
        target = self.num_timesteps + total_timesteps
        while self.num_timesteps < target:
            stats = self.collect_and_update()
        return stats

    def policy_fn(self, deterministic=True):
        """A batched policy for tournament.evaluate / hand-rolled loops: (board, dice, t) -> int8 [N,2]"""
        return lambda b, d, t: self.model.act(b, d, deterministic=deterministic, generator=self.gen)[0]

    def save(self, path):
        torch.save({"model": self.model.state_dict(), "opt": self.opt.state_dict(), "num_timesteps": self.num_timesteps}, path)

    def load(self, path):
        sd = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(sd["model"])
        self.opt.load_state_dict(sd["opt"])
        self.num_timesteps = sd["num_timesteps"]
