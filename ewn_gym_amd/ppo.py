"""PPO on the vectorised env: the `PPO` sub-command of the reference's train.py (train.py:39-49, 178-183).

The reference builds `stable_baselines3.PPO("MultiInputPolicy", env, batch_size=..., learning_rate=..., tanh)` and leaves
everything else at SB3's defaults.  SB3 is not vendored, not pinned and not installed here, so -- exactly as for
`ewn_gym_amd.a2c` -- its arithmetic is **parity unpinned**: this module follows SB3's documented PPO defaults
  * clipped surrogate objective, clip_range 0.2, no value-function clipping;
  * GAE(lambda 0.95), gamma 0.99, advantages normalised per minibatch;
  * n_epochs 10 passes over the rollout buffer in shuffled minibatches;
  * vf_coef 0.5, ent_coef 0, max_grad_norm 0.5, Adam(eps 1e-5), lr 3e-4;
  * the same two 64-64 tanh networks and action heads as A2C (`a2c.ActorCritic`),
and is tested for its own maths (tests/test_a2c_cpu.py: with one epoch, one minibatch and no normalisation the first PPO
step IS the A2C policy gradient), not against SB3.

What differs from SB3 by construction: the rollout buffer is n_steps x LANES (tens of thousands of lanes instead of
a handful of SubprocVecEnv workers), observations never leave the GPU, and `batch_size` counts samples of that
buffer (default: a quarter of it, i.e. four minibatches per epoch; train.py's `-b 8` would mean 10^5 optimiser steps
per rollout here).  Multi-GPU: as in A2C, ONE flattened-gradient all-reduce per optimiser step.
"""
import torch
import torch.nn as nn

from .a2c import A2CTrainer, all_reduce_gradients, n_step_returns


class PPOTrainer(A2CTrainer):
    algorithm = "PPO"

    def __init__(self, env, n_steps=32, batch_size=None, n_epochs=10, learning_rate=3e-4, gamma=0.99, gae_lambda=0.95,
                 clip_range=0.2, normalize_advantage=True, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, seed=None,
                 hidden=64, device=None, use_graph=True):
        super().__init__(env, n_steps=n_steps, learning_rate=learning_rate, gamma=gamma, gae_lambda=gae_lambda, ent_coef=ent_coef,
                         vf_coef=vf_coef, max_grad_norm=max_grad_norm, seed=seed, hidden=hidden, device=device, use_graph=use_graph)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=learning_rate, eps=1e-5)
        total = n_steps * env.N
        self.batch_size = max(1, total // 4) if batch_size is None else int(batch_size)
        if self.batch_size > total:
            raise ValueError("batch_size %d exceeds the rollout buffer (n_steps x lanes = %d)" % (self.batch_size, total))
        self.n_epochs, self.clip_range, self.normalize_advantage = n_epochs, clip_range, normalize_advantage
        # minibatch order: the same on every rank is fine (each rank shuffles its own lanes), but it must not be the sampling stream
        self.perm_gen = torch.Generator(device=self.device)
        self.perm_gen.manual_seed((0 if seed is None else int(seed)) * 31 + 17)

    def ppo_loss(self, boards, dices, acts, old_logp, adv, ret):
        """(loss, policy_loss, value_loss, entropy, clip_fraction) of one minibatch -- SB3 PPO.train's inner body"""
        logp, ent, value = self.model.evaluate_actions(boards, dices, acts)
        if self.normalize_advantage and adv.numel() > 1:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        ratio = torch.exp(logp - old_logp)
        policy_loss = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1.0 - self.clip_range, 1.0 + self.clip_range)).mean()
        value_loss = torch.nn.functional.mse_loss(ret, value)
        entropy_loss = -ent.mean()
        loss = policy_loss + self.ent_coef * entropy_loss + self.vf_coef * value_loss
        clip_frac = ((ratio - 1.0).abs() > self.clip_range).float().mean()
        return loss, policy_loss, value_loss, -entropy_loss, clip_frac

    def collect_and_update(self):
        env, T, N = self.env, self.n_steps, self.env.N
        self._collect()   # the n-step rollout, one hipGraph replay after the first (A2CTrainer._collect)
        boards = self._boards.reshape(T * N, env.S, env.S)
        dices, acts = self._dices.reshape(T * N), self._acts.reshape(T * N, 2)
        with torch.no_grad():
            _, _, last_value = self.model(env.board, env.dice)
            adv, ret = n_step_returns(self._rews, self._vals, self._dones, last_value, self.gamma, self.gae_lambda)
            old_logp, _, _ = self.model.evaluate_actions(boards, dices, acts)   # the behaviour policy's log-probabilities
        adv, ret = adv.reshape(-1), ret.reshape(-1)
        params = list(self.model.parameters())
        last = None
        for _ in range(self.n_epochs):
            perm = torch.randperm(T * N, device=self.device, generator=self.perm_gen)
            for lo in range(0, T * N - self.batch_size + 1, self.batch_size):   # whole minibatches only (SB3 drops nothing: its sizes divide)
                idx = perm[lo:lo + self.batch_size]
                loss, pl, vl, en, cf = self.ppo_loss(boards[idx], dices[idx], acts[idx], old_logp[idx], adv[idx], ret[idx])
                self.opt.zero_grad(set_to_none=False)
                loss.backward()
                all_reduce_gradients(params)
                nn.utils.clip_grad_norm_(params, self.max_grad_norm)
                self.opt.step()
                last = (loss, pl, vl, en)
        self.num_timesteps += T * N
        stats = torch.stack([last[0].detach(), last[1].detach(), last[2].detach(), last[3].detach(), self._rews.mean(), self._dones.sum()])
        self._last_stats = stats
        return stats
