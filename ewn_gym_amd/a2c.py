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


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b whose weight gradient dW = dy^T x is computed as P partial products over slices of the batch.

    The training batch is n_steps * lanes = 3 x 10^5 rows against 64-wide layers: dy^T x is a [64 x B] x [B x 64] product,
    all reduction and a single output tile, which the library GEMM runs on a handful of CUs (rocprofv3: 0.43-0.54 ms each,
    six per update = half of the update).  As a batched product over P slices it fills the chip (one bmm + one sum)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B = x.shape[0]
        P = 1
        while P < 512 and B % (2 * P) == 0 and B // (2 * P) >= 256:
            P *= 2
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        if P > 1:
            dw = torch.bmm(dy.reshape(P, B // P, -1).transpose(1, 2), x.reshape(P, B // P, -1)).sum(0)
        else:
            dw = dy.t() @ x
        return dx, dw, dy.sum(0)


class SplitKLinear(nn.Linear):
    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 4096 and torch.is_grad_enabled():
            return _SplitKLinearFn.apply(x.contiguous(), self.weight, self.bias)
        return super().forward(x)


class ActorCritic(nn.Module):
    def __init__(self, board_size=5, cube_num=6, hidden=64):
        super().__init__()
        self.S, self.cube_num = board_size, cube_num
        feat = board_size * board_size + cube_num + 1
        self.pi = nn.Sequential(SplitKLinear(feat, hidden), nn.Tanh(), SplitKLinear(hidden, hidden), nn.Tanh())
        self.vf = nn.Sequential(SplitKLinear(feat, hidden), nn.Tanh(), SplitKLinear(hidden, hidden), nn.Tanh())
        self.action_net = SplitKLinear(hidden, 5)   # logits of MultiDiscrete([2, 3])
        self.value_net = SplitKLinear(hidden, 1)
        for seq in (self.pi, self.vf):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=math.sqrt(2))
                    nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.action_net.weight, gain=0.01)
        nn.init.zeros_(self.action_net.bias)
        nn.init.orthogonal_(self.value_net.weight, gain=1.0)
        nn.init.zeros_(self.value_net.bias)

    def flat_parameters(self):
        """the parameters as ONE fp32 vector in the order the HIP kernels read them (include/ewn_hip.h, struct ewn_policy): body pi
        (W1, b1, W2, b2), body vf, action head (W, b), value head (W, b) -- the order of .parameters()"""
        return torch.cat([p.detach().reshape(-1) for p in self.parameters()]).to(torch.float32).contiguous()

    def load_flat_parameters(self, flat):
        off = 0
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        assert off == flat.numel()

    def features(self, board, dice):
        """board int8 [N,S,S], dice int8 [N] (1..cube_num) -> [N, S*S + cube_num + 1] in the parameters' dtype (float32; float64 when
        the model was cast for a reference evaluation, tools/a2c_accuracy.py)"""
        dt = self.value_net.weight.dtype
        b = board.reshape(board.shape[0], -1).to(dt)
        oh = torch.nn.functional.one_hot((dice.to(torch.int64) - 1).clamp_(0, self.cube_num), self.cube_num + 1).to(dt)
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
            # Gumbel-max: argmax(logits - log(-log u)) ~ Categorical(softmax(logits)); one rand call, no sync,
            # capturable in a hipGraph (torch.multinomial is not)
            u = torch.rand(l0.shape[0], 5, device=l0.device, generator=generator).clamp_(1e-20, 1.0)
            g = -torch.log(-torch.log(u))
            a0, a1 = (l0 + g[:, :2]).argmax(1), (l1 + g[:, 2:]).argmax(1)
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


def all_reduce_gradients(params, world_size=None, force=False):
    """ONE collective per update: flatten every gradient into a single bucket, all-reduce (sum), divide, scatter back.
    force: issue the collective even in a one-rank group (exercises the RCCL path on a one-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
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
                 max_grad_norm=0.5, seed=None, hidden=64, device=None, use_graph=True):
        self.env = env
        self.device = torch.device(device) if device is not None else env.board.device
        if seed is not None:
            torch.manual_seed(seed)
        self.model = ActorCritic(env.S, env.cube_num, hidden).to(self.device)
        self._sync_parameters()
        self.opt = torch.optim.RMSprop(self.model.parameters(), lr=learning_rate, alpha=0.99, eps=1e-5)
        self.n_steps, self.gamma, self.gae_lambda = n_steps, gamma, gae_lambda
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        # the sampling noise must differ between ranks (each owns different lanes): the generator is keyed by the global id
        # of this rank's first lane, NOT by the seed alone (which seeds the identical parameters on every rank)
        self.gen = torch.Generator(device=self.device)
        cfg = getattr(env, "cfg", None)
        self.lane_offset = int(getattr(cfg, "lane_offset", getattr(env, "lane_offset", 0)))
        self.gen.manual_seed((0 if seed is None else int(seed)) + 7919 * (self.lane_offset + 1))
        self.num_timesteps = 0
        # rollout storage, allocated once: [T, N, ...]
        T, N, S = n_steps, env.N, env.S
        dev = self.device
        self._boards = torch.zeros((T, N, S, S), dtype=torch.int8, device=dev)
        self._dices = torch.zeros((T, N), dtype=torch.int8, device=dev)
        self._acts = torch.zeros((T, N, 2), dtype=torch.int8, device=dev)
        self._vals = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self._rews = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self._dones = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self._graph = None
        self._warm = False
        self.use_graph = use_graph and self.device.type == "cuda"

    def _rollout(self):
        """n_steps of (observe, act, env.step) into the preallocated buffers; no host sync, no allocation that outlives it"""
        env = self.env
        for t in range(self.n_steps):
            self._boards[t].copy_(env.board)
            self._dices[t].copy_(env.dice)
            a, v = self.model.act(env.board, env.dice, generator=self.gen)
            self._acts[t].copy_(a)
            self._vals[t].copy_(v)
            _, _, r, term, _, _ = env.step(self._acts[t])
            self._rews[t].copy_(r)
            self._dones[t].copy_(term)

    def _sync_parameters(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)

    def _collect(self):
        """one n-step rollout into the preallocated buffers: eagerly the first time, as ONE hipGraph replay afterwards"""
        if self.use_graph and self._warm:
            # the whole n-step rollout (policy forward, sampling, ewn_step, bookkeeping) is one captured hipGraph: the
            # policy net is ~13 k parameters, so un-captured the loop is pure launch overhead (~25 tiny kernels per step).
            # The FIRST rollout ran eagerly (it is the warm-up capture needs, and its transitions are trained on like any
            # other: no env step is thrown away); capturing executes nothing.
            if self._graph is None:
                torch.cuda.synchronize()
                self._graph = torch.cuda.CUDAGraph()
                self._graph.register_generator_state(self.gen)
                with torch.cuda.graph(self._graph):
                    self._rollout()
            self._graph.replay()
        else:
            self._rollout()
            self._warm = True

    def collect_and_update(self):
        env, T, N = self.env, self.n_steps, self.env.N
        self._collect()
        with torch.no_grad():
            _, _, last_value = self.model(env.board, env.dice)
        rews, dones, vals = self._rews, self._dones, self._vals
        adv, ret = n_step_returns(rews, vals, dones, last_value, self.gamma, self.gae_lambda)
        logp, ent, value = self.model.evaluate_actions(self._boards.reshape(T * N, env.S, env.S), self._dices.reshape(T * N),
                                                       self._acts.reshape(T * N, 2))
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
        stats = torch.stack([loss.detach(), policy_loss.detach(), value_loss.detach(), -entropy_loss.detach(), rews.mean(), dones.sum()])
        self._last_stats = stats   # read lazily: a .item() here would stall the stream every update
        return stats

    def stats_dict(self, stats=None):
        s = (self._last_stats if stats is None else stats).tolist()
        return {"loss": s[0], "policy_loss": s[1], "value_loss": s[2], "entropy": s[3], "mean_reward": s[4], "episodes": int(s[5])}

    def learn(self, total_timesteps):
        target = self.num_timesteps + total_timesteps
        while self.num_timesteps < target:
            self.collect_and_update()
        return self.stats_dict()

    def policy_fn(self, deterministic=True):
        """A batched policy for tournament.evaluate / hand-rolled loops: (board, dice, t) -> int8 [N,2]"""
        return lambda b, d, t: self.model.act(b, d, deterministic=deterministic, generator=self.gen)[0]

    algorithm = "A2C"
    best_score = -1.0     # best evaluation win rate seen so far (train.py:109-112); travels with the checkpoint

    def save(self, path):
        torch.save({"algorithm": self.algorithm, "model": self.model.state_dict(), "opt": self.opt.state_dict(),
                    "num_timesteps": self.num_timesteps, "best_score": float(self.best_score),
                    "generator": self.gen.get_state().cpu()}, path)

    def load(self, path):
        sd = torch.load(path, map_location=self.device, weights_only=True)
        algo = sd.get("algorithm", "A2C")
        if algo != self.algorithm:   # an RMSprop state dict in PPO's Adam (or the reverse) would mis-load or fail obscurely
            raise ValueError("checkpoint %s was written by the %s trainer, this is the %s trainer" % (path, algo, self.algorithm))
        self.model.load_state_dict(sd["model"])
        self.opt.load_state_dict(sd["opt"])
        self.num_timesteps = sd["num_timesteps"]
        self.best_score = float(sd.get("best_score", -1.0))
        if "generator" in sd:        # the sampling noise continues where the saved run stopped
            self.gen.set_state(sd["generator"].cpu())
        self._graph = None           # a captured rollout holds the old generator registration
        self._warm = False


class FusedA2CTrainer:
    """A2C with the whole loop in the engine (BASELINE config 4): the n-step rollout is ONE kernel (ewn_step_k_policy: policy
    network on the matrix cores, Gumbel-max sampling, shaped env step, opponent search, records), the update three more
    (ewn_a2c_grad: value pass, policy pass, reduction; ewn_a2c_apply: global-norm clip + RMSprop) -- no torch operator on the
    training path, one all-reduce of the flat gradient between grad and apply when there are several ranks.  Same loss and
    optimiser as A2CTrainer (SB3's documented A2C defaults, gae_lambda = 1); the parameters live in ONE flat fp32 tensor that the
    torch module `self.model` views (evaluation, checkpoints)."""

    algorithm = "A2C"
    best_score = -1.0

    def __init__(self, env, n_steps=5, learning_rate=7e-4, gamma=0.99, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, seed=None,
                 rms_alpha=0.99, rms_eps=1e-5, use_graph=True):
        import ctypes as C
        from . import _lib
        if not env.supports_policy_rollout():
            raise _lib.EwnError("this env configuration has no policy-driven rollout kernel (ewn_step_k_policy): use A2CTrainer")
        self.env, self.lib, self.C = env, env.lib, C
        self.device = env.board.device
        if seed is not None:
            torch.manual_seed(seed)
        self.model = ActorCritic(env.S, env.cube_num).to(self.device)
        self.params = self.model.flat_parameters()                    # the flat vector the kernels read and update in place
        off = 0
        for p in self.model.parameters():                             # ... and the module's parameters become views of it
            p.data = self.params[off:off + p.numel()].view_as(p)
            off += p.numel()
        assert off == env.policy_param_count()
        self._sync_parameters()
        self.sq_avg = torch.zeros_like(self.params)
        self.grad = torch.zeros(self.params.numel() + 8, dtype=torch.float32, device=self.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.n_steps, self.num_timesteps = int(n_steps), 0
        world = 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size()
        self.world = world
        self.hyper = _lib.EwnA2cHyper(float(gamma), float(vf_coef), float(ent_coef), float(max_grad_norm), float(learning_rate),
                                      float(rms_alpha), float(rms_eps), int(world))
        nscr = _lib.check(self.lib.ewn_a2c_scratch_bytes(C.byref(env.cfg), self.n_steps), "ewn_a2c_scratch_bytes")
        self.scratch = torch.zeros(int(nscr), dtype=torch.uint8, device=self.device)
        self.traj = env.alloc_rollout(self.n_steps, layout="record", initial_obs=True)
        self.noise_key = (0 if seed is None else int(seed)) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF
        self.gen = torch.Generator(device=self.device)                # policy_fn's sampling only (evaluation is deterministic)
        self.gen.manual_seed(0 if seed is None else int(seed))
        self.force_collective = False     # tools/nccl_selftest.py: all-reduce the gradient even in a one-rank group
        self.use_graph = use_graph and world == 1
        self._graph = None
        self._warm = False

    _sync_parameters = A2CTrainer._sync_parameters

    def _launch(self):
        from ._lib import check
        from .vec_env import _ptr, _stream
        C, env = self.C, self.env
        env.rollout_policy(self.n_steps, self.params, traj=self.traj, noise_key=self.noise_key)
        check(self.lib.ewn_a2c_grad(C.byref(env.cfg), self.n_steps, _ptr(self.traj["record"]), _ptr(self.traj["reward"]), _ptr(self.params),
                                    C.byref(self.hyper), _ptr(self.grad), _ptr(self.scratch), _stream()), "ewn_a2c_grad")
        if self.world > 1 or self.force_collective:   # the one collective of the training path: the flat gradient (52 KB), summed; apply divides by the world size
            import torch.distributed as dist
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        check(self.lib.ewn_a2c_apply(C.byref(env.cfg), _ptr(self.params), _ptr(self.sq_avg), _ptr(self.grad), C.byref(self.hyper),
                                     _ptr(self.grad_norm), _stream()), "ewn_a2c_apply")

    def collect_and_update(self):
        if self.use_graph and self._warm:
            if self._graph is None:       # five kernel launches, no torch operator: captured once, replayed per update
                torch.cuda.synchronize()
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._launch()
            self._graph.replay()
        else:
            self._launch()
            self._warm = True
        self.num_timesteps += self.n_steps * self.env.N
        return self.grad

    def stats_dict(self):
        g = self.grad[-8:].tolist()
        n = float(self.n_steps * self.env.N * self.world)
        pl, vl, en = g[0] / n, g[5] / n, g[2] / n
        return {"loss": pl + self.hyper.vf_coef * vl - self.hyper.ent_coef * en, "policy_loss": pl, "value_loss": vl, "entropy": en,
                "mean_reward": float(self.traj["reward"].mean()), "episodes": int(self.traj["terminated"].sum()),
                "grad_norm": float(self.grad_norm)}

    def learn(self, total_timesteps):
        target = self.num_timesteps + total_timesteps
        while self.num_timesteps < target:
            self.collect_and_update()
        return self.stats_dict()

    def policy_fn(self, deterministic=True):
        return lambda b, d, t: self.model.act(b, d, deterministic=deterministic, generator=self.gen)[0]

    def save(self, path):
        torch.save({"algorithm": self.algorithm, "fused": True, "params": self.params, "sq_avg": self.sq_avg,
                    "num_timesteps": self.num_timesteps, "best_score": float(self.best_score)}, path)

    def load(self, path):
        sd = torch.load(path, map_location=self.device, weights_only=True)
        if sd.get("algorithm", "A2C") != self.algorithm or not sd.get("fused", False):
            raise ValueError("checkpoint %s was not written by the fused A2C trainer" % path)
        self.params.copy_(sd["params"])
        self.sq_avg.copy_(sd["sq_avg"])
        self.num_timesteps = sd["num_timesteps"]
        self.best_score = float(sd.get("best_score", -1.0))

