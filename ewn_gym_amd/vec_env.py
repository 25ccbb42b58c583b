"""VecEWN: N independent EinStein-wuerfelt-nicht games stepped on one MI355X.

Host side is plumbing only: torch owns the device buffers and the stream, every
rule, dice draw and search runs in libewn_hip.so (include/ewn_hip.h).  The class is
the batched counterpart of the reference's EinsteinWuerfeltNichtEnv /
MiniMaxHeuristicEnv (envs/ewn.py, envs/training_ewn.py): the single-game drop-in
classes in the top-level `envs` package are N=1 views of it.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AGENT, HEUR, OPP, RNG, EwnConfig, EwnPolicy, EwnRolloutOut, EwnState, EwnStepOut, check


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise _lib.EwnError("ewn_gym_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device(device)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_TABLES = {}


def search_tables(board_size, cube_layer, device):
    """Device copy of the depth-3 search tables (built on the host by ewn_build_tables), or None."""
    key = (int(board_size), int(cube_layer), str(device))
    if key not in _TABLES:
        lib = _lib.load()
        n = int(lib.ewn_tables_bytes(key[0], key[1]))
        t = None
        if n > 0:
            host = np.zeros(n, dtype=np.uint8)
            check(lib.ewn_build_tables(key[0], key[1], host.ctypes.data_as(C.c_void_p)), "ewn_build_tables")
            t = torch.zeros(n, dtype=torch.uint8, device=device)   # torch.zeros: the guard-zone test wraps this allocator
            t.copy_(torch.from_numpy(host))
        _TABLES[key] = t
    return _TABLES[key]


class VecEWN:
    def __init__(self, n_lanes, board_size=5, cube_layer=3, opponent_policy="random", max_depth=3, heuristic="hybrid",
                 num_simulations=10, num_env_copies=5, rng="mt19937", shaped=False, reward=1.0,
                 illegal_move_reward=-1.0, illegal_move_tolerance=10, autoreset=False, shaped_refresh_on_reset=False,
                 lane_offset=0, seed_stride=None, philox_key=0, mt_window=0, want_terminal_obs=False, device="cuda",
                 use_tables=True, want_random_action=False):
        self.lib = _lib.load()
        opp = str(opponent_policy)
        if opp not in OPP:
            raise _lib.EwnError("opponent policy %r is not supported by the HIP engine (random, minimax, mcts)" % opp)
        if heuristic not in HEUR:
            raise _lib.EwnError("heuristic %r is not supported (hybrid, min_dist, two_min_dist, attk, sim_winrate)" % heuristic)
        self.N, self.S, self.L = int(n_lanes), int(board_size), int(cube_layer)
        self.cube_num = self.L * (self.L + 1) // 2
        self.cfg = EwnConfig(self.S, self.L, self.N, OPP[opp], int(max_depth), HEUR[heuristic], int(num_simulations),
                             int(num_env_copies), RNG[rng], int(bool(shaped)), int(illegal_move_tolerance),
                             int(bool(autoreset)), int(bool(shaped_refresh_on_reset)), int(lane_offset),
                             (self.N if seed_stride is None else int(seed_stride)) & 0xFFFFFFFF, int(mt_window),
                             float(reward), float(illegal_move_reward), int(philox_key) & 0xFFFFFFFFFFFFFFFF)
        words = check(self.lib.ewn_rng_words(C.byref(self.cfg)), "ewn_rng_words")  # validates the whole config
        self.device = _require_gpu(device)
        dev, N, S = self.device, self.N, self.S
        self.board = torch.zeros((N, S, S), dtype=torch.int8, device=dev)
        self.dice = torch.zeros(N, dtype=torch.int8, device=dev)
        self.done = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.rng_state = torch.zeros((N, words), dtype=torch.int32, device=dev)
        self.prev_score = torch.zeros(N, dtype=torch.float64, device=dev) if shaped else None
        self.tolerance = torch.zeros(N, dtype=torch.int32, device=dev) if shaped else None
        self.reward = torch.zeros(N, dtype=torch.float64, device=dev)
        self.terminated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.info = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.terminal_board = torch.zeros((N, S, S), dtype=torch.int8, device=dev) if want_terminal_obs else None
        self.terminal_dice = torch.zeros(N, dtype=torch.int8, device=dev) if want_terminal_obs else None
        # RandomAgent.predict on the post-step observation, fused into ewn_step (ewn_step_out.random_action)
        self.random_action = torch.zeros((N, 2), dtype=torch.int8, device=dev) if want_random_action else None
        nscr = check(self.lib.ewn_step_scratch_bytes(C.byref(self.cfg)), "ewn_step_scratch_bytes")
        self.scratch = torch.zeros(max(int(nscr), 8), dtype=torch.uint8, device=dev)
        self._actions = torch.zeros((N, 2), dtype=torch.int8, device=dev)
        self.tables = search_tables(self.S, self.L, dev) if use_tables else None
        self._st = EwnState(_ptr(self.board), _ptr(self.dice), _ptr(self.done), _ptr(self.rng_state),
                            _ptr(self.prev_score), _ptr(self.tolerance), _ptr(self.tables))
        self._out = EwnStepOut(_ptr(self.reward), _ptr(self.terminated), _ptr(self.truncated), _ptr(self.info),
                               _ptr(self.terminal_board), _ptr(self.terminal_dice), _ptr(self.random_action))
        check(self.lib.ewn_init_aux(C.byref(self.cfg), C.byref(self._st), _stream()), "ewn_init_aux")

    # -- reset(seed) for the lanes selected by mask (envs/ewn.py:488-494)
    def reset(self, seeds=None, mask=None):
        if seeds is not None:
            if not isinstance(seeds, torch.Tensor):  # uint32 seeds: keep the bit pattern (np.random.seed takes 0..2**32-1)
                seeds = torch.from_numpy(np.asarray(seeds, dtype=np.uint32).reshape(-1).view(np.int32).copy())
            seeds = seeds.to(self.device).to(torch.int32).contiguous()
            assert seeds.numel() == self.N
        if mask is not None:
            mask = torch.as_tensor(mask).to(self.device).to(torch.uint8).contiguous()
            assert mask.numel() == self.N
        check(self.lib.ewn_reset(C.byref(self.cfg), C.byref(self._st), _ptr(seeds), _ptr(mask), _stream()), "ewn_reset")
        return self.board, self.dice

    # -- roll_dice() (envs/ewn.py:90-92): one draw from each selected lane's dice stream
    def roll_dice(self, mask=None):
        if mask is not None:
            mask = torch.as_tensor(mask).to(self.device).to(torch.uint8).contiguous()
            assert mask.numel() == self.N
        check(self.lib.ewn_roll_dice(C.byref(self.cfg), C.byref(self._st), _ptr(mask), _stream()), "ewn_roll_dice")
        return self.dice

    # -- step(actions) (envs/ewn.py:436-486 / envs/training_ewn.py:43-99)
    def step(self, actions):
        if not (isinstance(actions, torch.Tensor) and actions.dtype == torch.int8 and actions.is_cuda and actions.is_contiguous()):
            actions = self._actions.copy_(torch.as_tensor(actions).reshape(self.N, 2))
        assert actions.numel() == 2 * self.N
        check(self.lib.ewn_step(C.byref(self.cfg), C.byref(self._st), _ptr(actions), C.byref(self._out), _ptr(self.scratch),
                                _stream()), "ewn_step")
        return self.board, self.dice, self.reward, self.terminated, self.truncated, self.info

    # -- RandomAgent as a stateless device policy (classical_policies/random_policy.py:11-15)
    def sample_legal_actions(self, step, out=None, step_tensor=None):
        """step_tensor: optional int32 device scalar added to `step` on the device (lets a captured graph advance)"""
        out = self._actions if out is None else out
        check(self.lib.ewn_predict_random(self.S, self.L, self.N, _ptr(self.board), _ptr(self.dice),
                                          C.c_uint64(self.cfg.philox_key), C.c_uint32(int(step) & 0xFFFFFFFF),
                                          _ptr(step_tensor), int(self.cfg.lane_offset), _ptr(out), _stream()),
              "ewn_predict_random")
        return out

    def rng_overflow(self):
        """uint8 [N]: lanes whose current episode has consumed more MT19937 draws than the closed form covers (454; see
        DESIGN.md section 3) -- from then on their dice are NOT numpy's.  Always zero for the Philox kind."""
        if self.cfg.rng_kind != RNG["mt19937"]:
            return torch.zeros(self.N, dtype=torch.uint8, device=self.device)
        return (self.rng_state.view(-1)[3:4 * self.N:4] & 1).to(torch.uint8)

    def check_rng(self):
        """Raise if any lane ran past the MT19937-compat stream (one device reduction + sync: call it where a host sync
        happens anyway -- the N=1 drop-in env does so every step, tournament.evaluate once per evaluation)."""
        if self.cfg.rng_kind == RNG["mt19937"] and bool(self.rng_overflow().any()):
            lanes = torch.nonzero(self.rng_overflow()).reshape(-1)[:8].tolist()
            raise _lib.EwnError("MT19937-compat dice stream exhausted (>= 454 draws in one episode) on lanes %s: results are no "
                                "longer bit-identical to numpy's stream; use rng='philox' for such long games" % lanes)

    # -- K env steps per launch with an in-engine agent (ewn_step_k; eval_minimax.py:16-50's loop on the device)
    def supports_rollout(self, agent="random", agent_max_depth=3):
        """True when ewn_step_k exists for this configuration and agent"""
        if agent not in AGENT:
            return False
        if self.tables is None and int(self.lib.ewn_tables_bytes(self.S, self.L)) > 0:
            return False           # use_tables=False on a table geometry (geometries WITHOUT a table image run the generic K-step kernel)
        return self.lib.ewn_step_k_supported(C.byref(self.cfg), AGENT[agent], int(agent_max_depth)) == 1

    def alloc_rollout(self, K, board=True, layout="columns", initial_obs=False):
        """Trajectory buffers for rollout(K, traj=...): dict of [K, N, ...] tensors (the observation column is optional).
        layout="record": ONE 16-byte aligned record per lane-step (ewn_rollout_out.record: board | dice | action | terminated |
        truncated | info | padding; 32 bytes for 5x5) plus the f64 reward column; the dict's board / dice / action / flag entries
        are then strided VIEWS into the records."""
        dev, N, S = self.device, self.N, self.S
        if layout == "record":
            C2 = S * S
            stride = (C2 + 6 + 15) & ~15
            # initial_obs (rollout_policy(record_initial_obs=True)): one more row in front, the observation before step 0; the
            # per-step views below then start at row 1 and "obs_board" / "obs_dice" are the K + 1 observations s_0 .. s_K
            full = torch.zeros((K + (1 if initial_obs else 0), N, stride), dtype=torch.uint8, device=dev)
            rec = full[1:] if initial_obs else full
            t = {"record": full, "reward": torch.zeros((K, N), dtype=torch.float64, device=dev),
                 "board": rec[:, :, :C2].view(torch.int8).unflatten(2, (S, S)), "dice": rec[:, :, C2].view(torch.int8),
                 "action": rec[:, :, C2 + 1:C2 + 3].view(torch.int8), "terminated": rec[:, :, C2 + 3],
                 "truncated": rec[:, :, C2 + 4], "info": rec[:, :, C2 + 5]}
            if initial_obs:
                t["obs_board"] = full[:, :, :C2].view(torch.int8).unflatten(2, (S, S))
                t["obs_dice"] = full[:, :, C2].view(torch.int8)
            return t
        assert layout == "columns"
        t = {"dice": torch.zeros((K, N), dtype=torch.int8, device=dev),
             "action": torch.zeros((K, N, 2), dtype=torch.int8, device=dev),
             "reward": torch.zeros((K, N), dtype=torch.float64, device=dev),
             "terminated": torch.zeros((K, N), dtype=torch.uint8, device=dev),
             "truncated": torch.zeros((K, N), dtype=torch.uint8, device=dev),
             "info": torch.zeros((K, N), dtype=torch.uint8, device=dev)}
        if board:
            t["board"] = torch.zeros((K, N, S, S), dtype=torch.int8, device=dev)
        return t

    def alloc_totals(self):
        """Per-lane accumulators for rollout(..., totals=...): return_sum, n_steps, n_episodes, n_wins (rollout ADDS to them)"""
        dev, N = self.device, self.N
        return {"return_sum": torch.zeros(N, dtype=torch.float64, device=dev), "n_steps": torch.zeros(N, dtype=torch.int32, device=dev),
                "n_episodes": torch.zeros(N, dtype=torch.int32, device=dev), "n_wins": torch.zeros(N, dtype=torch.int32, device=dev)}

    def rollout(self, K, agent="random", agent_max_depth=3, traj=None, totals=None):
        """Play K steps of every lane in one launch, the agent being RandomAgent ("random"), ExpectiMinimaxAgent(agent_max_depth)
        ("minimax") or env.action_space.sample() ("sample": all six actions, illegal ones included).
        traj: dict from alloc_rollout (first dimension >= K) or None; totals: dict from alloc_totals or None."""
        traj, totals = traj or {}, totals or {}
        for v in traj.values():
            assert v.shape[0] >= K and v.shape[1] == self.N
        col = (lambda k: None) if "record" in traj else traj.get   # record layout: the column entries are views, not buffers
        out = EwnRolloutOut(_ptr(col("board")), _ptr(col("dice")), _ptr(col("action")), _ptr(traj.get("reward")),
                            _ptr(col("terminated")), _ptr(col("truncated")), _ptr(col("info")),
                            _ptr(totals.get("return_sum")), _ptr(totals.get("n_steps")), _ptr(totals.get("n_episodes")),
                            _ptr(totals.get("n_wins")), _ptr(traj.get("record")))
        check(self.lib.ewn_step_k(C.byref(self.cfg), C.byref(self._st), int(K), AGENT[agent], int(agent_max_depth), C.byref(out),
                                  _stream()), "ewn_step_k")
        return self.board, self.dice

    def bind_rollout(self, K, agent="random", agent_max_depth=3, traj=None, totals=None):
        """rollout(K, ...) with its arguments marshalled ONCE: returns a zero-argument callable that enqueues the launch (for loops that
        repeat the same call: one C-ABI call per invocation, a few microseconds of host time instead of the ~20 of building the structs).
        The buffers of traj / totals must stay alive as long as the callable is used."""
        traj, totals = traj or {}, totals or {}
        for v in traj.values():
            assert v.shape[0] >= K and v.shape[1] == self.N
        col = (lambda k: None) if "record" in traj else traj.get
        out = EwnRolloutOut(_ptr(col("board")), _ptr(col("dice")), _ptr(col("action")), _ptr(traj.get("reward")),
                            _ptr(col("terminated")), _ptr(col("truncated")), _ptr(col("info")),
                            _ptr(totals.get("return_sum")), _ptr(totals.get("n_steps")), _ptr(totals.get("n_episodes")),
                            _ptr(totals.get("n_wins")), _ptr(traj.get("record")))
        fn, cfg, st, outp = self.lib.ewn_step_k, C.byref(self.cfg), C.byref(self._st), C.byref(out)
        k, a, d = int(K), AGENT[agent], int(agent_max_depth)
        keep = (out, traj, totals)

        def call():
            rc = fn(cfg, st, k, a, d, outp, _stream())
            if rc != 0:
                check(rc, "ewn_step_k")
            return keep and None
        return call

    # -- K env steps per launch with the trained policy as the agent (ewn_step_k_policy; train.py:134, 148's rollout collection)
    def policy_param_count(self):
        return int(check(self.lib.ewn_policy_param_count(self.S, self.L), "ewn_policy_param_count"))

    def supports_policy_rollout(self):
        return self.tables is not None and self.lib.ewn_step_k_supported(C.byref(self.cfg), AGENT["mlp"], 0) == 1

    def rollout_policy(self, K, params, traj=None, totals=None, deterministic=False, noise_key=0, logits=None, value=None, noise=None):
        """Play K steps of every lane in one launch, actions sampled from the actor-critic whose flat fp32 parameter vector is
        `params` (a2c.ActorCritic.flat order).  traj: dict from alloc_rollout (a record layout allocated with initial_obs=True gets
        K + 1 rows); logits [K, N, 5] / value [K, N] / noise [K, N, 5] float32: optional per-step outputs of the policy."""
        traj, totals = traj or {}, totals or {}
        assert params.dtype == torch.float32 and params.is_contiguous() and params.numel() == self.policy_param_count()
        rec0 = 0
        if "record" in traj:
            rec0 = 1 if "obs_board" in traj else 0
            assert traj["record"].shape[0] >= K + rec0
        col = (lambda k: None) if "record" in traj else traj.get
        out = EwnRolloutOut(_ptr(col("board")), _ptr(col("dice")), _ptr(col("action")), _ptr(traj.get("reward")),
                            _ptr(col("terminated")), _ptr(col("truncated")), _ptr(col("info")),
                            _ptr(totals.get("return_sum")), _ptr(totals.get("n_steps")), _ptr(totals.get("n_episodes")),
                            _ptr(totals.get("n_wins")), _ptr(traj.get("record")))
        pol = EwnPolicy(_ptr(params), int(bool(deterministic)), rec0, int(noise_key) & 0xFFFFFFFFFFFFFFFF, _ptr(logits), _ptr(value), _ptr(noise))
        check(self.lib.ewn_step_k_policy(C.byref(self.cfg), C.byref(self._st), int(K), C.byref(pol), C.byref(out), _stream()),
              "ewn_step_k_policy")
        return self.board, self.dice

    def set_obs(self, boards, dice):
        """Overwrite the observation of every lane (agent = TOP_LEFT to move); RNG state is kept."""
        self.board.copy_(torch.as_tensor(boards).reshape(self.N, self.S, self.S))
        self.dice.copy_(torch.as_tensor(dice).reshape(self.N))
        self.done.zero_()

    def state_dict(self):
        """Checkpoint of the env (the reference never checkpoints env state)."""
        keys = ("board", "dice", "done", "rng_state", "scratch", "prev_score", "tolerance")  # scratch: the MT refill queue
        return {k: getattr(self, k).clone() for k in keys if getattr(self, k) is not None}

    def load_state_dict(self, sd):
        for k, v in sd.items():
            getattr(self, k).copy_(v)


# ---------------------------------------------------------------- stateless batched queries

def _prep(boards, dice=None, device="cuda"):
    dev = _require_gpu(device)
    b = torch.as_tensor(boards)
    if b.dim() == 2:
        b = b.unsqueeze(0)
    M, S = b.shape[0], b.shape[1]
    b = b.to(dev).to(torch.int8).contiguous()
    d = None
    if dice is not None:
        d = torch.as_tensor(dice).reshape(-1).to(dev).to(torch.int8).contiguous()
        assert d.numel() == M
    return b, d, M, S, dev


def legal_actions(boards, dice, player=1, cube_layer=3):
    """get_legal_actions / find_cube_to_move / check_win on M boards -> (acts[M,6,2], n[M], cube_small[M], cube_large[M], win[M])"""
    lib = _lib.load()
    b, d, M, S, dev = _prep(boards, dice)
    acts = torch.full((M, 6, 2), -1, dtype=torch.int8, device=dev)
    n = torch.zeros(M, dtype=torch.int8, device=dev)
    cs = torch.zeros(M, dtype=torch.int8, device=dev)
    cl = torch.zeros(M, dtype=torch.int8, device=dev)
    win = torch.zeros(M, dtype=torch.uint8, device=dev)
    check(lib.ewn_legal_actions(S, cube_layer, M, _ptr(b), _ptr(d), int(player), _ptr(acts), _ptr(n), _ptr(cs), _ptr(cl),
                                _ptr(win), _stream()), "ewn_legal_actions")
    return acts, n, cs, cl, win


def apply_action(boards, dice, actions, player=1, cube_layer=3):
    """make_simulated_action on M positions -> (new boards int8 [M,S,S], valid uint8 [M])"""
    lib = _lib.load()
    b, d, M, S, dev = _prep(boards, dice)
    a = torch.as_tensor(actions).reshape(M, 2).to(dev).to(torch.int8).contiguous()
    nb = torch.zeros_like(b)
    valid = torch.zeros(M, dtype=torch.uint8, device=dev)
    check(lib.ewn_apply_action(S, cube_layer, M, _ptr(b), _ptr(d), int(player), _ptr(a), _ptr(nb), _ptr(valid), _stream()),
          "ewn_apply_action")
    return nb, valid


def playout_wins(boards, first_player, n_sims=100, key=0, cube_layer=3):
    """MinimaxEnv.simulate on M positions -> int32 [M] playouts won by TOP_LEFT out of n_sims"""
    lib = _lib.load()
    b, _, M, S, dev = _prep(boards)
    wins = torch.zeros(M, dtype=torch.int32, device=dev)
    check(lib.ewn_playout_wins(S, cube_layer, M, _ptr(b), int(first_player), int(n_sims), C.c_uint64(key), _ptr(wins), _stream()),
          "ewn_playout_wins")
    return wins


def evaluate(boards, heuristic="hybrid", cube_layer=3):
    lib = _lib.load()
    if heuristic not in HEUR:
        raise _lib.EwnError("heuristic %r is not supported" % heuristic)
    b, _, M, S, dev = _prep(boards)
    out = torch.zeros(M, dtype=torch.float64, device=dev)
    check(lib.ewn_evaluate(S, cube_layer, M, _ptr(b), HEUR[heuristic], _ptr(out), _stream()), "ewn_evaluate")
    return out


def predict_minimax(boards, dice, max_depth, heuristic="hybrid", cube_layer=3, use_tables=True, key=0, obs_id=None):
    """ExpectiMinimaxAgent.predict on M observations -> (actions int8 [M,2], root values f64 [M]).  key / obs_id select the
    playouts' randomness of the 'sim_winrate' heuristic (ignored otherwise)."""
    lib = _lib.load()
    if heuristic not in HEUR:
        raise _lib.EwnError("heuristic %r is not supported" % heuristic)
    b, d, M, S, dev = _prep(boards, dice)
    acts = torch.zeros((M, 2), dtype=torch.int8, device=dev)
    vals = torch.zeros(M, dtype=torch.float64, device=dev)
    if heuristic == "sim_winrate":
        ids = None
        if obs_id is not None:
            ids = torch.from_numpy(np.asarray(obs_id, dtype=np.uint32).reshape(-1).view(np.int32).copy()).to(dev)
        check(lib.ewn_predict_minimax_sim(S, cube_layer, M, _ptr(b), _ptr(d), int(max_depth), C.c_uint64(key), _ptr(ids), _ptr(acts),
                                          _ptr(vals), _stream()), "ewn_predict_minimax_sim")
        return acts, vals
    tables = search_tables(S, cube_layer, dev) if use_tables else None
    check(lib.ewn_predict_minimax(S, cube_layer, M, _ptr(b), _ptr(d), int(max_depth), HEUR[heuristic], _ptr(acts),
                                  _ptr(vals), _ptr(tables), _stream()), "ewn_predict_minimax")
    return acts, vals


def predict_random(boards, dice, key=0, step=0, lane_offset=0, cube_layer=3):
    lib = _lib.load()
    b, d, M, S, dev = _prep(boards, dice)
    acts = torch.zeros((M, 2), dtype=torch.int8, device=dev)
    check(lib.ewn_predict_random(S, cube_layer, M, _ptr(b), _ptr(d), C.c_uint64(key), C.c_uint32(step), None,
                                 int(lane_offset), _ptr(acts), _stream()), "ewn_predict_random")
    return acts


def predict_mcts(boards, dice, num_simulations=10, num_env_copies=5, key=0, obs_id=None, cube_layer=3):
    lib = _lib.load()
    b, d, M, S, dev = _prep(boards, dice)
    acts = torch.zeros((M, 2), dtype=torch.int8, device=dev)
    wins = torch.zeros((M, 6), dtype=torch.int32, device=dev)
    ids = None
    if obs_id is not None:
        ids = torch.from_numpy(np.asarray(obs_id, dtype=np.uint32).reshape(-1).view(np.int32).copy()).to(dev)
    check(lib.ewn_predict_mcts(S, cube_layer, M, _ptr(b), _ptr(d), int(num_simulations), int(num_env_copies),
                               C.c_uint64(key), _ptr(ids), _ptr(acts), _ptr(wins), _stream()), "ewn_predict_mcts")
    return acts, wins
