"""stable_baselines3.VecEnv adapter over VecEWN: what `train.py` would hand to
`A2C("MultiInputPolicy", env, ...)` instead of `SubprocVecEnv([make_env] * num_envs)` (train.py:134).

SB3 is not installed in this image, so the class subclasses `stable_baselines3.common.vec_env.VecEnv`
when it is importable and a structural stand-in otherwise; either way it follows the VecEnv contract:
numpy observations `{"board": int16 [N,S,S], "dice_roll": int64 [N]}`, `step_async/step_wait`,
auto-reset with `infos[i]["terminal_observation"]` and `infos[i]["TimeLimit.truncated"]`.
Note the host round trip this API forces per step (1.7 MB at 65 536 lanes): the on-device trainer
(`ewn_gym_amd.a2c`) is the fast path; this adapter is for drop-in compatibility.
"""
import numpy as np

from . import _lib

try:  # pragma: no cover - depends on the image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
    from gymnasium import spaces as _spaces
except ImportError:
    from . import spaces_compat as _spaces

    class _VecEnvBase:  # the subset of the VecEnv surface SB3's on-policy algorithms touch
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
            self.render_mode = None

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()


class EWNVecEnv(_VecEnvBase):
    def __init__(self, vec_env):
        self.env = vec_env
        if not vec_env.cfg.autoreset or vec_env.terminal_board is None:
            raise _lib.EwnError("EWNVecEnv needs VecEWN(autoreset=True, want_terminal_obs=True)")
        S, cn = vec_env.S, vec_env.cube_num
        obs_space = _spaces.Dict({"board": _spaces.Box(low=-cn, high=cn, shape=(S, S), dtype=np.int16),
                                  "dice_roll": _spaces.Discrete(cn + 1, start=1)})     # envs/ewn.py:63-69
        super().__init__(vec_env.N, obs_space, _spaces.MultiDiscrete([2, 3]))
        self._actions = None
        self._seeds = None

    def _obs(self):
        return {"board": self.env.board.cpu().numpy().astype(np.int16), "dice_roll": self.env.dice.cpu().numpy().astype(np.int64)}

    def seed(self, seed=None):
        base = 0 if seed is None else int(seed)
        self._seeds = (np.arange(self.num_envs, dtype=np.uint64) + base).astype(np.uint32)   # SB3: env i gets seed + i
        return list(self._seeds)

    def reset(self):
        if self._seeds is None:
            self.seed(int.from_bytes(np.random.bytes(4), "little"))
        self.env.reset(seeds=self._seeds)
        self._seeds = None
        return self._obs()

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.int8).reshape(self.num_envs, 2)

    def step_wait(self):
        _, _, r, te, tr, info = self.env.step(self._actions)
        obs = self._obs()
        rewards = r.cpu().numpy().astype(np.float32)
        dones = te.cpu().numpy().astype(bool)
        trunc = tr.cpu().numpy().astype(bool)
        codes = info.cpu().numpy()
        infos = [{} for _ in range(self.num_envs)]
        idx = np.nonzero(dones)[0]
        if idx.size:
            tb = self.env.terminal_board.cpu().numpy().astype(np.int16)
            td = self.env.terminal_dice.cpu().numpy().astype(np.int64)
            for i in idx:
                infos[i]["terminal_observation"] = {"board": tb[i], "dice_roll": td[i]}
                infos[i]["TimeLimit.truncated"] = bool(trunc[i] and not dones[i])   # always False: truncation comes with termination here
        for i in np.nonzero(codes)[0]:
            msg = _lib.INFO_MESSAGES[int(codes[i])]
            infos[i]["message"] = msg.format(int(self.env.tolerance[i].item())) if codes[i] == 5 else msg
        return obs, rewards, dones, infos

    def close(self):
        pass

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * self.num_envs

    def set_attr(self, attr_name, value, indices=None):
        raise NotImplementedError

    def env_method(self, method_name, *args, indices=None, **kwargs):
        raise NotImplementedError

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs
