"""Import stub, container-only (see gymnasium stub). The reference only calls
A2C.load for checkpoint-path opponents (envs/ewn.py:287), which the golden
generator never uses."""


class _Algo:
    @classmethod
    def load(cls, *a, **k):
        raise RuntimeError("stable_baselines3 is not installed (stub)")


class A2C(_Algo):
    pass


class PPO(_Algo):
    pass
