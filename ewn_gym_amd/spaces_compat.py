"""Minimal stand-ins for the four gymnasium spaces the reference env constructs
(envs/ewn.py:61-69), used only when `gymnasium` is not importable.  Plumbing: no game
logic depends on them (the env draws dice from its own stream, never from a space)."""
import numpy as np


class Space:
    def __init__(self):
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]


class MultiDiscrete(Space):
    def __init__(self, nvec):
        super().__init__()
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.int64

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.nvec.shape and bool(((0 <= x) & (x < self.nvec)).all())


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__()
        self.n, self.start = int(n), int(start)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(self.start + self._rng.integers(self.n))

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n


class Box(Space):
    def __init__(self, low, high, shape, dtype=np.float32):
        super().__init__()
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    def sample(self):
        return self._rng.integers(self.low, self.high + 1).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(((self.low <= x) & (x <= self.high)).all())


class Dict(Space):
    def __init__(self, spaces):
        super().__init__()
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def seed(self, seed=None):
        return [s.seed(seed) for s in self.spaces.values()]

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def contains(self, x):
        return all(k in x and s.contains(x[k]) for k, s in self.spaces.items())


class DependencyNotInstalled(Exception):
    pass
