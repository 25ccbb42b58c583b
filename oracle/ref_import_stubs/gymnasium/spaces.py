class _Space:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k

    def seed(self, seed=None):
        return [seed]


class MultiDiscrete(_Space):
    pass


class Box(_Space):
    pass


class Discrete(_Space):
    pass


class Dict(_Space):
    def __init__(self, d=None, **k):
        super().__init__(d, **k)
        self.spaces = dict(d or {})

    def __getitem__(self, k):
        return self.spaces[k]
