"""Import stub, container-only: lets oracle/gen_golden.py import the reference's
pure-Python game logic where the third-party `gymnasium` package is not installed.
Supplies only the base class and the space containers the reference constructs
(envs/ewn.py:18,61-69); none of it is on the arithmetic path (the env draws dice
from the global numpy RNG, envs/ewn.py:91, never from a space)."""
from . import spaces, error  # noqa: F401


class Env:
    metadata = {}

    def __init__(self, *a, **k):
        pass
