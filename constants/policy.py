"""ClassicalPolicy enum with the reference's members and from_string quirk
(constants/policy.py:4-20): an unknown name comes back as the raw string, which the
reference env then treats as an SB3 checkpoint path (envs/ewn.py:285-287)."""
from enum import Enum


class ClassicalPolicy(Enum):
    random = "random"
    minimax = "minimax"
    uct = "uct"
    alpha_zero = "alpha_zero"
    mcts = "mcts"

    def __str__(self):
        return self.value

    @staticmethod
    def from_string(s):
        try:
            return ClassicalPolicy[s]
        except KeyError:
            return s
