"""PolicyBase (classical_policies/base.py:6-9) plus the batched entry point every
policy here also offers."""
from abc import abstractmethod


class PolicyBase:
    @abstractmethod
    def predict(self, obs, **kwargs):
        """obs = {"board": (S,S) int array, "dice_roll": int}, already canonicalised so the
        policy plays TOP_LEFT (envs/ewn.py:289-296) -> (action [flag, dir], None)"""
        raise NotImplementedError

    def predict_batch(self, boards, dice):
        """boards (M,S,S), dice (M,) device or host arrays -> int8 (M,2) device tensor"""
        raise NotImplementedError


def obs_arrays(obs):
    import numpy as np
    board = np.asarray(obs["board"])
    return board.astype(np.int8)[None], np.asarray([int(obs["dice_roll"])], dtype=np.int8)
