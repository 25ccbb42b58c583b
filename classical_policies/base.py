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


def reference_ctor_side_effect(cube_layer):
    """Upstream, constructing ExpectiMinimaxAgent / MctsAgent builds a private MinimaxEnv, whose constructor runs
    reset(seed=9487) (envs/minimax_ewn.py:17-19 -> envs/ewn.py:79, 488-494, 90-91): the process-global numpy stream is
    re-seeded to 9487 and one dice draw is consumed from it.  A script that builds a policy mid-run and then
    draws from np.random (a host-side RandomAgent agent) sees that; reproduce the two calls."""
    import numpy as np
    np.random.seed(9487)
    np.random.randint(1, cube_layer * (cube_layer + 1) // 2 + 1)
