"""MctsAgent (classical_policies/mcts.py:10-106): flat Monte-Carlo -- for every legal root
move, num_env_copies x num_simulations uniformly random playouts, BOTTOM_RIGHT replying
first; pick the first move with the most wins.  One GPU thread per playout; the
reference forks a multiprocessing.Pool per move.  Rollout randomness is a Philox stream
(the reference uses an unseeded Python `random`), so parity is statistical."""
import itertools

import numpy as np

from classical_policies.base import PolicyBase, obs_arrays, reference_ctor_side_effect


class MctsAgent(PolicyBase):
    _calls = itertools.count()

    def __init__(self, cube_layer, board_size, num_simulations=10, num_env_copies=5, seed=None, **kwargs):
        import ewn_gym_amd
        self._ea = ewn_gym_amd
        self.cube_layer = cube_layer
        self.board_size = board_size
        self.num_simulations = num_simulations
        self.num_env_copies = num_env_copies
        if kwargs.get("reference_quirks", True):
            reference_ctor_side_effect(cube_layer)
        self.key = int(np.random.SeedSequence(seed).generate_state(2, np.uint32).view(np.uint64)[0])

    def predict_batch(self, boards, dice, return_wins=False):
        call = next(self._calls)
        acts, wins = self._ea.predict_mcts(boards, dice, self.num_simulations, self.num_env_copies,
                                           key=(self.key + call * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF,
                                           cube_layer=self.cube_layer)
        return (acts, wins) if return_wins else acts

    def predict(self, obs, **kwargs):
        b, d = obs_arrays(obs)
        return self.predict_batch(b, d)[0].cpu().numpy(), None  # np.array, like upstream (mcts.py:69)
