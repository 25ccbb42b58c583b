"""ExpectiMinimaxAgent (classical_policies/minimax.py:9-93): alpha-beta expectiminimax
with the reference's exact cut-off behaviour, as a fixed-depth kernel (depth 3 'hybrid'
runs the table-driven kernel of ewn_gym_amd/csrc/ewn_fast.hpp, other depths/heuristics
the compile-time-unrolled recursion of ewn_core.hpp)."""
import numpy as np

from classical_policies.base import PolicyBase, obs_arrays, reference_ctor_side_effect


class ExpectiMinimaxAgent(PolicyBase):
    def __init__(self, max_depth, cube_layer, board_size, heuristic="hybrid", **kwargs):
        import ewn_gym_amd
        self._ea = ewn_gym_amd
        self.max_depth = max_depth
        self.cube_layer = cube_layer
        self.board_size = board_size
        self.heuristic = heuristic
        if kwargs.get("reference_quirks", True):
            reference_ctor_side_effect(cube_layer)
        # 'sim_winrate' (envs/minimax_ewn.py:215-238): every leaf is 100 random playouts; their randomness is a per-agent key
        # and a per-call counter (the reference draws from the never-seeded Python `random`): statistical parity
        self._key = int(np.random.SeedSequence(kwargs.get("seed")).generate_state(2, np.uint32).view(np.uint64)[0])
        self._calls = 0

    def _search(self, boards, dice):
        self._calls += 1
        return self._ea.predict_minimax(boards, dice, self.max_depth, self.heuristic, cube_layer=self.cube_layer,
                                        key=(self._key + self._calls * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)

    def predict_batch(self, boards, dice, return_values=False):
        acts, vals = self._search(boards, dice)
        return (acts, vals) if return_values else acts

    def expectiminimax_root(self, obs):
        """(root value, action) of the search, i.e. expectiminimax(max_depth, TOP_LEFT, None, -inf, inf) upstream"""
        b, d = obs_arrays(obs)
        acts, vals = self._search(b, d)
        a = acts[0].cpu().numpy()
        return float(vals[0].item()), ([int(a[0]), int(a[1])] if a[0] >= 0 else None)

    def predict(self, obs, **kwargs):
        _, action = self.expectiminimax_root(obs)
        return action, None  # a Python list, like upstream (minimax.py:93)


class AlphaZeroMinimaxAgent(PolicyBase):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("AlphaZeroMinimaxAgent needs the un-vendored alpha_zero_models weights "
                                  "(classical_policies/minimax.py:96-223); out of scope of the MI355X hot path")
