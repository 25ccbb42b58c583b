"""RandomAgent (classical_policies/random_policy.py:6-15): ignores `obs`, looks at the
LIVE env it was constructed with, and draws a uniform index into that env's legal
actions from the global numpy RNG -- exactly as upstream, so a seeded script sees the
same stream.  The legal-action list itself comes from the HIP engine."""
import numpy as np

from classical_policies.base import PolicyBase


class RandomAgent(PolicyBase):
    def __init__(self, env):
        self.env = env

    def predict(self, obs, **kwargs):
        env = getattr(self.env, "unwrapped", self.env)
        legal_moves = env.get_legal_actions(env.current_player)
        action_idx = np.random.randint(0, len(legal_moves))
        return np.array(legal_moves[action_idx]), None

    def predict_batch(self, boards, dice, key=0, step=0):
        import ewn_gym_amd
        return ewn_gym_amd.predict_random(boards, dice, key=key, step=step)
