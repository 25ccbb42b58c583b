"""RandomAgent (classical_policies/random_policy.py:6-15): ignores `obs`, looks at the
LIVE env it was constructed with, and draws a uniform index into that env's legal
actions from the global numpy RNG, as upstream.  The legal-action list itself comes from
the HIP engine.

Parity of a host-side RandomAgent *agent* with the reference is STATISTICAL only: upstream
the env's roll_dice and a random opponent consume the same global numpy stream between the
agent's draws, here the dice live on the device (one numpy-compatible stream per episode),
so the agent's draws -- and from its first move on the trajectory -- differ from upstream's
for the same seed (deviation i).  As the env's OPPONENT the policy runs
inside the step kernel and is bit-exact."""
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
