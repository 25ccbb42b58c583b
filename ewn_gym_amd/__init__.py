"""ewn_gym_amd: MI355X-native vectorised EinStein-wuerfelt-nicht env step + opponent search.

Only what the hot path needs: csrc/ (HIP kernels + C ABI), the ctypes binding, and
VecEWN (the batched engine the drop-in `envs` / `classical_policies` packages wrap).
"""
from ._lib import EwnError, INFO_MESSAGES, LIB_PATH  # noqa: F401
from .vec_env import (VecEWN, apply_action, evaluate, legal_actions, playout_wins, predict_mcts,  # noqa: F401
                      predict_minimax, predict_random)

__all__ = ["VecEWN", "EwnError", "INFO_MESSAGES", "legal_actions", "apply_action", "playout_wins", "evaluate", "predict_minimax", "predict_random",
           "predict_mcts"]
