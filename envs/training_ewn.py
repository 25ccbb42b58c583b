"""MiniMaxHeuristicEnv (envs/training_ewn.py:14-99): TD-shaped reward from the hybrid
heuristic plus a tolerance for illegal moves -- flags of the same fused step kernel."""
from typing import Optional

from constants import ClassicalPolicy, Player
from envs.minimax_ewn import MinimaxEnv


class MiniMaxHeuristicEnv(MinimaxEnv):
    _shaped = True

    def __init__(self, board_size: int = 5, cube_layer: int = 3, seed: int = 9487, goal_reward: float = 10.,
                 agent_player: Player = Player.TOP_LEFT, render_mode: Optional[str] = None,
                 opponent_policy=ClassicalPolicy.random, illegal_move_reward: float = -1.0,
                 illegal_move_tolerance: int = 10, reference_quirks: bool = True, **policy_kwargs):
        self._illegal_move_reward = illegal_move_reward
        self._tolerance0 = illegal_move_tolerance
        self._refresh = not reference_quirks  # upstream sets prev_score in the ctor only (App. D3)
        super().__init__(board_size=board_size, cube_layer=cube_layer, seed=seed, reward=goal_reward,
                         agent_player=agent_player, render_mode=render_mode, opponent_policy=opponent_policy,
                         reference_quirks=reference_quirks, **policy_kwargs)

    def _engine_kwargs(self):
        kw = super()._engine_kwargs()
        kw.update(shaped=True, illegal_move_reward=self._illegal_move_reward, illegal_move_tolerance=self._tolerance0,
                  shaped_refresh_on_reset=self._refresh)
        return kw

    @property
    def prev_score(self) -> float:
        return float(self._engine.prev_score[0].item())

    @property
    def illegal_move_reward(self) -> float:
        return self._illegal_move_reward

    @property
    def illegal_move_tolerance(self) -> int:
        return int(self._engine.tolerance[0].item())
