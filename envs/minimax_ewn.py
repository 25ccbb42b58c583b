"""MinimaxEnv (envs/minimax_ewn.py:11-238): the env plus board heuristics."""
from typing import Optional

import numpy as np

from constants import ClassicalPolicy, Player
from envs.ewn import EinsteinWuerfeltNichtEnv


class MinimaxEnv(EinsteinWuerfeltNichtEnv):
    def __init__(self, board_size: int = 5, cube_layer: int = 3, seed: int = 9487, reward: float = 1.,
                 agent_player: Player = Player.TOP_LEFT, render_mode: Optional[str] = None,
                 opponent_policy=ClassicalPolicy.random, reference_quirks: bool = True, **policy_kwargs):
        if reference_quirks:
            # the reference forwards ONLY board_size and cube_layer (envs/minimax_ewn.py:17-19), so every subclass
            # (incl. the training env) really gets RandomAgent, reward 1.0 and seed 9487
            super().__init__(board_size=board_size, cube_layer=cube_layer)
        else:
            super().__init__(board_size=board_size, cube_layer=cube_layer, seed=seed, reward=reward,
                             agent_player=agent_player, render_mode=render_mode, opponent_policy=opponent_policy,
                             **policy_kwargs)
        self.num_simulations = 100

    def set_dice_roll(self, roll: int):
        self.dice_roll = roll
        self._engine.dice[0] = int(roll)

    def simulate(self) -> float:
        """envs/minimax_ewn.py:215-238: fraction of num_simulations random playouts TOP_LEFT wins; the player who is NOT
        `current_player` moves first (upstream switches before every move), and -- upstream quirk -- current_player is
        left wherever the last playout ended; here it is simply left unchanged.  Statistical parity only."""
        import os
        import ewn_gym_amd
        first = Player.get_opponent(self.current_player)
        wins = ewn_gym_amd.playout_wins(self.board.astype(np.int8)[None], first.value, self.num_simulations,
                                        key=int.from_bytes(os.urandom(8), "little"), cube_layer=self.cube_layer)
        return int(wins[0].item()) / self.num_simulations

    def evaluate(self, heuristic="hybrid"):
        import ewn_gym_amd
        if heuristic == "sim_winrate":
            return self.simulate()
        v = float(ewn_gym_amd.evaluate(self.board.astype(np.int8)[None], heuristic, cube_layer=self.cube_layer)[0].item())
        if heuristic != "hybrid" or abs(v) == 10.0:
            return int(v)  # upstream returns Python ints for the integer heuristics and the +-10 terminals
        return v
