"""EinsteinWuerfeltNichtEnv: the reference's gymnasium environment (envs/ewn.py:18-576)
as a single-lane view of the MI355X engine.  All rules, dice and the opponent's reply
run in libewn_hip.so; this class only mirrors the API surface and keeps host copies of
the observation for callers that read `env.board` / `env.dice_roll`."""
import os
from typing import List, Optional

import numpy as np

from constants import ClassicalPolicy, Player

try:  # gymnasium is optional: the reference needs it, the engine does not
    import gymnasium as gym
    from gymnasium import spaces
    from gymnasium.error import DependencyNotInstalled
    _Base = gym.Env
except ImportError:  # pragma: no cover - depends on the image
    from ewn_gym_amd import spaces_compat as spaces
    from ewn_gym_amd.spaces_compat import DependencyNotInstalled
    _Base = object

VIEWPORT_SIZE = 700
FONT_SIZE = VIEWPORT_SIZE // 25
FPS = 30


class EinsteinWuerfeltNichtEnv(_Base):
    metadata = {"render_modes": ["human", "rgb_array", "ansi"], "render_fps": FPS}
    _shaped = False

    def __init__(self, board_size: int = 5, cube_layer: int = 3, seed: int = 9487, reward: float = 1.,
                 agent_player: Player = Player.TOP_LEFT, render_mode: Optional[str] = None,
                 opponent_policy=ClassicalPolicy.random, **policy_kwargs):
        super().__init__()
        assert cube_layer < board_size - 1                      # envs/ewn.py:47
        if agent_player != Player.TOP_LEFT:
            # the reference's BOTTOM_RIGHT-agent branch (:110-129) is dead/broken upstream
            raise NotImplementedError("only agent_player=Player.TOP_LEFT is supported")
        self.board = np.zeros((board_size, board_size), dtype=np.int16)
        self.cube_num = cube_layer * (cube_layer + 1) // 2
        self.cube_layer = cube_layer
        self.dice_roll = 1
        self.action_space = spaces.MultiDiscrete([2, 3])
        self.observation_space = spaces.Dict({
            "board": spaces.Box(low=-self.cube_num, high=self.cube_num, shape=(board_size, board_size), dtype=np.int16),
            "dice_roll": spaces.Discrete(self.cube_num + 1, start=1),  # width cube_num+1: the SB3 one-hot workaround, :66-68
        })
        self.current_player = Player.TOP_LEFT
        self.agent_player = agent_player
        self.reward = reward
        assert opponent_policy is not None
        self._policy_kwargs = dict(policy_kwargs)
        self.load_opponent_policy(opponent_policy, **policy_kwargs)
        self._engine = self._make_engine()
        self.reset(seed=seed)
        self.render_mode = render_mode
        self.screen = None
        self.clock = None
        self.surf = None
        self.history = []

    # ------------------------------------------------------------------ engine plumbing
    def _engine_kwargs(self):
        kw = self._policy_kwargs
        return dict(board_size=self.board.shape[0], cube_layer=self.cube_layer, opponent_policy=str(self._opponent_kind),
                    max_depth=kw.get("max_depth", 3), heuristic=kw.get("heuristic", "hybrid"),
                    num_simulations=kw.get("num_simulations", 10), num_env_copies=kw.get("num_env_copies", 5),
                    rng="mt19937", reward=self.reward, philox_key=int.from_bytes(os.urandom(8), "little"))

    def _make_engine(self):
        import ewn_gym_amd
        return ewn_gym_amd.VecEWN(1, **self._engine_kwargs())

    def _pull(self):
        self.board[:] = self._engine.board[0].cpu().numpy()     # in place: obs["board"] aliases env state upstream too
        self.dice_roll = int(self._engine.dice[0].item())

    def _obs(self):
        return {"board": self.board, "dice_roll": self.dice_roll}

    # ------------------------------------------------------------------ reference API
    def load_opponent_policy(self, opponent_policy, **policy_kwargs):
        """envs/ewn.py:265-287.  The reply itself is computed inside the fused step kernel;
        the agent object is kept for introspection (`env.opponent_policy`)."""
        from classical_policies import ExpectiMinimaxAgent, MctsAgent, RandomAgent
        if opponent_policy == ClassicalPolicy.random:
            self.opponent_policy = RandomAgent(self)
        elif opponent_policy == ClassicalPolicy.minimax:
            self.opponent_policy = ExpectiMinimaxAgent(cube_layer=self.cube_layer, board_size=self.board.shape[0],
                                                       **policy_kwargs)  # max_depth is required, as upstream
        elif opponent_policy == ClassicalPolicy.mcts:
            self.opponent_policy = MctsAgent(cube_layer=self.cube_layer, board_size=self.board.shape[0], **policy_kwargs)
        elif isinstance(opponent_policy, ClassicalPolicy):
            raise NotImplementedError("opponent policy %s is out of scope of the HIP engine" % opponent_policy)
        else:
            assert isinstance(opponent_policy, str)
            raise NotImplementedError("SB3 checkpoint opponents (A2C.load, envs/ewn.py:287) need stable_baselines3")
        self._opponent_kind = opponent_policy

    def reset(self, seed: Optional[int] = None):
        """envs/ewn.py:488-494.  seed=None draws OS entropy, like np.random.seed(None)."""
        self.current_player = Player.TOP_LEFT
        if seed is None:
            seed = int.from_bytes(os.urandom(4), "little")
        np.random.seed(seed)          # the reference seeds the GLOBAL numpy stream here; host-side agents rely on it
        self.action_space.seed(seed)
        self._engine.reset(seeds=[int(seed) & 0xFFFFFFFF])
        self._pull()
        return self._obs(), {}

    def step(self, action):
        """envs/ewn.py:436-486 -> (obs, reward, terminated, truncated, info)"""
        import torch
        import ewn_gym_amd
        a = np.asarray(action).reshape(-1)
        e = self._engine
        _, _, r, te, tr, info = e.step(np.array([[int(a[0]), int(a[1])]], dtype=np.int8))
        # ONE device-to-host copy (and one synchronisation) for everything the caller gets back: board, dice, reward, the three
        # flags, the remaining tolerance (shaped env) and the MT-stream overflow flag
        S2 = self.board.size
        tol = e.tolerance if e.tolerance is not None else torch.zeros(1, dtype=torch.int32, device=e.device)
        packed = torch.cat([e.board.reshape(-1).view(torch.uint8), e.dice.view(torch.uint8), te, tr, info, e.rng_overflow(),
                            r.view(torch.uint8), tol.view(torch.uint8)]).cpu().numpy()
        self.board[:] = packed[:S2].view(np.int8).reshape(self.board.shape)   # in place: obs["board"] aliases env state upstream too
        self.dice_roll = int(packed[S2:S2 + 1].view(np.int8)[0])
        terminated, truncated, code, overflow = (int(x) for x in packed[S2 + 1:S2 + 5])
        reward = float(packed[S2 + 5:S2 + 13].view(np.float64)[0])
        if overflow:
            e.check_rng()             # raises: an episode longer than the MT19937-compat stream supports must not pass silently
        msg = ewn_gym_amd.INFO_MESSAGES[code]
        if code == 5:
            msg = msg.format(int(packed[S2 + 13:S2 + 17].view(np.int32)[0]))
        self.current_player = Player.BOTTOM_RIGHT if code in (3, 4) else Player.TOP_LEFT
        return self._obs(), reward, bool(terminated), bool(truncated), ({"message": msg} if msg else {})

    def roll_dice(self):
        """envs/ewn.py:90-92: dice_roll = np.random.randint(1, cube_num + 1) -- drawn on the device from this env's own dice
        stream (the numpy-compatible MT19937 kind: the same value upstream's next global draw gives a freshly seeded env)."""
        self.dice_roll = int(self._engine.roll_dice().cpu()[0])

    def switch_player(self):
        self.current_player = Player.get_opponent(self.current_player)

    def _query(self, player):
        import ewn_gym_amd
        return [t.cpu().numpy() for t in ewn_gym_amd.legal_actions(self.board.astype(np.int8)[None], [self.dice_roll],
                                                                   player=player.value, cube_layer=self.cube_layer)]

    def check_win(self) -> bool:
        return bool(self._query(Player.TOP_LEFT)[4][0])

    def get_legal_actions(self, player: Player) -> List[List[int]]:
        if player == Player.CHANCE:
            raise ValueError("Invalid player")
        acts, n = self._query(player)[:2]
        return [[int(a[0]), int(a[1])] for a in acts[0, :int(n[0])]]

    def find_cube_to_move(self, chose_larger: bool, player: Optional[Player] = None) -> int:
        player = self.current_player if player is None else player
        q = self._query(player)
        num = int(q[3][0] if chose_larger else q[2][0])
        assert num != 0
        return num - 1 if player == Player.TOP_LEFT else -num    # python-style index into cube_pos, :185-186

    def _push_board(self):
        self._engine.set_obs(self.board.astype(np.int8)[None], [self.dice_roll])

    def make_simulated_action(self, player: Player, action):
        """envs/ewn.py:377-412: apply [flag, dir] for `player` on the host-visible board, remembering how to undo it.
        An illegal move pushes None, like upstream."""
        import ewn_gym_amd
        a = np.asarray(action).reshape(-1)
        nb, valid = ewn_gym_amd.apply_action(self.board.astype(np.int8)[None], [self.dice_roll], [[int(a[0]), int(a[1])]],
                                             player=player.value, cube_layer=self.cube_layer)
        if not bool(valid[0].item()):
            self.history.append(None)
            return
        self.history.append(self.board.copy())
        self.board[:] = nb[0].cpu().numpy()
        self._push_board()

    def undo_simulated_action(self):
        """envs/ewn.py:414-434"""
        if not self.history:
            return
        last = self.history.pop()
        if last is None:
            return
        self.board[:] = last
        self._push_board()

    @property
    def cube_pos(self):
        """The reference's masked structured array of cube coordinates (envs/ewn.py:57), rebuilt from the board."""
        cp = np.ma.zeros((self.cube_num * 2,), dtype=[("x", int), ("y", int)])
        cp[:] = np.ma.masked
        for (i, j), c in np.ndenumerate(self.board):
            if c > 0:
                cp[c - 1] = (i, j)
            elif c < 0:
                cp[c] = (i, j)
        return cp

    def render(self):
        if self.render_mode == "ansi":                           # envs/ewn.py:498-502
            print("dice:")
            print(self.dice_roll)
            print("board:")
            print(self.board)
        elif self.render_mode == "rgb_array":
            return self._render_rgb()
        elif self.render_mode == "human":
            raise DependencyNotInstalled("the pygame window (envs/ewn.py:503-569) is not built; use render_mode='rgb_array' or 'ansi'")

    def _render_rgb(self):
        """The frame envs/ewn.py:520-569 draws with pygame, drawn with PIL ('rgb via PIL only if pygame absent'):
        same geometry and colours -- (VIEWPORT_SIZE + FONT_SIZE) x VIEWPORT_SIZE x 3 uint8, board colour (211, 179, 104), grid
        lines every VIEWPORT_SIZE // S pixels, TOP_LEFT cubes white / BOTTOM_RIGHT black discs of radius 0.4 cells with the
        cube number, "dice: n" under the board.  Glyph rasterisation differs from SDL's, nothing else."""
        try:
            from PIL import Image, ImageDraw, ImageFont
        except ImportError as e:   # pragma: no cover
            raise DependencyNotInstalled("rgb_array rendering needs Pillow") from e
        S = self.board.shape[0]
        img = Image.new("RGB", (VIEWPORT_SIZE, VIEWPORT_SIZE + FONT_SIZE), (211, 179, 104))
        dr = ImageDraw.Draw(img)
        try:
            font = ImageFont.load_default(size=FONT_SIZE * 3 // 4)
        except TypeError:          # pragma: no cover - old Pillow
            font = ImageFont.load_default()
        dr.text((0, VIEWPORT_SIZE), "dice: %s" % self.dice_roll, fill=(0, 0, 0), font=font)
        lw = VIEWPORT_SIZE // S
        for i in range(1, S):
            dr.line([(0, i * lw), (VIEWPORT_SIZE, i * lw)], fill=(0, 0, 0))
            dr.line([(i * lw, 0), (i * lw, VIEWPORT_SIZE)], fill=(0, 0, 0))
        dr.line([(0, VIEWPORT_SIZE), (VIEWPORT_SIZE, VIEWPORT_SIZE)], fill=(0, 0, 0))
        rad = int(lw * 0.4)
        for (x, y), cube in np.ndenumerate(self.board):
            if cube == 0:
                continue
            color, text_color = ((255, 255, 255), (0, 0, 0)) if cube > 0 else ((0, 0, 0), (255, 255, 255))
            cx, cy = int((y + 0.5) * lw), int((x + 0.5) * lw)
            dr.ellipse([cx - rad, cy - rad, cx + rad, cy + rad], fill=color, outline=color)
            dr.text((cx - FONT_SIZE // 3, cy - FONT_SIZE // 2), str(abs(int(cube))), fill=text_color, font=font)
        return np.asarray(img, dtype=np.uint8)

    def close(self):
        pass
