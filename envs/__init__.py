"""Drop-in for the reference's `envs` package (envs/__init__.py:1-3): same class names,
constructor signatures, reset()/step() contract and entry-point strings
('envs:EinsteinWuerfeltNichtEnv', 'envs:MiniMaxHeuristicEnv', train.py:23-32).
Each instance is an N=1 view of the vectorised HIP engine (ewn_gym_amd.VecEWN)."""
from envs.ewn import EinsteinWuerfeltNichtEnv
from envs.minimax_ewn import MinimaxEnv
from envs.training_ewn import MiniMaxHeuristicEnv

__all__ = ["EinsteinWuerfeltNichtEnv", "MinimaxEnv", "MiniMaxHeuristicEnv"]
