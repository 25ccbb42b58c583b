"""Drop-in for the reference's `classical_policies` package
(classical_policies/__init__.py:1-4): same class names, constructor signatures and
`predict(obs, **kw) -> (action, None)` contract; every search runs in libewn_hip.so.

AlphaZeroAgent / AlphaZeroMinimaxAgent are out of scope (un-vendored weights, SURVEY
section 2 rows 12-14): they exist as names that raise on construction.
"""
from classical_policies.base import PolicyBase
from classical_policies.random_policy import RandomAgent
from classical_policies.minimax import ExpectiMinimaxAgent, AlphaZeroMinimaxAgent
from classical_policies.mcts import MctsAgent
from classical_policies.alpha_zero import AlphaZeroAgent

# the names BASELINE.json's north_star uses
RandomPolicy = RandomAgent
MiniMaxPolicy = ExpectiMinimaxAgent
MCTSPolicy = MctsAgent

__all__ = ["PolicyBase", "RandomAgent", "ExpectiMinimaxAgent", "MctsAgent", "AlphaZeroAgent", "AlphaZeroMinimaxAgent",
           "RandomPolicy", "MiniMaxPolicy", "MCTSPolicy"]
