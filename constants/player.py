"""Player enum with the reference's values and helper (constants/player.py:4-17)."""
from enum import Enum


class Player(Enum):
    TOP_LEFT = 1
    BOTTOM_RIGHT = 2
    CHANCE = 3  # only meaningful inside the expectiminimax search

    @classmethod
    def get_opponent(cls, player):
        if player == cls.TOP_LEFT:
            return cls.BOTTOM_RIGHT
        if player == cls.BOTTOM_RIGHT:
            return cls.TOP_LEFT
        raise ValueError("Invalid player")
