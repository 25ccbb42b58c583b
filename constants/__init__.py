"""Drop-in for the reference's `constants` package (constants/__init__.py:1-2)."""
from constants.player import Player
from constants.policy import ClassicalPolicy

__all__ = ["Player", "ClassicalPolicy"]
