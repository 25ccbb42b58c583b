"""Import stub, container-only: classical_policies/__init__.py:4 drags in
alpha_zero/ewn/EWNPlayers.py:2 which imports pygame at module top. Never called."""
