"""Placeholder for the reference's AlphaZero fork (classical_policies/alpha_zero/**):
dense-network inference whose weights live in an un-vendored submodule; out of scope."""


class AlphaZeroAgent:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("AlphaZeroAgent needs the un-vendored alpha_zero_models weights; "
                                  "out of scope of the MI355X hot path (SURVEY section 2, rows 13-14)")
