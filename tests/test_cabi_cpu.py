"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and validates configurations on the host (no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

from ewn_gym_amd import _lib
from ewn_gym_amd._lib import EwnConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg(**kw):
    base = dict(board_size=5, cube_layer=3, n_lanes=64, opponent_kind=0, max_depth=3, heuristic=0, num_simulations=10,
                num_env_copies=5, rng_kind=0, shaped=0, illegal_move_tolerance=10, autoreset=0, shaped_refresh_on_reset=0,
                lane_offset=0, seed_stride=64, mt_window=0, reward=1.0, illegal_move_reward=-1.0, philox_key=0)
    base.update(kw)
    return EwnConfig(**base)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ewn_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char \*)\s*\*?\s*(ewn_\w+)\s*\(", hdr, re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ewn_abi_version() == 4
    assert lib.ewn_strerror(0) == b"ok" and b"invalid" in lib.ewn_strerror(-1)


def test_struct_layout_matches_header():
    assert C.sizeof(EwnConfig) == 14 * 4 + 2 * 4 + 2 * 8 + 8
    assert C.sizeof(_lib.EwnState) == 7 * 8 and C.sizeof(_lib.EwnStepOut) == 7 * 8
    assert C.sizeof(_lib.EwnRolloutOut) == 12 * 8


def test_step_k_availability_is_decided_on_the_host():
    """ewn_step_k: table-driven configurations, and the generic K-step kernel for the geometries without a table image (RandomAgent /
    sample agents); MT19937-compat dice only without auto-reset (its windows are rebuilt between launches); unknown agents are
    invalid, unsupported ones answer 0."""
    lib = _lib.load()
    ok = lambda **kw: lib.ewn_step_k_supported(C.byref(cfg(**kw)), kw.pop("_agent", 0), kw.pop("_depth", 3))  # noqa: E731
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1, autoreset=1)), 0, 0) == 1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=0, autoreset=1)), 0, 0) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=0, autoreset=0)), 1, 5) == 1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=0, rng_kind=1)), 1, 3) == 1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=2, rng_kind=1)), 0, 0) == 1          # MCTS opponent: k_rollout_mcts (RandomAgent / sample agents)
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=2, rng_kind=1)), 2, 0) == 1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=2, rng_kind=1)), 1, 3) == 0          # ... not with a minimax agent
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=2, rng_kind=0, autoreset=1)), 0, 0) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=2, rng_kind=1, shaped=1)), 0, 0) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1, shaped=1)), 0, 0) in (0, 1)
    # no table image for cube_layer 4 / 5 and 9x9 .. 11x11: the generic K-step kernel, RandomAgent / sample agents only
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1, board_size=7, cube_layer=4)), 0, 0) == 1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=0, rng_kind=1, board_size=9, cube_layer=3)), 2, 0) == 1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1, board_size=7, cube_layer=4)), 1, 3) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=0, autoreset=1, board_size=7, cube_layer=5)), 0, 0) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1, shaped=1, board_size=7, cube_layer=4)), 0, 0) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1, heuristic=4, board_size=7, cube_layer=4)), 0, 0) == 0   # 'sim_winrate' leaves
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1)), 1, 7) == 0
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1)), 1, 0) == -1
    assert lib.ewn_step_k_supported(C.byref(cfg(opponent_kind=1, rng_kind=1)), 9, 3) == -1
    assert lib.ewn_step_k_supported(C.byref(cfg(n_lanes=0)), 0, 0) == -1
    st = _lib.EwnState()
    assert lib.ewn_step_k(C.byref(cfg(rng_kind=1)), C.byref(st), 4, 0, 0, None, None) == -2
    assert lib.ewn_step_k(C.byref(cfg(rng_kind=1)), C.byref(st), 0, 0, 0, None, None) == -1


def test_config_validation_on_host():
    lib = _lib.load()
    assert lib.ewn_rng_words(C.byref(cfg())) == 4 + 3 * 128 + 1   # header + three rotating MT windows + reset epoch
    assert lib.ewn_rng_words(C.byref(cfg(rng_kind=1))) == 4
    assert lib.ewn_rng_words(C.byref(cfg(mt_window=227))) == 4 + 3 * 227 + 1
    assert lib.ewn_rng_words(C.byref(cfg(mt_window=228))) == -1
    assert lib.ewn_rng_words(C.byref(cfg(cube_layer=4))) == -1            # assert cube_layer < board_size - 1 (envs/ewn.py:47)
    assert lib.ewn_rng_words(C.byref(cfg(board_size=9, cube_layer=3))) == 4 + 3 * 128 + 1   # 9x9 .. 11x11: mask-free generic kernels
    assert lib.ewn_rng_words(C.byref(cfg(board_size=12, cube_layer=3))) == -4  # valid upstream, not built here (7-bit positions)
    assert lib.ewn_rng_words(C.byref(cfg(n_lanes=0))) == -1
    assert lib.ewn_rng_words(C.byref(cfg(opponent_kind=1, max_depth=7))) == -4
    assert lib.ewn_rng_words(C.byref(cfg(opponent_kind=1, cube_layer=2))) == -4  # dice loop 1..6 needs 6 cubes
    assert lib.ewn_rng_words(C.byref(cfg(opponent_kind=7))) == -1
    assert lib.ewn_rng_words(None) == -2
    assert lib.ewn_step_scratch_bytes(C.byref(cfg())) == 0
    assert lib.ewn_step_scratch_bytes(C.byref(cfg(opponent_kind=2))) == 64 * (4 + 4 + 24 + 25)
    # MT kind with auto-reset: phase word + 2 x per-block counts + 2 x per-block request regions, for the lanes-per-game choice
    # that needs most (64 lanes, one lane per game: 1 block of 256 game slots, two requests each)
    assert lib.ewn_step_scratch_bytes(C.byref(cfg(autoreset=1))) == 16 + 2 * 4 * 4 + 2 * 1 * 512 * 16
    assert lib.ewn_step_scratch_bytes(C.byref(cfg(autoreset=1, rng_kind=1))) == 0


def test_null_pointers_are_rejected_before_any_launch():
    lib = _lib.load()
    st = _lib.EwnState()
    out = _lib.EwnStepOut()
    assert lib.ewn_reset(C.byref(cfg()), C.byref(st), None, None, None) == -2
    assert lib.ewn_step(C.byref(cfg()), C.byref(st), None, C.byref(out), None, None) == -2
    assert lib.ewn_init_aux(C.byref(cfg()), None, None) == -2
    assert lib.ewn_evaluate(5, 3, 4, None, 0, None, None) == -2
    assert lib.ewn_evaluate(5, 3, 0, None, 0, None, None) == 0            # empty batch is a no-op
    assert lib.ewn_predict_minimax(5, 3, 0, None, None, 3, 0, None, None, None, None) == 0
    assert lib.ewn_predict_minimax(5, 3, 0, None, None, 0, 0, None, None, None, None) == -1
    assert lib.ewn_evaluate(5, 3, 0, None, 4, None, None) == -4            # 'sim_winrate' not built
    assert lib.ewn_legal_actions(5, 3, 0, None, None, 3, None, None, None, None, None, None) == -1  # Player.CHANCE


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    import ewn_gym_amd
    if not torch.cuda.is_available():
        with pytest.raises(ewn_gym_amd.EwnError):
            ewn_gym_amd.VecEWN(4)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libewn_hip.so")
    with pytest.raises(ewn_gym_amd.EwnError):
        _lib.load()


def test_mt_refill_queue_fits_every_lanes_per_game_choice():
    """ewn_step_scratch_bytes for the MT19937 kind must cover the refill hand-off area for whichever lanes-per-game T the step
    launch picks: ctrl 16 B | cnt [2][nb4] u32 | list [2][nblk][2 * 256 / T] x 16 B, nblk = ceil(N / (256 / T)).  (A 64-lane,
    T = 1 launch once wrote 8 KB past a buffer sized for T = 4.)"""
    lib = _lib.load()
    for N in (1, 63, 64, 65, 255, 256, 257, 1000, 3000, 4096, 65536, 100001):
        have = int(lib.ewn_step_scratch_bytes(C.byref(cfg(n_lanes=N, autoreset=1))))
        for T in (1, 2, 4):
            gpb = 256 // T
            nblk = (N + gpb - 1) // gpb
            nb4 = (nblk + 3) // 4 * 4
            assert have >= 16 + 2 * nb4 * 4 + 2 * nblk * (2 * gpb) * 16, (N, T, have)


def test_default_kernel_selection_is_not_changed_by_the_environment():
    """EWN_D3_T / EWN_ROLLOUT_T / EWN_ROLLOUT_SLOTS (0 = the lock-step rollout kernel at every size) are tuning knobs read once by the library; with neither set (as in every test and bench run) the
    lanes-per-game choice is a function of the lane count alone, at the measured thresholds."""
    assert "EWN_D3_T" not in os.environ and "EWN_ROLLOUT_T" not in os.environ and "EWN_ROLLOUT_SLOTS" not in os.environ
    lib = _lib.load()
    lanes = lambda entry, **kw: lib.ewn_lanes_per_game(C.byref(cfg(rng_kind=1, autoreset=1, **kw)), entry)  # noqa: E731
    for entry in (0, 1):
        assert lanes(entry, opponent_kind=1, n_lanes=1024) == 2
        assert lanes(entry, opponent_kind=1, n_lanes=32767) == 2
        assert lanes(entry, opponent_kind=1, n_lanes=32768) == 2
        assert lanes(entry, opponent_kind=1, n_lanes=65536) == 2          # the headline configuration
        assert lanes(entry, opponent_kind=1, n_lanes=131071) == 2
        assert lanes(entry, opponent_kind=1, n_lanes=131072) == 1
        assert lanes(entry, opponent_kind=0, n_lanes=65536) == 1          # RandomAgent opponent: one lane per game
        assert lanes(entry, opponent_kind=1, max_depth=5, n_lanes=65536) == 2
        assert lanes(entry, opponent_kind=1, max_depth=5, n_lanes=262144) == 1
        assert lanes(entry, opponent_kind=1, heuristic=2, n_lanes=65536) == 2   # 'two_min_dist': the fused step kernel and (since round 3) the K-step rollout kernels have instances for its image
        assert lanes(entry, opponent_kind=1, board_size=7, cube_layer=4, n_lanes=65536) == 0
    assert lanes(0, opponent_kind=1, shaped=1, n_lanes=65536) == 2 and lanes(1, opponent_kind=1, shaped=1, n_lanes=65536) == 0
    assert lib.ewn_lanes_per_game(C.byref(cfg(n_lanes=0)), 0) == -1
