"""ctypes binding of libewn_hip.so (C ABI: include/ewn_hip.h).

There is no CPU fallback: if the library is missing this module raises, loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EWN_HIP_LIB", os.path.join(HERE, "lib", "libewn_hip.so"))  # override: A/B builds while profiling

OPP = {"random": 0, "minimax": 1, "mcts": 2}
RNG = {"mt19937": 0, "philox": 1}
HEUR = {"hybrid": 0, "min_dist": 1, "two_min_dist": 2, "attk": 3, "sim_winrate": 4}
INFO_MESSAGES = {
    0: None,
    1: "Invalid move for player! End the game.",      # envs/ewn.py:448
    2: "You won!",                                     # :454
    3: "Invalid move for opponent! End the game.",    # :473
    4: "You lost!",                                    # :478
    5: "Invalid move for player! Tolerance left {}.",  # envs/training_ewn.py:56
}

EXPORTS = ["ewn_abi_version", "ewn_strerror", "ewn_rng_words", "ewn_step_scratch_bytes", "ewn_tables_bytes",
           "ewn_build_tables", "ewn_init_aux", "ewn_reset",
           "ewn_step", "ewn_legal_actions", "ewn_apply_action", "ewn_playout_wins", "ewn_evaluate", "ewn_predict_minimax", "ewn_predict_random",
           "ewn_predict_mcts", "ewn_step_k", "ewn_step_k_supported", "ewn_predict_minimax_sim", "ewn_lanes_per_game", "ewn_roll_dice",
           "ewn_policy_param_count", "ewn_step_k_policy", "ewn_a2c_scratch_bytes", "ewn_a2c_grad", "ewn_a2c_apply"]
AGENT = {"random": 0, "minimax": 1, "sample": 2, "mlp": 3}   # "mlp": the trained policy, through ewn_step_k_policy   # "sample": env.action_space.sample(), all six actions (EWN_AGENT_SAMPLE)


class EwnConfig(C.Structure):  # struct ewn_config
    _fields_ = [
        ("board_size", C.c_int32), ("cube_layer", C.c_int32), ("n_lanes", C.c_int32),
        ("opponent_kind", C.c_int32), ("max_depth", C.c_int32), ("heuristic", C.c_int32),
        ("num_simulations", C.c_int32), ("num_env_copies", C.c_int32),
        ("rng_kind", C.c_int32), ("shaped", C.c_int32), ("illegal_move_tolerance", C.c_int32),
        ("autoreset", C.c_int32), ("shaped_refresh_on_reset", C.c_int32), ("lane_offset", C.c_int32),
        ("seed_stride", C.c_uint32), ("mt_window", C.c_uint32),
        ("reward", C.c_double), ("illegal_move_reward", C.c_double),
        ("philox_key", C.c_uint64),
    ]


class EwnState(C.Structure):  # struct ewn_state
    _fields_ = [("board", C.c_void_p), ("dice", C.c_void_p), ("done", C.c_void_p), ("rng", C.c_void_p),
                ("prev_score", C.c_void_p), ("tolerance", C.c_void_p), ("tables", C.c_void_p)]


class EwnStepOut(C.Structure):  # struct ewn_step_out
    _fields_ = [("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("info", C.c_void_p),
                ("terminal_board", C.c_void_p), ("terminal_dice", C.c_void_p), ("random_action", C.c_void_p)]


class EwnRolloutOut(C.Structure):  # struct ewn_rollout_out
    _fields_ = [("board", C.c_void_p), ("dice", C.c_void_p), ("action", C.c_void_p), ("reward", C.c_void_p),
                ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("info", C.c_void_p),
                ("return_sum", C.c_void_p), ("n_steps", C.c_void_p), ("n_episodes", C.c_void_p), ("n_wins", C.c_void_p),
                ("record", C.c_void_p)]


class EwnPolicy(C.Structure):  # struct ewn_policy
    _fields_ = [("params", C.c_void_p), ("deterministic", C.c_int32), ("record_initial_obs", C.c_int32), ("noise_key", C.c_uint64),
                ("logits", C.c_void_p), ("value", C.c_void_p), ("noise", C.c_void_p)]


class EwnA2cHyper(C.Structure):  # struct ewn_a2c_hyper
    _fields_ = [("gamma", C.c_float), ("vf_coef", C.c_float), ("ent_coef", C.c_float), ("max_grad_norm", C.c_float),
                ("learning_rate", C.c_float), ("rms_alpha", C.c_float), ("rms_eps", C.c_float), ("world_size", C.c_int32)]


class EwnError(RuntimeError):
    pass


_lib = None


def load():
    """Load libewn_hip.so.  Raises EwnError if it has not been built (python -m ewn_gym_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        built = False
        if "EWN_HIP_LIB" not in os.environ:
            try:  # a fresh checkout: compile in-tree once (hipcc cross-compiles gfx950 without a GPU)
                from . import build as _build
                _build.build()
                built = os.path.exists(LIB_PATH)
            except Exception:
                built = False
        if not built:
            raise EwnError("libewn_hip.so not found at %s and could not be built: run `python -m ewn_gym_amd.build` "
                           "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
    cfgp, stp, outp = C.POINTER(EwnConfig), C.POINTER(EwnState), C.POINTER(EwnStepOut)
    sig = {
        "ewn_abi_version": (i32, []),
        "ewn_strerror": (C.c_char_p, [i32]),
        "ewn_rng_words": (i32, [cfgp]),
        "ewn_step_scratch_bytes": (C.c_int64, [cfgp]),
        "ewn_tables_bytes": (C.c_int64, [i32, i32]),
        "ewn_build_tables": (i32, [i32, i32, vp]),
        "ewn_init_aux": (i32, [cfgp, stp, vp]),
        "ewn_reset": (i32, [cfgp, stp, vp, vp, vp]),
        "ewn_roll_dice": (i32, [cfgp, stp, vp, vp]),
        "ewn_step": (i32, [cfgp, stp, vp, outp, vp, vp]),
        "ewn_legal_actions": (i32, [i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
        "ewn_apply_action": (i32, [i32, i32, i32, vp, vp, i32, vp, vp, vp, vp]),
        "ewn_playout_wins": (i32, [i32, i32, i32, vp, i32, i32, u64, vp, vp]),
        "ewn_evaluate": (i32, [i32, i32, i32, vp, i32, vp, vp]),
        "ewn_predict_minimax": (i32, [i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp]),
        "ewn_predict_random": (i32, [i32, i32, i32, vp, vp, u64, u32, vp, i32, vp, vp]),
        "ewn_predict_mcts": (i32, [i32, i32, i32, vp, vp, i32, i32, u64, vp, vp, vp, vp]),
        "ewn_predict_minimax_sim": (i32, [i32, i32, i32, vp, vp, i32, u64, vp, vp, vp, vp]),
        "ewn_step_k": (i32, [cfgp, stp, i32, i32, i32, C.POINTER(EwnRolloutOut), vp]),
        "ewn_step_k_supported": (i32, [cfgp, i32, i32]),
        "ewn_lanes_per_game": (i32, [cfgp, i32]),
        "ewn_policy_param_count": (C.c_int64, [i32, i32]),
        "ewn_step_k_policy": (i32, [cfgp, stp, i32, C.POINTER(EwnPolicy), C.POINTER(EwnRolloutOut), vp]),
        "ewn_a2c_scratch_bytes": (C.c_int64, [cfgp, i32]),
        "ewn_a2c_grad": (i32, [cfgp, i32, vp, vp, vp, C.POINTER(EwnA2cHyper), vp, vp, vp]),
        "ewn_a2c_apply": (i32, [cfgp, vp, vp, vp, C.POINTER(EwnA2cHyper), vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.ewn_abi_version() != 4:
        raise EwnError("libewn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc < 0:
        msg = load().ewn_strerror(int(rc)).decode()
        if rc == -1:
            raise AssertionError("%s: %s" % (what, msg))  # the reference asserts on bad configs (envs/ewn.py:47)
        raise EwnError("%s: %s (code %d)" % (what, msg, rc))
    return rc
