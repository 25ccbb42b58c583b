"""ctypes binding of the CPU oracle (oracle/ewn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Product code must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OPP = {"random": 0, "minimax": 1, "mcts": 2}
RNG = {"mt19937": 0, "philox": 1}
HEUR = {"hybrid": 0, "min_dist": 1, "two_min_dist": 2, "attk": 3, "sim_winrate": 4}
INFO_MESSAGES = {
    0: None,
    1: "Invalid move for player! End the game.",
    2: "You won!",
    3: "Invalid move for opponent! End the game.",
    4: "You lost!",
    5: "Invalid move for player! Tolerance left",
}


class Cfg(C.Structure):
    _fields_ = [
        ("board_size", C.c_int32), ("cube_layer", C.c_int32), ("n_lanes", C.c_int32),
        ("opponent_kind", C.c_int32), ("max_depth", C.c_int32), ("heuristic", C.c_int32),
        ("num_simulations", C.c_int32), ("num_env_copies", C.c_int32),
        ("rng_kind", C.c_int32), ("shaped", C.c_int32), ("illegal_move_tolerance", C.c_int32),
        ("autoreset", C.c_int32), ("shaped_refresh_on_reset", C.c_int32), ("lane_offset", C.c_int32),
        ("seed_stride", C.c_uint32), ("reserved", C.c_uint32),
        ("reward", C.c_double), ("illegal_move_reward", C.c_double),
        ("philox_key", C.c_uint64),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libewn_oracle.so")
    src = os.path.join(_HERE, "ewn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libewn_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ewn_oracle_create.restype = C.c_void_p
        _LIB.ewn_oracle_create.argtypes = [C.POINTER(Cfg)]
        _LIB.ewn_oracle_destroy.argtypes = [C.c_void_p]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def predict_minimax(boards, dice, max_depth, heuristic="hybrid", cube_layer=3):
    boards = _i8(boards)
    M, S = boards.shape[0], boards.shape[1]
    dice = _i8(dice)
    acts = np.zeros((M, 2), np.int8)
    vals = np.zeros(M, np.float64)
    leaves = np.zeros(M, np.uint64)
    rc = lib().ewn_oracle_predict_minimax(C.c_int(S), C.c_int(cube_layer), C.c_int(M), _p(boards), _p(dice), C.c_int(max_depth),
                                          C.c_int(HEUR[heuristic]), _p(acts), _p(vals), _p(leaves))
    assert rc == 0
    return acts, vals, leaves


def predict_minimax_sim(boards, dice, max_depth, key=0, obs_id=None, nsims=100, cube_layer=3):
    """ExpectiMinimaxAgent(heuristic='sim_winrate').predict with the HIP engine's playout generator"""
    boards = _i8(boards)
    M, S = boards.shape[0], boards.shape[1]
    dice = _i8(dice)
    acts = np.zeros((M, 2), np.int8)
    vals = np.zeros(M, np.float64)
    ids = None if obs_id is None else np.ascontiguousarray(obs_id, dtype=np.uint32)
    rc = lib().ewn_oracle_predict_minimax_sim(C.c_int(S), C.c_int(cube_layer), C.c_int(M), _p(boards), _p(dice), C.c_int(max_depth),
                                              C.c_uint64(key), _p(ids), C.c_int(nsims), _p(acts), _p(vals))
    assert rc == 0
    return acts, vals


def evaluate(boards, heuristic="hybrid", cube_layer=3):
    boards = _i8(boards)
    M, S = boards.shape[0], boards.shape[1]
    out = np.zeros(M, np.float64)
    rc = lib().ewn_oracle_evaluate(C.c_int(S), C.c_int(cube_layer), C.c_int(M), _p(boards), C.c_int(HEUR[heuristic]), _p(out))
    assert rc == 0
    return out


def legal_actions(boards, dice, player=1, cube_layer=3):
    boards = _i8(boards)
    M, S = boards.shape[0], boards.shape[1]
    dice = _i8(dice)
    acts = np.zeros((M, 6, 2), np.int8)
    n = np.zeros(M, np.int8)
    cs = np.zeros(M, np.int8)
    cl = np.zeros(M, np.int8)
    win = np.zeros(M, np.uint8)
    rc = lib().ewn_oracle_legal_actions(C.c_int(S), C.c_int(cube_layer), C.c_int(M), _p(boards), _p(dice), C.c_int(player),
                                        _p(acts), _p(n), _p(cs), _p(cl), _p(win))
    assert rc == 0
    return acts, n, cs, cl, win


def predict_mcts(boards, dice, num_simulations=10, num_env_copies=5, key=0, obs_id=None, cube_layer=3):
    boards = _i8(boards)
    M, S = boards.shape[0], boards.shape[1]
    dice = _i8(dice)
    acts = np.zeros((M, 2), np.int8)
    wins = np.zeros((M, 6), np.int32)
    ids = None if obs_id is None else np.ascontiguousarray(obs_id, dtype=np.uint32)
    rc = lib().ewn_oracle_predict_mcts(C.c_int(S), C.c_int(cube_layer), C.c_int(M), _p(boards), _p(dice), C.c_int(num_simulations),
                                       C.c_int(num_env_copies), C.c_uint64(key), _p(ids), _p(acts), _p(wins))
    assert rc == 0
    return acts, wins


def playout_wins(boards, first_player, n_sims=100, key=0, cube_layer=3):
    boards = _i8(boards)
    M, S = boards.shape[0], boards.shape[1]
    wins = np.zeros(M, np.int32)
    rc = lib().ewn_oracle_playout_wins(C.c_int(S), C.c_int(cube_layer), C.c_int(M), _p(boards), C.c_int(first_player), C.c_int(n_sims),
                                       C.c_uint64(key), _p(wins))
    assert rc == 0
    return wins


def prng_draws(c1, x, count, tag=0x4D435453, key=0):
    """(dice - 1, 24-bit fraction) of `count` plies of one playout stream"""
    d = np.zeros(count, np.uint8)
    f = np.zeros(count, np.uint32)
    lib().ewn_oracle_prng_draws(C.c_uint32(c1), C.c_uint32(x), C.c_uint32(tag), C.c_uint64(key), C.c_int(count), _p(d), _p(f))
    return d, f


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().ewn_oracle_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def np_randint_seq(seed, lo, hi):
    lo = np.ascontiguousarray(lo, np.int32)
    hi = np.ascontiguousarray(hi, np.int32)
    out = np.zeros(len(lo), np.int32)
    lib().ewn_oracle_np_randint_seq(C.c_uint32(seed), C.c_int(len(lo)), _p(lo), _p(hi), _p(out))
    return out


def mt_outputs(seed, count):
    out = np.zeros(count, np.uint32)
    lib().ewn_oracle_mt_outputs(C.c_uint32(seed), C.c_int(count), _p(out))
    return out


class OracleVecEnv:
    """N independent games stepped on the CPU by the oracle (same call shape as
    the HIP engine's VecEWN so the parity tests read the same on both sides)."""

    def __init__(self, n_lanes, board_size=5, cube_layer=3, opponent="random", max_depth=3, heuristic="hybrid",
                 num_simulations=10, num_env_copies=5, rng="mt19937", shaped=False, illegal_move_tolerance=10,
                 autoreset=False, shaped_refresh_on_reset=False, lane_offset=0, seed_stride=None, reward=1.0,
                 illegal_move_reward=-1.0, philox_key=0):
        self.cfg = Cfg(board_size, cube_layer, n_lanes, OPP[opponent], max_depth, HEUR[heuristic], num_simulations,
                       num_env_copies, RNG[rng], int(shaped), illegal_move_tolerance, int(autoreset),
                       int(shaped_refresh_on_reset), lane_offset, n_lanes if seed_stride is None else seed_stride, 0,
                       reward, illegal_move_reward, philox_key)
        self.N, self.S = n_lanes, board_size
        self.h = lib().ewn_oracle_create(C.byref(self.cfg))
        if not self.h:
            raise AssertionError("bad config")  # envs/ewn.py:47

    def __del__(self):
        if getattr(self, "h", None) and lib is not None and C is not None:   # (module globals are gone when the interpreter shuts down)
            lib().ewn_oracle_destroy(C.c_void_p(self.h))
            self.h = None

    def obs(self):
        b = np.zeros((self.N, self.S, self.S), np.int8)
        d = np.zeros(self.N, np.int8)
        lib().ewn_oracle_get_obs(C.c_void_p(self.h), _p(b), _p(d))
        return b, d

    def set_obs(self, boards, dice):
        lib().ewn_oracle_set_obs(C.c_void_p(self.h), _p(_i8(boards)), _p(_i8(dice)))

    def aux(self):
        ps = np.zeros(self.N, np.float64)
        tol = np.zeros(self.N, np.int32)
        dr = np.zeros(self.N, np.uint64)
        lib().ewn_oracle_get_aux(C.c_void_p(self.h), _p(ps), _p(tol), _p(dr))
        return ps, tol, dr

    def reset(self, seeds=None, mask=None):
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().ewn_oracle_reset(C.c_void_p(self.h), _p(s), _p(m))
        return self.obs()

    def step(self, actions, want_terminal=False):
        a = _i8(actions).reshape(self.N, 2)
        b = np.zeros((self.N, self.S, self.S), np.int8)
        d = np.zeros(self.N, np.int8)
        r = np.zeros(self.N, np.float64)
        te = np.zeros(self.N, np.uint8)
        tr = np.zeros(self.N, np.uint8)
        info = np.zeros(self.N, np.uint8)
        tb = np.zeros((self.N, self.S, self.S), np.int8) if want_terminal else None
        td = np.zeros(self.N, np.int8) if want_terminal else None
        lib().ewn_oracle_step(C.c_void_p(self.h), _p(a), _p(b), _p(d), _p(r), _p(te), _p(tr), _p(info), _p(tb), _p(td))
        out = (b, d, r, te, tr, info)
        return out + (tb, td) if want_terminal else out

    def roll_dice(self, mask=None):
        """roll_dice (envs/ewn.py:90-92) on the selected lanes; returns the dice"""
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().ewn_oracle_roll_dice(C.c_void_p(self.h), _p(m))
        return self.obs()[1]

    def policy_noise(self, noise_key=0):
        """[N, 5] float32: the uniforms ewn_step_k_policy turns into Gumbel noise at the lanes' current step"""
        u = np.zeros((self.N, 5), np.float32)
        lib().ewn_oracle_policy_noise(C.c_void_p(self.h), C.c_uint64(noise_key), _p(u))
        return u

    def random_actions(self):
        a = np.zeros((self.N, 2), np.int8)
        lib().ewn_oracle_random_actions(C.c_void_p(self.h), _p(a))
        return a

    def sample_actions(self):
        """env.action_space.sample() per lane, the hash-driven draw of ewn_step_k's EWN_AGENT_SAMPLE"""
        a = np.zeros((self.N, 2), np.int8)
        lib().ewn_oracle_sample_actions(C.c_void_p(self.h), _p(a))
        return a

    def sample_legal_actions(self, step):
        a = np.zeros((self.N, 2), np.int8)
        lib().ewn_oracle_sample_legal_actions(C.c_void_p(self.h), C.c_uint32(step), _p(a))
        return a
