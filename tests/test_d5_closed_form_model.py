"""A Python model of the CLOSED FORM the max_depth 5 / 6 kernel search uses (ewn_gym_amd/csrc/ewn_search_d5.hpp),
checked on the CPU against the oracle's literal recursion (classical_policies/minimax.py:19-73).

The reference searches root move -> dice -> reply -> dice -> move -> leaf with its alpha-beta window handed through
the chance nodes unchanged.  The model below states what that loop computes WITHOUT recursion, breaks or a window:

* a root's value is  sum_d1 M3(d1) / 6;  M3(d1) scans the replies of the cube pair (F = the dice cube or its larger
  neighbour, G = the smaller neighbour) the dice selects (envs/ewn.py:338-375);
* the value of one reply, C2(reply, beta) = sum_d2 X1(d2) / 6, depends on the scan only through beta = the running
  minimum in front of it; X1 stops at the first prefix maximum >= beta (the root's alpha never reaches beta);
* a cube's three replies AS F start from beta = +inf: their values are the same for every dice that selects the cube
  as F.  AS G the chain starts from its F partner's minimum -- and a cube's F partner is always the next cube above it
  that is still on the board, so that chain too is computed once.  At most 6 cubes x (3 + 3) reply evaluations per root
  instead of 6 dice x 6 replies, over <= 18 x 18 distinct leaf positions per root instead of 6 x 6 x 6 x 6.

The HIP kernel evaluates exactly these quantities (leaves as table ranks); this file pins the derivation itself.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

INF = float("inf")
DIRS_P = ((0, 1), (1, 0), (1, 1))      # TOP_LEFT (the searcher, positive cubes): right, down, diagonal
DIRS_N = ((0, -1), (-1, 0), (-1, -1))  # BOTTOM_RIGHT (the replier, negative cubes)


def hybrid(P, N, S):
    """envs/minimax_ewn.py:56-86 for agent_player = TOP_LEFT on a position nobody has won"""
    mdp = min(max(S - 1 - i, S - 1 - j) for (i, j) in P.values())
    mdn = min(max(S - 1 - i, S - 1 - j) for (i, j) in N.values())
    score = 0
    score += (S - mdp) * (1 / len(P))
    score -= (S - mdn) * (1 / len(N))
    return score


def leaf_value(P, N, S, depth6):
    if (S - 1, S - 1) in P.values() or not N:
        return 10
    if (0, 0) in N.values() or not P:
        return -10
    e = hybrid(P, N, S)
    if not depth6:
        return e
    acc = 0
    for _ in range(6):  # the extra chance node of max_depth 6: evaluate() / 6 summed over the dice
        acc += e / 6
    return acc


def move(P, N, mover_is_p, k, dest):
    """envs/ewn.py:252-261: whatever stands on dest leaves the board, own cubes included"""
    P2 = {c: q for c, q in P.items() if q != dest}
    N2 = {c: q for c, q in N.items() if q != dest}
    (P2 if mover_is_p else N2)[k] = dest
    return P2, N2


def pair_for(side, d):
    """cubes a dice value selects, in list order: (F, G); cube numbers 1..6 (envs/ewn.py:144-176, 338-375)"""
    if d in side:
        return d, None
    up = next((k for k in range(d + 1, 7) if k in side), None)
    down = next((k for k in range(d - 1, 0, -1) if k in side), None)
    if up is not None:
        return up, down
    return down, None


def dests(pos, dirs, S):
    out = []
    for di, dj in dirs:
        i, j = pos[0] + di, pos[1] + dj
        out.append((i, j) if 0 <= i < S and 0 <= j < S else None)
    return out


def c2(P2, N2, S, beta, depth6):
    """chance node under a reply: sum over dice of the depth-1 max node's value / 6, the max node cut at beta"""
    # per cube of the searcher: the three prefix maxima of its leaves, a direction that leaves the board skipped
    pre = {}
    for k, pos in P2.items():
        best, seq = -INF, []
        for q in dests(pos, DIRS_P, S):
            if q is None:
                continue
            P3, N3 = move(P2, N2, True, k, q)
            best = max(best, leaf_value(P3, N3, S, depth6))
            seq.append(best)
        pre[k] = seq
    # key of a cube: (cut?, value) -- the first prefix maximum >= beta, else its overall maximum
    key = {}
    for k, seq in pre.items():
        cut = next((v for v in seq if v >= beta), None)
        key[k] = (True, cut) if cut is not None else (False, seq[-1])
    val = 0
    for d in range(1, 7):
        F, G = pair_for(P2, d)
        if key[F][0] or G is None:
            x = key[F][1]
        elif key[G][0]:
            x = key[G][1]
        else:
            x = max(key[F][1], key[G][1])
        val += x / 6
    return val


def chain(P1, N1, S, k, start, alpha, depth6):
    """min node restricted to cube k's replies, entered with running minimum `start`:
    -> (cut?, value): the value at which `worst <= alpha` stops the loop inside this cube, else the minimum reached"""
    worst = start
    for q in dests(N1[k], DIRS_N, S):
        if q is None:
            continue
        P2, N2 = move(P1, N1, False, k, q)
        if q == (0, 0) or not P2:
            val = -10
        else:
            val = c2(P2, N2, S, worst, depth6)
        if val < worst:
            worst = val
        if worst <= alpha:
            return True, worst
    return False, worst


def d5_closed_form(board, dice, depth):
    S = board.shape[0]
    depth6 = depth == 6
    P = {int(v): (i, j) for (i, j), v in np.ndenumerate(board) if v > 0}
    N = {int(-v): (i, j) for (i, j), v in np.ndenumerate(board) if v < 0}
    best, action = -INF, (0, 0)
    F0, G0 = pair_for(P, dice)
    roots = [(F0, 1 if F0 > dice else 0)] + ([(G0, 0)] if G0 is not None else [])
    for cube, flag in roots:
        for d, q in enumerate(dests(P[cube], DIRS_P, S)):
            if q is None:
                continue
            P1, N1 = move(P, N, True, cube, q)
            if q == (S - 1, S - 1) or not N1:
                v = 10
            else:
                alpha = best
                # every replier cube once as F, and once as G behind the next cube above it
                A, B = {}, {}
                for k in sorted(N1, reverse=True):
                    A[k] = chain(P1, N1, S, k, INF, alpha, depth6)
                    up = next((u for u in range(k + 1, 7) if u in N1), None)
                    if up is not None and not A[up][0]:
                        B[k] = chain(P1, N1, S, k, A[up][1], alpha, depth6)
                v = 0
                for d1 in range(1, 7):
                    F, G = pair_for(N1, d1)
                    if A[F][0] or G is None:
                        w = A[F][1]
                    else:
                        w = B[G][1]
                    v += w / 6
            if v > best:
                best, action = v, (flag, d)
    return action, best


def positions(S, n, seed, max_steps):
    env = po.OracleVecEnv(n, board_size=S, opponent="random", rng="philox", philox_key=seed)
    env.reset(np.arange(n, dtype=np.uint32) + seed)
    rs = np.random.RandomState(seed)
    stop = rs.randint(0, max_steps, n)
    keep_b, keep_d = env.obs()
    keep_b, keep_d = keep_b.copy(), keep_d.copy()
    for t in range(max_steps):
        b, d, r, te, tr, info = env.step(env.sample_legal_actions(t))
        live = (te == 0) & (stop > t)
        keep_b[live], keep_d[live] = b[live], d[live]
        if not live.any():
            break
    return keep_b, keep_d


@pytest.mark.parametrize("S,depth,n", [(5, 5, 60), (5, 6, 16), (6, 5, 12)])
def test_closed_form_equals_the_reference_recursion(S, depth, n):
    b, d = positions(S, n, 77 + S + depth, 12 if S == 5 else 18)
    oa, ov, _ = po.predict_minimax(b, d, depth, "hybrid")
    for i in range(n):
        a, v = d5_closed_form(b[i], int(d[i]), depth)
        assert (a[0], a[1]) == (int(oa[i][0]), int(oa[i][1])), (i, b[i], d[i])
        assert np.float64(v).tobytes() == np.float64(ov[i]).tobytes(), (i, v, ov[i])
