// ewn_step_d3.hpp -- the headline kernel: one fused env step with the depth-3 'hybrid'
// expectiminimax opponent (BASELINE config: 5x5, cube_layer 3), lean and occupancy-aware.
//
// Differences from the generic k_step (ewn_kernels.hip), same results bit for bit:
//  * the whole step lives in the opponent's CANONICAL view, ring-ordered (ewn_fast.hpp):
//    the board bytes are decoded straight into that space with compile-time bit positions
//    (real cell c -> canonical cell S*S-1-c -> ring index), so nothing is converted before
//    the search.  The agent is the canonical BOTTOM_RIGHT side ("replier" geometry), the
//    opponent the canonical TOP_LEFT side; an action [flag, dir] means the same in both
//    views (envs/ewn.py:289-296);
//  * T lanes cooperate on one game (the code is written for T = 1, 2, 4; the library instantiates 1 and 2): each computes a share of the 108 leaves
//    and of the 36 (root, dice) scans and they swap results with DPP lane permutes inside
//    the wavefront.  At 65 536 lanes a thread per game is ONE wave per SIMD; T lanes per
//    game give T waves per SIMD, which is what hides LDS/global latency and doubles the
//    VALU issue rate.  Everything not split is computed redundantly (identically) by the
//    T lanes; only sub-lane 0 stores.
#pragma once
#include "ewn_fast.hpp"

// ---- compile-time ring geometry (same order as build_fast_tables: ascending min(row,col), row-major inside a level)
template <int S>
struct RingGeo {
    int ring_of_rm[S * S];   // canonical row-major cell -> ring index
    int rm_of_ring[S * S];
    constexpr RingGeo() : ring_of_rm(), rm_of_ring()
    {
        int n = 0;
        for (int t = 0; t < S; t++)
            for (int i = 0; i < S; i++)
                for (int j = 0; j < S; j++)
                    if ((i < j ? i : j) == t) { ring_of_rm[i * S + j] = n; rm_of_ring[n] = i * S + j; n++; }
    }
};

// Quad-permute DPP (a VALU operand modifier: no LDS round trip).  The T lanes of a game are T consecutive
// lanes inside one quad (T <= 4), so "lane j of my group" is a quad_perm broadcast.
template <int CTRL> EWN_DEV u32 dpp_u32(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> EWN_DEV double dpp_f64(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = dpp_u32<CTRL>((u32)b), hi = dpp_u32<CTRL>((u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
// broadcast from lane J of a group of T lanes
template <int T, int J> struct Bcast {
    // T == 4: quad_perm [J,J,J,J]; T == 2: pairs (0,1) and (2,3): [J,J,2+J,2+J]
    static constexpr int CTRL = T == 4 ? (J | (J << 2) | (J << 4) | (J << 6)) : (J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6));
};
// xor-1 / xor-2 lane swaps inside a quad
#define DPP_XOR1 0xB1 /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E /* quad_perm [2,3,0,1] */

// A side's six cubes: byte k of a u64 = ring index of cube k (0-based), bit 6 set once the cube is off the board (the low six
// bits are then stale); bytes 6 and 7 are permanently "off the board", which is what a table answers "no such cube" with.
EWN_DEV int pk_get(u64 w, int k) { return (int)((w >> (8 * k)) & 0xFFull); }
#define PK_OFF 0x40   // flag in a position byte
#define PK_PADS (0x4040ull << 48)

template <int S>
struct RState {                 // one game, canonical ring space
    typename MaskOf<S>::type P, N; // P: canonical TOP_LEFT (the opponent), N: canonical BOTTOM_RIGHT (the agent)
    u64 posP, posN;
};

// bit k = cube k is on the board
EWN_DEV u32 pk_alive(u64 w)
{
    const u32 t = (~(u32)w & 0x40404040u) >> 6, u = ~(u32)(w >> 32);   // t: bits 0, 8, 16, 24
    return ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) | ((u >> 2) & 0x10u) | ((u >> 9) & 0x20u);
}

// FastTab::sel entry for the side whose cubes are pos: which cube(s) may move for this dice
template <int S>
EWN_DEV u32 pk_sel(const FastTab<S> *Tb, u64 pos, int dice) { return Tb->sel[pk_alive(pos) * 8u + ((u32)(dice - 1) & 7u)]; }
// ring bytes of the first (bits 0-7) and second (bits 8-15) cube of a sel entry; "none" reads a permanently-off byte
EWN_DEV u32 pk_pair(u64 pos, u32 e) { return __builtin_amdgcn_perm((u32)(pos >> 32), (u32)pos, (e & 0x0707u) | 0x0C0C0000u); }
// find_cube_to_move (envs/ewn.py:178-215): the dice cube if it is on the board (then it is the only entry); else the
// requested neighbour if it exists, else the other one; 6 when the side has no cube at all
EWN_DEV int pk_cube(u32 e, bool larger) { const int first = (int)(e & 7u), second = (int)((e >> 8) & 7u); return (larger || second == 6) ? first : second; }

// whatever stands on ring cell q leaves the board (envs/ewn.py:254-258): byte == q  <=>  bit 7 of ((byte ^ q ^ 0x7F) + 1);
// bytes never reach 0x80, so the additions do not carry between bytes
EWN_DEV u64 pk_capture(u64 w, u32 qb)
{
    u32 lo = (u32)w, hi = (u32)(w >> 32);
    lo |= (((lo ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    hi |= (((hi ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    return ((u64)hi << 32) | lo;
}

// move cube k of one side to ring cell q (capture whatever is there, own cube included: envs/ewn.py:252-261)
template <int S, bool MOVER_IS_P>
EWN_DEV void rs_move(RState<S> &s, int k, int q)
{
    typedef typename MaskOf<S>::type M;
    const M one = 1, bq = one << q;
    const int p = pk_get(MOVER_IS_P ? s.posP : s.posN, k) & 63;
    const u32 q7 = (u32)q ^ 0x7Fu, qb = __builtin_amdgcn_perm(q7, q7, 0u);
    s.posP = pk_capture(s.posP, qb);
    s.posN = pk_capture(s.posN, qb);
    const u64 mv = (u64)(u32)(p ^ q) << (8 * k);
    if (MOVER_IS_P) { s.posP ^= mv; s.P = (s.P & ~(one << p)) | bq; s.N &= ~bq; }
    else { s.posN ^= mv; s.N = (s.N & ~(one << p)) | bq; s.P &= ~bq; }
}

// max_depth 5 (and, on the second table image, 6): the reference's loops as they stand (classical_policies/minimax.py:19-73) --
// root move, chance, reply, chance, move, leaf -- with its alpha-beta windows passed through the chance nodes unchanged, on the
// byte-per-cube state and the table leaf.  One lane per game; breaks diverge between the lanes of a wave, trip counts are
// bounded (6^5 leaves).  A leaf is a rank lookup, an inner node a masked move: ~20x fewer instructions than the generic
// template recursion on GState.
template <int S, int T = 1, bool H2 = false>
__device__ __forceinline__ double d5_search(const FastTab<S> *Tb, const RState<S> &c, int dice, int sub, int &bflag, int &bdir)
{
    static_assert(T == 1 || T == 2, "the six dice of the inner chance node are split over one or two lanes");
    typedef typename MaskOf<S>::type M;
    const M one = 1;
    const double inf = __builtin_inf();
    const u32 rank10 = 8u * (u32)Tb->nv; // +10 is the largest value of the table (ranks travel as byte offsets, ewn_fast.hpp)
    double best = -inf, alpha = -inf;
    bflag = 0; bdir = 0;
    const u32 e0 = pk_sel<S>(Tb, c.posP, dice), pp0 = pk_pair(c.posP, e0);
    #pragma unroll 1
    for (int r = 0; r < 6; r++) {
        const int slot = r >= 3 ? 1 : 0, dir = r - 3 * slot;
        const int cube = (int)((slot ? e0 >> 8 : e0) & 7u), rb = (int)((slot ? pp0 >> 8 : pp0) & 0xFFu);
        const int dest = Tb->nbp[dir][rb];
        if (dest == 255) continue;                     // no such cube (byte 6) or off the board
        RState<S> s1 = c;
        rs_move<S, true>(s1, cube, dest);
        double v;
        if (dest == FastTab<S>::CELLS - 1 || s1.N == 0) v = 10.0;
        else {
            v = 0.0;
            #pragma unroll 1
            for (int d1 = 1; d1 <= 6; d1++) {          // chance node, depth 4
                const u32 e1 = pk_sel<S>(Tb, s1.posN, d1), pp1 = pk_pair(s1.posN, e1);
                double worst = inf, beta = inf;        // min node, depth 3: (alpha from the root, beta = +inf)
                #pragma unroll 1
                for (int q = 0; q < 6; q++) {
                    const int qs = q >= 3 ? 1 : 0, qd = q - 3 * qs;
                    const int qc = (int)((qs ? e1 >> 8 : e1) & 7u), qb = (int)((qs ? pp1 >> 8 : pp1) & 0xFFu);
                    const int dn = Tb->nbn[qd][qb];
                    if (dn == 255) continue;
                    RState<S> s2 = s1;
                    rs_move<S, false>(s2, qc, dn);
                    double val;
                    if (dn == Tb->ri_origin || s2.P == 0) val = -10.0;
                    else {
                        val = 0.0;
                        u32 br[6 / T];                       // my dice: d2 = sub + 1, sub + 1 + T, ...
                        #pragma unroll
                        for (int j = 0; j < 6 / T; j++) {   // chance node, depth 2 (the T lanes of the game share its six dice)
                            const int d2 = sub + 1 + T * j;
                            const u32 e2 = pk_sel<S>(Tb, s2.posP, d2), pp2 = pk_pair(s2.posP, e2);
                            // max node, depth 1.  All (at most six) leaves first, their LDS reads batched in three stages, then the
                            // reference's loop in registers: a move that does not exist is skipped; after each real move
                            // best = max(best, leaf), alpha' = max(alpha, best), stop once beta <= alpha'.
                            u32 lx[6], ly[6], lr[6];
                            bool ex[6], won[6];
                            #pragma unroll
                            for (int m = 0; m < 6; m++) {
                                const int ms = m >= 3 ? 1 : 0, md = m - 3 * ms;
                                const int mb = (int)((ms ? pp2 >> 8 : pp2) & 0xFFu);
                                const int dp = Tb->nbp[md][mb];
                                ex[m] = dp != 255;
                                const M bd = ex[m] ? (one << (dp & 63)) : (M)0;
                                const M P3 = (s2.P & ~(one << (mb & 63))) | bd, N3 = s2.N & ~bd;
                                won[m] = dp == FastTab<S>::CELLS - 1 || N3 == 0;          // evaluate() of a won position: +10
                                lx[m] = ft_side_h<S, H2>(Tb, P3);
                                ly[m] = ft_side_h<S, H2>(Tb, N3);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            #pragma unroll
                            for (int m = 0; m < 6; m++) lr[m] = ft_rank8<S>(Tb, ft_addr(lx[m], ly[m]));
                            __builtin_amdgcn_sched_barrier(0);
                            double lv[6];
                            #pragma unroll
                            for (int m = 0; m < 6; m++) { lr[m] = won[m] ? rank10 : lr[m]; lv[m] = ft_val<S>(Tb, lr[m]); }
                            __builtin_amdgcn_sched_barrier(0);
                            u32 bestr = 0;                   // rank 0 = -inf
                            double bestv = -inf;
                            bool stop = false;
                            #pragma unroll
                            for (int m = 0; m < 6; m++) {
                                const bool live = ex[m] && !stop;
                                bestr = live ? max(bestr, lr[m]) : bestr;
                                bestv = live ? fmax(bestv, lv[m]) : bestv;
                                stop = stop || (live && beta <= fmax(alpha, bestv));
                            }
                            br[j] = bestr;
                        }
                        u32 bd6[6];
                        #pragma unroll
                        for (int j = 0; j < 6 / T; j++) {
                            if constexpr (T == 1) bd6[j] = br[j];
                            else { bd6[2 * j] = dpp_u32<Bcast<2, 0>::CTRL>(br[j]); bd6[2 * j + 1] = dpp_u32<Bcast<2, 1>::CTRL>(br[j]); }
                        }
                        double q6[6];
                        #pragma unroll
                        for (int d = 0; d < 6; d++) q6[d] = ft_val6<S>(Tb, bd6[d]);
                        #pragma unroll
                        for (int d = 0; d < 6; d++) val = val + q6[d];   // expected_val += val / 6 in dice order
                    }
                    if (val < worst) worst = val;
                    beta = fmin(beta, worst);
                    if (beta <= alpha) break;
                }
                v = v + worst / 6.0;
            }
        }
        if (v > best) { best = v; bflag = slot == 0 ? (int)(e0 >> 15) : 0; bdir = dir; }
        alpha = fmax(alpha, best);
    }
    return best;
}

// Wave priority inside the search (s_setprio): the two waves that share a SIMD run the same loop, and while one of them is in the
// front half of a root iteration -- address arithmetic and three rounds of table reads -- every issue slot it loses to its partner
// delays reads whose latency it then sits out.  Raised priority there, normal priority in the cut-off replay (long dependent
// select chains that tolerate yielding): 9.13 -> 8.80 us per step of the 50-step rollout, 13.9 -> 13.3 us for the one-step kernel
// (levels 1, 2 and 3 measure the same; 2 keeps the MT19937 refill waves, which run at 3, in front).
#define D3_PRIO_HI() __builtin_amdgcn_s_setprio(2)
#define D3_PRIO_LO() __builtin_amdgcn_s_setprio(0)
// keeps the stages of a root apart in the ROLLED root loop (without it the register-pressure-driven schedule re-serialises the LDS
// reads into one round trip per leaf).  The slot-task kernel's three roots are unrolled (PERLANE): there the compiler's own schedule is
// the better one (measured round 3: 6.53 -> 6.46 us per env step without the fences; without the priority switches 6.74)
#define D3_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define D3_SEARCH_FENCE() do { if constexpr (!PERLANE) __builtin_amdgcn_sched_barrier(0); } while (0)   // inside d3_search
#define D3_CUT 0x4000u    // key flag of d3_search: the reply loop stops inside this cube's replies (byte-offset ranks stay below 0x2000)

// element `sub + T * i` of a six-element array held identically by the T lanes of a group, for the lane with index `sub`
template <int T> EWN_DEV u32 own_of(const u32 (&v)[6], int i, int sub)
{
    if constexpr (T == 1) return v[i];
    else if constexpr (T == 2) return sub ? v[2 * i + 1] : v[2 * i];
    else {
        if (i == 0) { const u32 lo = (sub & 1) ? v[1] : v[0], hi = (sub & 1) ? v[3] : v[2]; return (sub & 2) ? hi : lo; }
        return (sub & 1) ? v[5] : v[4];   // lanes 2 and 3 have no second dice (6 % 4 = 2): their value is never used
    }
}

// every lane of a group of T gets all T lanes' `mine`: out[T * i + j] = lane j's value
template <int T> EWN_DEV void publish(u32 mine, int i, u32 (&out)[6])
{
    if constexpr (T == 1) out[i] = mine;
    else if constexpr (T == 2) { out[2 * i] = dpp_u32<Bcast<2, 0>::CTRL>(mine); out[2 * i + 1] = dpp_u32<Bcast<2, 1>::CTRL>(mine); }
    else {
        if (i == 0) { out[0] = dpp_u32<Bcast<4, 0>::CTRL>(mine); out[1] = dpp_u32<Bcast<4, 1>::CTRL>(mine);
                      out[2] = dpp_u32<Bcast<4, 2>::CTRL>(mine); out[3] = dpp_u32<Bcast<4, 3>::CTRL>(mine); }
        else { out[4] = dpp_u32<Bcast<4, 0>::CTRL>(mine); out[5] = dpp_u32<Bcast<4, 1>::CTRL>(mine); }
    }
}

// Depth-3 search in ring space ('hybrid', 'min_dist' or 'attk' by table image), shared by T lanes (sub = my index in the group).
// Every lane of the group returns the same (value, action).
//
// Instruction budget (round 2, measured with tools/valu_probe.hip: only and/or/xor/add/mov issue at the full rate, everything
// else -- shifts by a variable, v_cndmask, v_bcnt, v_ffbh, DPP, compares -- at about half of it, so the loop is written to need
// few instructions of the second kind): per leaf two mask operations, two v_ffbh, two byte reads of the level table (no address
// arithmetic), two v_bcnt with the level as their accumulate operand, a shift + v_lshl_or for the byte address of the rank, one
// v_and_or for the root-invariant exceptions; ranks travel as byte offsets (8 x rank) so that value reads need no shift either;
// a cube's result is ONE 16-bit key (the cut-off value if its replies cut, else 0x8000 | minimum) so that the pair selection per
// dice is one min and a select (a cut carries the flag D3_CUT, dropped on the copy that acts as F); the lanes exchange keys and
// chosen ranks, never doubles.
// PERLANE (the slot-task rollout kernel, ewn_rollout.hpp): one call searches the three roots of ONE root cube -- slotL = 0: the
// first cube of the legal list, 1: the second -- and carries best / bflag / bdir in and out, so that a game whose dice selects a
// single cube (most do) is finished after one call; returns through `have_second` whether the list has a second cube.
template <int S, int T, bool H2 = false, bool PERLANE = false>
__device__ __forceinline__ double d3_search(const FastTab<S> *Tb, const RState<S> &c, int dice, int sub, int depth, int &bflag, int &bdir,
                                            int slotL = 0, double best_in = 0.0, bool *have_second = nullptr)
{
    typedef typename MaskOf<S>::type M;
    constexpr int KPT = 6 / T + (6 % T ? 1 : 0); // replier cubes / dice values per lane: k = sub + T*i
    const M one = 1;

    // root slots: <= 2 cubes x 3 dirs in the reference's list order (envs/ewn.py:338-375)
    const u32 rsel = pk_sel<S>(Tb, c.posP, dice), rpp = pk_pair(c.posP, rsel);
    const int rb0 = (int)(rpp & 0xFFu), rb1 = (int)(rpp >> 8);
    const bool have0 = !(rb0 & PK_OFF), have1 = !(rb1 & PK_OFF);
    const int flag0 = (int)(rsel >> 15);
    const int rp0 = rb0 & 63, rp1 = rb1 & 63;

    double best = -__builtin_inf(); // alpha = max(alpha, best_val): the running best (root beta stays +inf)
    if constexpr (PERLANE) {
        *have_second = have1;
        if (slotL == 0) { bflag = 0; bdir = 0; } else best = best_in;
    } else { bflag = 0; bdir = 0; }

    if (depth < 3) {
        // max_depth 1 and 2 bottom out before the replier moves (classical_policies/minimax.py:19-73: every node, chance nodes
        // included, spends one unit of depth): depth 1 evaluates the position after the root move at the chance node; depth 2
        // at the six min nodes below it, i.e. evaluate()/6 summed six times in dice order.  Every lane of the group does all six roots.
        #pragma unroll 1
        for (int r = 0; r < 6; r++) {
            const int slot = r >= 3 ? 1 : 0, dir = r - 3 * slot;
            const int rp = slot == 0 ? rp0 : rp1;
            const int dest = Tb->nbp[dir][rp];
            const bool valid = (slot == 0 ? have0 : have1) && dest != 255;
            const M bd = valid ? (one << dest) : (M)0;
            const M P1 = (c.P & ~(one << rp)) | bd;
            const M N1 = c.N & ~bd;
            const bool term = dest == FastTab<S>::CELLS - 1 || N1 == 0; // win(B1): value 10 at any depth
            const u32 rk = ft_rank8<S>(Tb, ft_addr(ft_side_h<S, H2>(Tb, P1), ft_side_h<S, H2>(Tb, N1)));
            const double e1 = ft_val<S>(Tb, rk), e6 = ft_val6<S>(Tb, rk);
            double v = 0.0;
            #pragma unroll
            for (int d = 0; d < 6; d++) v = v + e6;
            v = term ? 10.0 : (depth == 1 ? e1 : v);
            if (valid && v > best) { best = v; bflag = slot == 0 ? flag0 : 0; bdir = dir; }
        }
        if constexpr (PERLANE) *have_second = false; // both cubes were searched in this one call
        return best;
    }

    // my share of the replier's (cube, dir) moves; they never change during the search
    M rset[KPT][3], rclr[KPT];
    int rnk[KPT];
    // per reply, root-invariant: byte address of the rank = (address & keep) | fixed.  An illegal reply reads rank[0] ("no such
    // reply", +inf), a reply onto the origin rank[1] (-10, envs/minimax_ewn.py:45-47).  The exceptions steer the ADDRESS instead
    // of selecting the loaded value: a select on a loaded value makes the compiler branch around the LDS reads.
    u32 keep[KPT][3], fixed[KPT][3];
    u32 mine_alive = 0;
    #pragma unroll
    for (int i = 0; i < KPT; i++) {
        const int k = sub + T * i;          // may be >= 6 for the last slot when 6 % T != 0: byte 6/7 = a cube that is off the board
        const int rb = pk_get(c.posN, k < 6 ? k : 6);
        rnk[i] = rb & 63;
        rclr[i] = ~(one << rnk[i]);
        mine_alive |= ((rb & PK_OFF) ? 0u : 1u) << i;
        #pragma unroll
        for (int d = 0; d < 3; d++) {
            const u32 kf = Tb->rkf[d][rb];
            rset[i][d] = Tb->rsetT[d][rb];
            keep[i][d] = kf & 0xFFFFu;    // rank addresses stay below 2^13
            fixed[i][d] = kf >> 16;
        }
    }

    // the six root destinations, read before the loop: inside it the read's whole latency stood in front of every root
    const u32 dq0 = (u32)Tb->nbp[0][rp0] | ((u32)Tb->nbp[1][rp0] << 8) | ((u32)Tb->nbp[2][rp0] << 16);
    const u32 dq1 = (u32)Tb->nbp[0][rp1] | ((u32)Tb->nbp[1][rp1] << 8) | ((u32)Tb->nbp[2][rp1] << 16);

    // a real loop, not unrolled: the body is ~300 instructions and six copies of it (plus the rest of the
    // kernel) do not fit the instruction cache shared by two CUs
    // (the slot-task kernel's three roots ARE unrolled: 24 KB of code still fit, the direction becomes a constant: -1.5 %, 163 -> 152 VGPRs)
    constexpr int ROOT_UNROLL = PERLANE ? 3 : 1;
    #pragma unroll ROOT_UNROLL
    for (int r = 0; r < (PERLANE ? 3 : 6); r++) {
        const int slot = PERLANE ? slotL : (r >= 3 ? 1 : 0), dir = PERLANE ? r : r - 3 * slot;
        const int rp = slot == 0 ? rp0 : rp1;
        const int dest = (int)(((slot == 0 ? dq0 : dq1) >> (8 * dir)) & 0xFFu);
        const bool valid = (slot == 0 ? have0 : have1) && dest != 255;
        const M bd = valid ? (one << dest) : (M)0;
        const M P1 = (c.P & ~(one << rp)) | bd; // own capture: the bit is already set, the count drops by itself
        const M N1 = c.N & ~bd;
        const bool term = dest == FastTab<S>::CELLS - 1 || N1 == 0; // win(B1): value 10

        // Four stages, each issuing ALL its LDS reads before the next one consumes them (sched_barrier keeps the compiler from
        // re-serialising them into one round trip per leaf, which is what the register-pressure-driven schedule does):
        // levels -> ranks -> values of the three prefix minima -> cut.
        u32 tr[6];
        constexpr int CH = KPT > 3 ? 3 : KPT; // cubes staged together: 9 leaves in flight; more only costs registers
        #pragma unroll
        for (int i0 = 0; i0 < KPT; i0 += CH) {
            D3_PRIO_HI();
            M P2[CH][3], N2[CH][3];
            u32 lp[CH][3], ln[CH][3], a[CH][3];
            #pragma unroll
            for (int ii = 0; ii < CH; ii++) {
                const M Nk = N1 & rclr[i0 + ii];
                #pragma unroll
                for (int d = 0; d < 3; d++) {
                    N2[ii][d] = Nk | rset[i0 + ii][d];
                    P2[ii][d] = P1 & ~rset[i0 + ii][d];
                    lp[ii][d] = Tb->lvl[lvl_index(P2[ii][d])]; // P2 == 0 (last cube captured): level 0 + count 0 = a row of -10
                    ln[ii][d] = Tb->lvl[lvl_index(N2[ii][d])]; // N2 holds the moved cube unless the reply is absent (address masked)
                    if constexpr (H2) { // 'two_min_dist': distance of the highest cell + distance of the second highest
                        lp[ii][d] += Tb->lvl[lvl_index(drop_top(P2[ii][d]))];
                        ln[ii][d] += Tb->lvl[lvl_index(drop_top(N2[ii][d]))];
                    }
                }
            }
            D3_SEARCH_FENCE();
            #pragma unroll
            for (int ii = 0; ii < CH; ii++) {
                #pragma unroll
                for (int d = 0; d < 3; d++) {
                    const u32 ix = H2 ? lp[ii][d] : lp[ii][d] + (u32)popc_m(P2[ii][d]), iy = H2 ? ln[ii][d] : ln[ii][d] + (u32)popc_m(N2[ii][d]);
                    a[ii][d] = ft_rank8<S>(Tb, (ft_addr(ix, iy) & keep[i0 + ii][d]) | fixed[i0 + ii][d]);
                }
            }
            D3_SEARCH_FENCE();
            u32 p1[CH], p2[CH];
            double va[CH], v1[CH], v2[CH];
            #pragma unroll
            for (int ii = 0; ii < CH; ii++) {
                p1[ii] = min(a[ii][0], a[ii][1]); p2[ii] = min(p1[ii], a[ii][2]);
                va[ii] = ft_val<S>(Tb, a[ii][0]); v1[ii] = ft_val<S>(Tb, p1[ii]); v2[ii] = ft_val<S>(Tb, p2[ii]); // +inf for "no such reply"
            }
            D3_SEARCH_FENCE();
            // (the priority stays raised through the cut-off replay and the expectation below, until the search returns: the wave that
            // shares the SIMD is then mostly in the latency-bound halves of its env step.  Lowering it here, for the arithmetic part of
            // a root, was round 2's choice; measured round 3 on the final loop: 6.46 -> 6.20 us per env step with the raise kept)
            #pragma unroll
            for (int ii = 0; ii < CH; ii++) {
                const int i = i0 + ii;
                // per cube: `cut` = the value at which the reference's reply loop would stop inside this cube's replies
                // (`worst <= alpha`, minimax.py:59-61; alpha = best so far), 0 if it would not.  The running minimum along a
                // cube's replies is non-increasing (a0 >= p1 >= p2), so the loop stops at the first of them that is <= alpha.
                // (three independent selects, last one wins: written as a nested conditional the compiler branches on it)
                // The cube's key: D3_CUT | its cut value if it cuts, else 0x8000 | its minimum.  A cube that is off the board (or was just
                // captured by the root move) reads garbage leaves above: its key is forced to "no such cube" here.
                u32 key = 0x8000u | p2[ii];
                key = v2[ii] <= best ? (p2[ii] | D3_CUT) : key;
                key = v1[ii] <= best ? (p1[ii] | D3_CUT) : key;
                key = va[ii] <= best ? (a[ii][0] | D3_CUT) : key;
                const bool there = ((mine_alive >> i) & 1u) && rnk[i] != dest;
                key = there ? key : FAST_KNONE;
                publish<T>(key, i, tr); // after this every lane holds tr[k] for all six cubes (cube k = j + T*i lives in lane j)
            }
        }
        // which cubes a dice value selects (find_near_cube): carry the nearest on-board cube's key along
        // A key that cuts carries D3_CUT (ranks stay below it, "does not cut" keys above it); the copy that travels as "nearest cube
        // ABOVE" (the F of a pair) drops the flag, which puts a cutting F below everything else: the pair's result is then ONE min.
        u32 upT[6], downT[6];
        {
            u32 cur = FAST_KNONE;
            #pragma unroll
            for (int d = 5; d >= 0; d--) { upT[d] = cur; cur = tr[d] != FAST_KNONE ? (tr[d] & ~D3_CUT) : cur; }
            cur = FAST_KNONE;
            #pragma unroll
            for (int d = 0; d < 6; d++) { downT[d] = cur; cur = tr[d] != FAST_KNONE ? tr[d] : cur; }
        }
        // my share of the six chance branches: dice index d = sub + T*i.  Reply order is the larger-neighbour cube F, then
        // the smaller-neighbour cube G (get_legal_actions): the loop stops in F if F cuts (key < 0x8000: the result is F's cut
        // value whatever G is), else in G if G cuts, else it returns the minimum over both -- min(F, G) & 0x7fff in either case.
        u32 wq[6];
        #pragma unroll
        for (int i = 0; i < KPT; i++) {
            const u32 e = own_of<T>(tr, i, sub), u = own_of<T>(upT, i, sub), dn = own_of<T>(downT, i, sub);
            // the dice cube alone if it is on the board; else F = up, G = down: F's cut value if F cuts (flag dropped: the smallest),
            // else G's if G cuts (flagged: below every key that does not cut), else the smaller minimum; no cube above: down alone
            const u32 w = (e != FAST_KNONE ? e : min(u, dn)) & 0x1FFFu;
            publish<T>(w, i, wq);
        }
        double v = 0.0;
        #pragma unroll
        for (int d = 0; d < 6; d++) v = v + ft_val6<S>(Tb, wq[d]); // expected_val += val / 6 in dice order, minimax.py:72
        v = term ? 10.0 : v;
        if (valid && v > best) { best = v; bflag = slot == 0 ? flag0 : 0; bdir = dir; }
    }
    D3_PRIO_LO();
    return best;
}

#include "ewn_search_d5.hpp"

// max_depth 5 / 6 on the image the search was built for: the closed form for the (level, count) images, the loops for 'two_min_dist'
template <int S, int T, bool H2 = false>
__device__ __forceinline__ double d5_dispatch(const FastTab<S> *Tb, const RState<S> &c, int dice, int sub, int &bflag, int &bdir)
{
    if constexpr (H2) return d5_search<S, T, true>(Tb, c, dice, sub, bflag, bdir);
    else return d5c_search<S, T>(Tb, c, dice, sub, bflag, bdir);
}

// The same search from a row-major canonical GState (stateless predict kernel, generic step kernel's fast path)
template <int S, bool H2 = false>
__device__ __forceinline__ double fast_d3(const FastTab<S> *Tb, const GState<1> &c, int dice, int depth, int &bflag, int &bdir)
{
    typedef typename MaskOf<S>::type M;
    RState<S> s;
    s.P = 0; s.N = 0; s.posP = s.posN = PK_PADS;
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        const int rp = Tb->ri[pos_get<1>(c.posP, k)], rn = Tb->ri[pos_get<1>(c.posN, k)];
        const bool ap = (c.aliveP >> k) & 1u, an = (c.aliveN >> k) & 1u;
        if (ap) s.P |= (M)1 << rp;
        if (an) s.N |= (M)1 << rn;
        s.posP |= (u64)(ap ? rp : PK_OFF) << (8 * k);
        s.posN |= (u64)(an ? rn : PK_OFF) << (8 * k);
    }
    if (depth >= 5) return d5_dispatch<S, 1, H2>(Tb, s, dice, 0, bflag, bdir);
    return d3_search<S, 1, H2>(Tb, s, dice, 0, depth, bflag, bdir);
}

// ---------------------------------------------------------------- the fused step kernel

struct D3Cfg {
    int N, rng_kind, autoreset, lane_offset, depth, refill_blocks; // depth: the opponent's max_depth (1..3); the first refill_blocks blocks of the grid rebuild MT windows (fused MT path), else 0
    u32 seed_stride, W;
    double reward;
    u64 key;
    int shaped, refresh;     // MiniMaxHeuristicEnv.step (envs/training_ewn.py:43-99); refresh: prev_score re-evaluated on auto-reset
    double illegal_reward;
};

struct D3Buf {
    int8_t *board; int8_t *dice; uint8_t *done; u32 *rng; const void *tables; const int8_t *actions;
    double *reward; uint8_t *terminated; uint8_t *truncated; uint8_t *info; int8_t *tboard; int8_t *tdice; int8_t *ract;
    void *mtq; // MT kind with auto-reset: the refill hand-off area (scratch), else NULL
    double *prev_score; int32_t *tolerance; // shaped env only
};

// MinimaxEnv.evaluate('hybrid') of the REAL position as the shaped training env calls it (agent_player = TOP_LEFT,
// envs/training_ewn.py:94-96, envs/minimax_ewn.py:56-86), for a position that is not won or lost: both sides' distance to the real
// bottom-right corner = max(row, col) of the canonical square (table dtl), fp64 in the reference's order of operations.
// The agent is the canonical BOTTOM_RIGHT side (posN), the opponent the canonical TOP_LEFT side (posP).
template <int S>
EWN_DEV double d3_shaped_score(const FastTab<S> *Tb, const RState<S> &s)
{
    u32 da[6], dq[6];
    #pragma unroll
    for (int k = 0; k < 6; k++) { da[k] = Tb->dtl[pk_get(s.posN, k) & 63]; dq[k] = Tb->dtl[pk_get(s.posP, k) & 63]; }
    u32 mda = 255u, mdo = 255u;
    int na = 0, no = 0;
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        const bool aa = !(pk_get(s.posN, k) & PK_OFF), ao = !(pk_get(s.posP, k) & PK_OFF);
        mda = aa ? min(mda, da[k]) : mda; na += aa ? 1 : 0;
        mdo = ao ? min(mdo, dq[k]) : mdo; no += ao ? 1 : 0;
    }
    const double a = (double)(S - (int)mda) * (1.0 / (double)na);   // score += (L - md+) * (1 / n+)
    const double b = (double)(S - (int)mdo) * (1.0 / (double)no);   // score -= (L - md-) * (1 / n-)
    return (0.0 + a) - b;
}

#ifndef D3_BS
#define D3_BS 256
#endif

// bytes of STATIC LDS of k_step_d3<S, T, ., RNGK> (boards + terminal boards | table image | 16 bytes per game), 0 = the instance
// takes its LDS dynamically (MT kind; blocks that would not fit the 64 KB static limit)
template <int S, int T, int RNGK>
constexpr int d3_lds_static()
{
    const int sz = ((2 * (D3_BS / T) * S * S + 15) & ~15) + FAST_TAB_BYTES(S) + (D3_BS / T) * 16;
    return (RNGK == 1 && sz <= 60 * 1024) ? sz : 0;
}
#ifndef D3_REFILL_PRIO
#define D3_REFILL_PRIO 3
#define D3_STEP_PRIO 0
#endif

// real board bytes -> canonical ring state.  Real value v > 0 is the agent's cube v (canonical BOTTOM_RIGHT side),
// v < 0 the opponent's cube |v| (canonical TOP_LEFT side); real cell c sits at canonical cell S*S-1-c, whose ring
// index is a compile-time constant.  The T lanes of a game scan interleaved cells: a cell's ring index is scattered
// into byte (v + 8) of the game's 16 LDS bytes ga (7..2 = opponent cubes 1..6, 8 = empty cells, 9..14 = agent cubes 1..6),
// which every lane then reads back whole; the occupancy masks are OR-ed across the lanes with DPP.  LDS operations of
// one wave execute in program order, so the T lanes (same wave) need no barrier.
template <int S, int T>
EWN_DEV void d3_decode(const int8_t *b, int sub, uint8_t *ga, RState<S> &s)
{
    typedef typename MaskOf<S>::type M;
    constexpr RingGeo<S> G{};
    constexpr int NC = (S * S + T - 1) / T;
    int v[NC]; // my cells first, all reads in flight together: the compiler cannot move a read of b across a write to ga
    #pragma unroll
    for (int i = 0; i < NC; i++) { const int c = i * T + sub; const bool in = c < S * S; v[i] = in ? (int)b[in ? c : 0] : 0; }
    *(uint4 *)ga = make_uint4(0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u); // every cube off the board
    __builtin_amdgcn_wave_barrier();
    M P = 0, N = 0;
    #pragma unroll
    for (int i = 0; i < NC; i++) {
        // ring index of my i-th cell: a compile-time constant per (i, sub); select among the T candidates
        int ring = 0;
        #pragma unroll
        for (int j = 0; j < T; j++) { const int c = i * T + j; if (c < S * S && (T == 1 || sub == j)) ring = G.ring_of_rm[S * S - 1 - c]; }
        ga[(v[i] + 8) & 15] = (uint8_t)ring;                                        // an empty (or out-of-range) cell hits the spare byte
        const M mpos = (M)(long long)((0 - v[i]) >> 31), mneg = (M)(long long)(v[i] >> 31); // all ones where v > 0 / v < 0
        N |= ((M)1 << ring) & mpos;
        P |= ((M)1 << ring) & mneg;
    }
    __builtin_amdgcn_wave_barrier();
    const uint4 x = *(const uint4 *)ga;
    __builtin_amdgcn_wave_barrier();
    // bytes 6 and 7 are forced "off the board" whatever a malformed board (|v| > 6) scattered there
    s.posN = ((((u64)x.w << 32) | x.z) >> 8) | PK_PADS;                                      // bytes 9..14 -> 0..5
    s.posP = (((u64)__builtin_bswap32(x.x) << 32) | __builtin_bswap32(x.y)) | PK_PADS;       // bytes 7..2 -> 0..5, 1..0 -> 6..7
    if constexpr (T >= 2) {
        #define D3_OR64(x, ctrl) x |= ((u64)dpp_u32<ctrl>((u32)(x >> 32)) << 32) | dpp_u32<ctrl>((u32)x)
        #define D3_ORM(x, ctrl) do { if constexpr (sizeof(M) == 4) x |= (M)dpp_u32<ctrl>((u32)x); else { u64 t_ = (u64)x; D3_OR64(t_, ctrl); x = (M)t_; } } while (0)
        D3_ORM(P, DPP_XOR1); D3_ORM(N, DPP_XOR1);
        if constexpr (T == 4) { D3_ORM(P, DPP_XOR2); D3_ORM(N, DPP_XOR2); }
        #undef D3_ORM
        #undef D3_OR64
    }
    s.P = P; s.N = N;
}

// canonical ring state -> real board bytes (LDS), the T lanes writing interleaved shares
template <int S, int T>
EWN_DEV void d3_encode(const FastTab<S> *Tb, const RState<S> &s, int sub, int8_t *b)
{
    #pragma unroll
    for (int c = 0; c < S * S; c++) if (T == 1 || c % T == sub) b[c] = 0;
    // the table reads first (all in flight together), then the stores
    int cell[12];
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        cell[2 * k] = Tb->real_of_ring[pk_get(s.posN, k) & 63];
        cell[2 * k + 1] = Tb->real_of_ring[pk_get(s.posP, k) & 63];
    }
    __builtin_amdgcn_wave_barrier(); // the zeroing of every lane of the group is issued before any cube byte
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        if (T == 1 || (2 * k) % T == sub) if (!(pk_get(s.posN, k) & PK_OFF)) b[cell[2 * k]] = (int8_t)(k + 1);
        if (T == 1 || (2 * k + 1) % T == sub) if (!(pk_get(s.posP, k) & PK_OFF)) b[cell[2 * k + 1]] = (int8_t)(-(k + 1));
    }
}

// the cube bytes only, into a board the caller has zeroed (ewn_rollout.hpp zeroes a wave's boards with 16-byte stores)
template <int S, int T>
EWN_DEV void d3_encode_cubes(const FastTab<S> *Tb, const RState<S> &s, int sub, int8_t *b)
{
    // lane `sub` of the game's T lanes places the cubes whose slot number (2k: agent cube k, 2k + 1: opponent cube k) is sub mod T;
    // for T = 2 that is one whole side per lane: six table reads and six byte stores each
    if constexpr (T == 2) {
        const u64 pos = sub ? s.posP : s.posN;
        int cell[6];
        #pragma unroll
        for (int k = 0; k < 6; k++) cell[k] = Tb->real_of_ring[pk_get(pos, k) & 63];
        #pragma unroll
        for (int k = 0; k < 6; k++) if (!(pk_get(pos, k) & PK_OFF)) b[cell[k]] = (int8_t)(sub ? -(k + 1) : (k + 1));
    } else {
        int cell[12];
        #pragma unroll
        for (int k = 0; k < 6; k++) {
            cell[2 * k] = Tb->real_of_ring[pk_get(s.posN, k) & 63];
            cell[2 * k + 1] = Tb->real_of_ring[pk_get(s.posP, k) & 63];
        }
        #pragma unroll
        for (int k = 0; k < 6; k++) {
            if (T == 1 || (2 * k) % T == sub) if (!(pk_get(s.posN, k) & PK_OFF)) b[cell[2 * k]] = (int8_t)(k + 1);
            if (T == 1 || (2 * k + 1) % T == sub) if (!(pk_get(s.posP, k) & PK_OFF)) b[cell[2 * k + 1]] = (int8_t)(-(k + 1));
        }
    }
}

template <int S>
EWN_DEV void d3_init_state(const FastTab<S> *Tb, RState<S> &s)
{
    typedef typename MaskOf<S>::type M;
    s.P = (M)Tb->init_P; s.N = (M)Tb->init_N; s.posP = Tb->init_posP; s.posN = Tb->init_posN;
}

// EinsteinWuerfeltNichtEnv.step (envs/ewn.py:436-486), minimax(depth 3, hybrid) opponent, cube_layer 3.
// T lanes per game; lanes of a group run identical code on identical data except inside d3_search.
// Between two fused MT launches: the list just produced becomes the one to consume.
static __global__ void k_mtq_flip(u32 *ctrl) { if (threadIdx.x == 0 && blockIdx.x == 0) ctrl[0] ^= 1u; }

// OPP 0: ExpectiMinimaxAgent(max_depth=3, 'hybrid') reply;  OPP 1: RandomAgent reply (classical_policies/random_policy.py:11-15)
// RNGK: the dice RNG kind as a compile-time constant, so each instantiation carries only its own generator's registers
// H2: the 'two_min_dist' table image (a side's index is the sum of its two smallest distances, ewn_fast.hpp)
template <int S, int T, int OPP, int RNGK, bool H2 = false>
__global__ __launch_bounds__(D3_BS, (OPP == 2 ? 2 : 1)) void k_step_d3(D3Cfg c, D3Buf B) // max_depth 5 / 6: held to 256 registers (two waves per SIMD)
{
    constexpr int CELLS = S * S, GPB = D3_BS / T; // games per block
    // Philox instances whose block fits keep their LDS in a STATIC array: a compile-time address folds into the offset field of
    // every ds_read, the dynamic-LDS base costs one VALU add in front of every table read (ewn_rollout.hpp has the numbers).
    // The MT kind sizes its LDS at launch (the refill role needs (W + 1) x 65 words).
    extern __shared__ __attribute__((aligned(16))) int8_t lds_dyn[];
    constexpr int LDS_ST = d3_lds_static<S, T, RNGK>();
    __shared__ __attribute__((aligned(16))) int8_t lds_st[LDS_ST ? LDS_ST : 16];
    int8_t *lds = LDS_ST ? lds_st : lds_dyn;
    // ---- MT kind with auto-reset: window refills ride along in this launch.  The step blocks of the PREVIOUS launch
    // parked (lane, slot, seed) requests in list[phase ^ 1]; the first refill_blocks blocks of THIS launch rebuild those
    // windows while the step blocks run (the 397+W-step recurrence is pure latency: hidden behind the step instead of
    // serialised after it).  Nobody reads a window before the launch after the one that rebuilt it (header fields
    // READY / X / Y, ewn_core.hpp); k_mtq_flip toggles `phase` between launches; no atomics anywhere.
    [[maybe_unused]] MtQueue Q;
    [[maybe_unused]] u32 phase = 0;
    if constexpr (RNGK == 0) {
        if (B.mtq) {
            const int step_blocks = (int)gridDim.x - c.refill_blocks;
            Q = mtq_make(B.mtq, step_blocks, 2 * GPB);
            phase = Q.ctrl[0] & 1u;
            if ((int)blockIdx.x < c.refill_blocks) {
                if (threadIdx.x < 64) {
                    // a latency-bound dependent chain that shares its SIMD with busy step waves: let it win arbitration
                    __builtin_amdgcn_s_setprio(D3_REFILL_PRIO);
                    const u32 *cnt = Q.cnt + (size_t)(phase ^ 1u) * Q.nb4;
                    // refill block j serves the regions of step blocks j, j + refill_blocks, ...
                    for (int sb = (int)blockIdx.x; sb < step_blocks; sb += c.refill_blocks) {
                        const u32 n = min(cnt[sb], (u32)Q.per_blk);
                        const uint4 *list = Q.list + ((size_t)(phase ^ 1u) * Q.nblk + sb) * Q.per_blk;
                        for (u32 e = threadIdx.x; e < n; e += 64u) {
                            const uint4 q = list[e];
                            const int lane = (int)(q.x & 0x3FFFFFFFu);
                            if (q.z == *rng_epoch_ptr(B.rng, c.N, c.W, lane)) // not invalidated by an explicit reset
                                mt_window_lds(q.y, c.W, (u32 *)lds, threadIdx.x, rng_win_ptr(B.rng, c.N, c.W, lane, q.x >> 30));
                        }
                    }
                }
                return;
            }
        }
    }
    if constexpr (RNGK == 0) { if (D3_STEP_PRIO) __builtin_amdgcn_s_setprio(D3_STEP_PRIO); }
    // the terminal-observation staging area exists only when the caller asked for terminal observations (or the block is static):
    // together with the refill requests going straight to their global list (below), the MT kind's step block then needs 37 KB, four
    // blocks fit a CU and all 512 step + 512 refill blocks of a 65 536-lane launch are resident at once -- at 44.5 KB three fitted,
    // 768 of the 1 024, and the launch ran in two rounds (26.6 -> 32.8 us per step between rounds 1 and 2, when the table image grew)
    const int nboards = (LDS_ST != 0 || B.tboard != nullptr) ? 2 : 1;
    int8_t *lds_t = lds + GPB * CELLS;
    int8_t *tb = lds + ((nboards * GPB * CELLS + 15) & ~15);
    tables_to_lds<FAST_TAB_BYTES(S)>(tb, (const int8_t *)B.tables); // LDS-DMA, waited for at the barrier
    const FastTab<S> *Tb = (const FastTab<S> *)tb;
    uint8_t *garr = (uint8_t *)(tb + FAST_TAB_BYTES(S));          // 16 bytes per game: d3_decode's scatter area
    [[maybe_unused]] u32 *qlds = (u32 *)(garr + GPB * 16);        // [0] = refill requests parked by this block so far (they go straight to its global list)
    if constexpr (RNGK == 0) { if (B.mtq && threadIdx.x == 0) qlds[0] = 0; }

    const int g0 = ((int)blockIdx.x - c.refill_blocks) * GPB, ng = min(GPB, c.N - g0);
    const int gl = threadIdx.x / T, sub = threadIdx.x % T, game = g0 + gl;
    const bool live = game < c.N, writer = sub == 0;

    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1, aflag = 0, adir = 0, tol = 0;
    double prev = 0.0;
    bool frozen = false;
    if (live) {
        hdr = *rng_hdr_ptr(B.rng, game);
        dice = B.dice[game];
        frozen = B.done[game] != 0;
        const uint16_t a2 = ((const uint16_t *)B.actions)[game];
        aflag = (int8_t)(a2 & 0xff); adir = (int8_t)(a2 >> 8);
        if (c.shaped) { tol = B.tolerance[game]; prev = B.prev_score[game]; }
    }
    // boards: request them now, put them into LDS after the RNG work below, whose ~200 issue slots (one Philox block) then
    // run while the data is on its way instead of in the agent half
    const int8_t *gsrc = B.board + (size_t)g0 * CELLS;
    const int nbytes = ng * CELLS, nq = nbytes >> 4;
    const bool split = (((uintptr_t)gsrc | (uintptr_t)nbytes) & 15) == 0 && nq <= 2 * D3_BS;
    uint4 st0 = make_uint4(0u, 0u, 0u, 0u), st1 = st0;
    if (split) {
        if ((int)threadIdx.x < nq) st0 = ((const uint4 *)gsrc)[threadIdx.x];
        if ((int)threadIdx.x + D3_BS < nq) st1 = ((const uint4 *)gsrc)[threadIdx.x + D3_BS];
    }

    int8_t *mine = lds + gl * CELLS, *mine_t = lds_t + gl * CELLS;
    double reward = 0.0;
    int term = 0, trunc = 0, info = EWN_INFO_NONE;
    RState<S> s;
    LaneRng r; r.load(RNGK, hdr, rng_win_ptr(B.rng, c.N, c.W, live ? game : 0, RNGF_CUR(hdr.w)), c.W, c.key);
    r.begin_kernel();
    r.prefetch();
    [[maybe_unused]] u32 epoch = 0;
    if constexpr (RNGK == 0) { if (live) { r.prefetch_next(B.rng, c.N, game); epoch = *rng_epoch_ptr(B.rng, c.N, c.W, game); } }
    r.begin_step();
    if constexpr (RNGK == 1) r.ps.prime();
    if (split) {
        if ((int)threadIdx.x < nq) ((uint4 *)lds)[threadIdx.x] = st0;
        if ((int)threadIdx.x + D3_BS < nq) ((uint4 *)lds)[threadIdx.x + D3_BS] = st1;
    } else block_copy_in(lds, gsrc, nbytes);
    lds_dma_wait(); // this wave's table chunks are in LDS before any other wave is let past the barrier (ewn_kernels.hip)
    __syncthreads();
    d3_decode<S, T>(live ? mine : lds, sub, garr + gl * 16, s); // every lane takes part (DPP combine); non-live lanes read game 0 of the block
    const bool active = live && !frozen;
    bool reply = false;
    if (frozen) term = 1; // no reference counterpart: a finished, un-reset game stays put
    if (active) {
        // agent half, envs/ewn.py:438-458 (the agent is the canonical BOTTOM_RIGHT side)
        const int k = pk_cube(pk_sel<S>(Tb, s.posN, dice), aflag == 1);
        const int q = (adir >= 0 && adir <= 2) ? Tb->nbn[adir][pk_get(s.posN, k)] : 255; // no cube at all: byte 6 -> 255
        if (q == 255) {
            if (c.shaped) { // envs/training_ewn.py:48-56: an illegal move costs tolerance; the game goes on until it is used up
                const int t = tol - 1;
                if (writer) B.tolerance[game] = t;
                if (t <= 0) { reward = -c.reward; term = 1; trunc = 1; info = EWN_INFO_INVALID_PLAYER; }
                else { reward = c.illegal_reward; info = EWN_INFO_TOLERANCE; }
            } else { reward = -c.reward; term = 1; trunc = 1; info = EWN_INFO_INVALID_PLAYER; }
        }
        else {
            rs_move<S, false>(s, k, q);
            if (q == Tb->ri_origin || s.P == 0) { reward = c.reward; term = 1; info = EWN_INFO_WON; }
            else { dice = r.randint(1, 7); reply = true; }
        }
    }
    // the opponent's search: run by every lane (lanes without a pending reply compute on a harmless state),
    // so the DPP exchanges inside always see their partners
    int oflag = 0, odir = 0;
    if constexpr (OPP == 0) d3_search<S, T, H2>(Tb, s, dice, sub, c.depth, oflag, odir);
    if constexpr (OPP == 2) d5_dispatch<S, (T > 2 ? 2 : T), H2>(Tb, s, dice, sub, oflag, odir); // max_depth 5 / 6: its own instances (T = 1, 2), so its registers do not weigh on the others
    if (reply) {
        // opponent half, envs/ewn.py:464-486
        const u32 e = pk_sel<S>(Tb, s.posP, dice);
        if constexpr (OPP == 1) {
            // uniform index into the opponent's legal list (reference order: larger-neighbour cube first, dirs ascending),
            // drawn from the lane's own dice stream like the reference's shared global stream
            const u32 pp = pk_pair(s.posP, e);
            const u32 okm = (u32)Tb->lgp[pp & 0xFFu] | ((u32)Tb->lgp[pp >> 8] << 3);
            const int slot = Tb->nth[okm * 8u + (u32)r.randint(0, __popc(okm))]; // 0..5 = cube slot * 3 + dir
            oflag = slot < 3 ? (int)(e >> 15) : 0;
            odir = slot < 3 ? slot : slot - 3;
        }
        const int k = pk_cube(e, oflag == 1);
        const int q = Tb->nbp[odir][pk_get(s.posP, k)];
        rs_move<S, true>(s, k, q);
        if (q == CELLS - 1 || s.N == 0) { reward = -c.reward; term = 1; info = EWN_INFO_LOST; }
        else {
            dice = r.randint(1, 7);
            if (c.shaped) { // envs/training_ewn.py:94-96: reward = evaluate() - prev_score
                const double cur = d3_shaped_score<S>(Tb, s);
                reward = cur - prev;
                if (writer) B.prev_score[game] = cur;
            }
        }
    }
    if (live && B.tboard) { if (active) d3_encode<S, T>(Tb, s, sub, mine_t); else if (writer) for (int i = 0; i < CELLS; i++) mine_t[i] = mine[i]; }
    if (live && writer && B.tdice) B.tdice[game] = (int8_t)dice;
    if (active) {
        if (term) {
            if (c.autoreset) { // reset(seed = next_seed) + setup_game (envs/ewn.py:488-494, 94-108)
                LaneRng::Pending pend; pend.n = 0; pend.slot0 = pend.slot1 = pend.seed0 = pend.seed1 = 0;
                r.next_episode(B.rng, c.N, game, c.seed_stride, c.key, (RNGK == 0 && B.mtq) ? &pend : nullptr);
                if constexpr (RNGK == 0) {
                    if (B.mtq && writer) { // park the refill requests in this block's own region of this launch's list (<= 2 per game: never full)
                        uint4 *list = Q.list + ((size_t)phase * Q.nblk + ((int)blockIdx.x - c.refill_blocks)) * Q.per_blk;
                        if (pend.n >= 1u) list[atomicAdd(&qlds[0], 1u)] = make_uint4((u32)game | (pend.slot0 << 30), pend.seed0, epoch, 0u);
                        if (pend.n >= 2u) list[atomicAdd(&qlds[0], 1u)] = make_uint4((u32)game | (pend.slot1 << 30), pend.seed1, epoch, 0u);
                    }
                }
                d3_init_state<S>(Tb, s);
                dice = r.first_dice(6);
                if (c.shaped && c.refresh && writer) B.prev_score[game] = d3_shaped_score<S>(Tb, s);
            }
            else if (writer) B.done[game] = 1;
        }
    }
    uint16_t ract_word = 0;
    if (live && writer && B.ract) {
        // RandomAgent.predict on the post-step observation (the agent is the canonical BOTTOM_RIGHT side)
        int f = 0, d = 0;
        if (!(frozen || (term && !c.autoreset))) {
            const u32 e = pk_sel<S>(Tb, s.posN, dice), pp = pk_pair(s.posN, e);
            const u32 okm = (u32)Tb->lgn[pp & 0xFFu] | ((u32)Tb->lgn[pp >> 8] << 3);
            const int n = __popc(okm);
            if (n > 0) {
                const u32 w = agent_hash(r.seed_mix(), r.draws(), (u32)(c.lane_offset + game), c.key);
                const int slot = Tb->nth[okm * 8u + __umulhi(w, (u32)n)];
                f = slot < 3 ? (int)(e >> 15) : 0;
                d = slot < 3 ? slot : slot - 3;
            }
        }
        ract_word = (uint16_t)((uint8_t)f | ((uint16_t)(uint8_t)d << 8));
    }
    // table reads above, board bytes below: the compiler cannot move an LDS read across an LDS write it cannot tell apart
    if (active) {
        d3_encode<S, T>(Tb, s, sub, mine);
        if (writer) {
            *rng_hdr_ptr(B.rng, game) = r.header();
            B.dice[game] = (int8_t)dice;
        }
    }
    if (live && writer && B.ract) ((uint16_t *)B.ract)[game] = ract_word;
    if (live && writer) {
        B.reward[game] = reward; B.terminated[game] = (uint8_t)term;
        B.truncated[game] = (uint8_t)trunc; B.info[game] = (uint8_t)info;
    }
    __syncthreads();
    block_copy_out(B.board + (size_t)g0 * CELLS, lds, ng * CELLS);
    if (B.tboard) block_copy_out(B.tboard + (size_t)g0 * CELLS, lds_t, ng * CELLS);
    if constexpr (RNGK == 0) {
        if (B.mtq) { // how many requests this block parked: a plain store (the __syncthreads above ordered the LDS counter)
            const int sb = (int)blockIdx.x - c.refill_blocks;
            if (threadIdx.x == 0) Q.cnt[(size_t)phase * Q.nb4 + sb] = min(qlds[0], (u32)Q.per_blk);
        }
    }
}
