// ewn_core.hpp -- device-side game core for the gfx950 EWN engine.
//
// One game lives in a handful of registers: two 64-bit occupancy masks (bit c =
// row*S+col), the cube positions packed 6 bits each, and one alive mask per side.
// Nothing here follows the reference's numpy data structures (a signed int16 board
// plus a masked structured cube_pos array, envs/ewn.py:49-58); the rules it has to
// reproduce are cited per function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef uint32_t u32;

#define EWN_DEV __device__ __forceinline__

// Board geometry, built on the host (ewn_capi.hip: make_geom) and passed by value
// as a kernel argument, so every field is wave-uniform (SGPRs).
struct Geom {
    int S, L, CN, cells;
    u64 not_lastcol, not_lastrow, not_firstcol, not_firstrow, corner_br; // boards up to 8x8 only (one 64-bit mask)
    u64 sq[8];          // sq[t] = cells with row >= t and col >= t  (distance to the bottom-right corner <= S-1-t)
    int8_t init[128];   // initial board, envs/ewn.py:94-107
    // the same position already decoded (GState layout), so reset() copies registers instead of scanning cells
    u64 init_occP, init_occN, init_posP[3], init_posN[3];
    u32 init_alive;
    u32 div_magic;      // (c * div_magic) >> 16 == c / S for every cell c (boards from 9x9: rows and columns by arithmetic)
};

// SIDE 0 = TOP_LEFT (positive numbers), SIDE 1 = BOTTOM_RIGHT (negative numbers).
template <int NW>
struct GState {
    u64 occP, occN;
    u64 posP[NW], posN[NW]; // cube k (0-based) at bits [6*(k%10), +6) of word k/10
    u32 aliveP, aliveN;
};

// NW == 3: boards from 9x9 to 11x11.  81..121 cells fit neither a 64-bit occupancy mask nor 6-bit positions, and nobody
// benchmarks these sizes (the reference merely allows them: assert cube_layer < board_size - 1, envs/ewn.py:47), so the state is
// mask-free -- 7-bit positions, nine per word, plus the alive masks -- and "what stands on cell q" is a scan over <= 30 cubes.
// Every rule function below has an `if constexpr (NW == 3)` branch; the kernels are the same templates.
template <>
struct GState<3> {
    u64 posP[3], posN[3];   // cube k (0-based) at bits [7*(k%9), +7) of word k/9
    u32 aliveP, aliveN;
};

template <int NW>
EWN_DEV int pos_get(const u64 (&w)[NW], int k)
{
    if constexpr (NW == 1) return (int)((w[0] >> (6 * k)) & 63ull);
    else if constexpr (NW == 2) return k < 10 ? (int)((w[0] >> (6 * k)) & 63ull) : (int)((w[1] >> (6 * (k - 10))) & 63ull);
    else { const int wi = k >= 9 ? 1 : 0; return (int)((w[wi] >> (7 * (k - 9 * wi))) & 127ull); } // <= 15 cubes: words 0 and 1
}

template <int NW>
EWN_DEV void pos_set(u64 (&w)[NW], int k, int c)
{
    if constexpr (NW == 1) {
        w[0] = (w[0] & ~(63ull << (6 * k))) | ((u64)c << (6 * k));
    } else if constexpr (NW == 2) {
        if (k < 10) w[0] = (w[0] & ~(63ull << (6 * k))) | ((u64)c << (6 * k));
        else w[1] = (w[1] & ~(63ull << (6 * (k - 10)))) | ((u64)c << (6 * (k - 10)));
    } else {
        if (k < 9) w[0] = (w[0] & ~(127ull << (7 * k))) | ((u64)c << (7 * k));
        else w[1] = (w[1] & ~(127ull << (7 * (k - 9)))) | ((u64)c << (7 * (k - 9)));
    }
}

template <int SIDE, int NW> EWN_DEV u64 &occ_of(GState<NW> &s) { if constexpr (SIDE == 0) return s.occP; else return s.occN; }
template <int SIDE, int NW> EWN_DEV u32 &alive_of(GState<NW> &s) { if constexpr (SIDE == 0) return s.aliveP; else return s.aliveN; }
template <int SIDE, int NW> EWN_DEV u32 alive_of(const GState<NW> &s) { if constexpr (SIDE == 0) return s.aliveP; else return s.aliveN; }
template <int SIDE, int NW> EWN_DEV int pos_of(const GState<NW> &s, int k) { if constexpr (SIDE == 0) return pos_get<NW>(s.posP, k); else return pos_get<NW>(s.posN, k); }
template <int SIDE, int NW> EWN_DEV void set_pos_of(GState<NW> &s, int k, int c) { if constexpr (SIDE == 0) pos_set<NW>(s.posP, k, c); else pos_set<NW>(s.posN, k, c); }

// board bytes -> registers (the 25/49-cell scan of restore_env_with_obs,
// classical_policies/minimax.py:75-87, done once per lane per call)
template <int NW, class BytePtr>
EWN_DEV void decode_board(const Geom &g, BytePtr b, GState<NW> &s)
{
    s.aliveP = s.aliveN = 0;
    if constexpr (NW != 3) s.occP = s.occN = 0;
    for (int w = 0; w < NW; w++) { s.posP[w] = 0; s.posN[w] = 0; }
    for (int c = 0; c < g.cells; c++) {
        int v = (int)(int8_t)b[c];
        if (v > 0 && v <= g.CN) { if constexpr (NW != 3) s.occP |= 1ull << c; s.aliveP |= 1u << (v - 1); pos_set<NW>(s.posP, v - 1, c); }
        else if (v < 0 && -v <= g.CN) { if constexpr (NW != 3) s.occN |= 1ull << c; s.aliveN |= 1u << (-v - 1); pos_set<NW>(s.posN, -v - 1, c); }
    }
}

template <int NW, class BytePtr>
EWN_DEV void encode_board(const Geom &g, const GState<NW> &s, BytePtr b)
{
    for (int c = 0; c < g.cells; c++) b[c] = 0;
    for (int k = 0; k < g.CN; k++) {
        if ((s.aliveP >> k) & 1u) b[pos_get<NW>(s.posP, k)] = (int8_t)(k + 1);
        if ((s.aliveN >> k) & 1u) b[pos_get<NW>(s.posN, k)] = (int8_t)(-(k + 1));
    }
}

template <int NW>
EWN_DEV void init_state(const Geom &g, GState<NW> &s)
{
    if constexpr (NW != 3) { s.occP = g.init_occP; s.occN = g.init_occN; }
    s.aliveP = s.aliveN = g.init_alive;
    for (int w = 0; w < NW; w++) { s.posP[w] = g.init_posP[w]; s.posN[w] = g.init_posN[w]; }
}

// opponent_action's np.rot90(-board, 2) (envs/ewn.py:294): swap sides, mirror every cell index.
template <int NW>
EWN_DEV GState<NW> canonicalize(const Geom &g, const GState<NW> &s)
{
    GState<NW> c;
    c.aliveP = s.aliveN; c.aliveN = s.aliveP;
    // fieldwise (cells-1) - pos: every field holds a value <= cells-1, so no borrow crosses fields
    u64 rep = 0;
    if constexpr (NW == 3) {
        for (int k = 0; k < 9; k++) rep |= (u64)(g.cells - 1) << (7 * k);
    } else {
        const int sh = 64 - g.cells;
        c.occP = __brevll(s.occN) >> sh;
        c.occN = __brevll(s.occP) >> sh;
        for (int k = 0; k < 10; k++) rep |= (u64)(g.cells - 1) << (6 * k);
    }
    for (int w = 0; w < NW; w++) { c.posP[w] = rep - s.posN[w]; c.posN[w] = rep - s.posP[w]; }
    return c;
}

// check_win, envs/ewn.py:131-142
template <int NW>
EWN_DEV bool is_win(const Geom &g, const GState<NW> &s)
{
    if constexpr (NW == 3) {
        bool home = false;
        for (int k = 0; k < g.CN; k++) {
            home |= ((s.aliveP >> k) & 1u) && pos_get<3>(s.posP, k) == g.cells - 1;
            home |= ((s.aliveN >> k) & 1u) && pos_get<3>(s.posN, k) == 0;
        }
        return home || s.aliveP == 0 || s.aliveN == 0;
    } else return (s.occN & 1ull) || (s.occP & g.corner_br) || s.occP == 0 || s.occN == 0;
}

// mcts.py:39-41 / minimax_ewn.py:233-235: TOP_LEFT has won a finished playout (its cube on the far corner, or no opponent left)
template <int NW>
EWN_DEV bool top_left_won(const Geom &g, const GState<NW> &s)
{
    if constexpr (NW == 3) {
        bool home = false;
        for (int k = 0; k < g.CN; k++) home |= ((s.aliveP >> k) & 1u) && pos_get<3>(s.posP, k) == g.cells - 1;
        return home || s.aliveN == 0;
    } else return (s.occP & g.corner_br) || s.occN == 0;
}

// Dice -> candidate cubes (find_near_cube, envs/ewn.py:144-176; both players index by
// |cube number|): the dice cube if alive, else the nearest alive larger / smaller one.
struct CubeSel { bool exact, has_up, has_down; int k_exact, k_up, k_down; };

EWN_DEV CubeSel select_cubes(u32 alive, int dice)
{
    CubeSel r;
    const u32 bit = 1u << (dice - 1);
    r.exact = (alive & bit) != 0;
    r.k_exact = dice - 1;
    const u32 up = alive & ~((bit << 1) - 1u);
    const u32 down = alive & (bit - 1u);
    r.has_up = up != 0; r.has_down = down != 0;
    r.k_up = __ffs((int)up) - 1;
    r.k_down = 31 - __clz((int)down);
    return r;
}

// find_cube_to_move, envs/ewn.py:178-215 (returns -1 where the reference's assert fires)
EWN_DEV int cube_to_move(const CubeSel &c, bool larger)
{
    if (c.exact) return c.k_exact;
    if (larger) return c.has_up ? c.k_up : (c.has_down ? c.k_down : -1);
    return c.has_down ? c.k_down : (c.has_up ? c.k_up : -1);
}

// is_within_board(update_position(...)), envs/ewn.py:301-323
template <int SIDE>
EWN_DEV bool dir_ok(const Geom &g, int c, int dir)
{
    if (g.S > 8) { // no 64-bit mask of the board: row and column by arithmetic
        const int row = (int)(((u32)c * g.div_magic) >> 16), col = c - row * g.S;
        const bool col_ok = SIDE == 0 ? col < g.S - 1 : col > 0, row_ok = SIDE == 0 ? row < g.S - 1 : row > 0;
        return dir == 0 ? col_ok : (dir == 1 ? row_ok : (col_ok && row_ok));
    }
    const u64 b = 1ull << c;
    const u64 colm = SIDE == 0 ? g.not_lastcol : g.not_firstcol;
    const u64 rowm = SIDE == 0 ? g.not_lastrow : g.not_firstrow;
    const bool col_ok = (b & colm) != 0, row_ok = (b & rowm) != 0;
    return dir == 0 ? col_ok : (dir == 1 ? row_ok : (col_ok && row_ok));
}

template <int SIDE>
EWN_DEV int dest_cell(const Geom &g, int c, int dir)
{
    const int d = dir == 0 ? 1 : (dir == 1 ? g.S : g.S + 1);
    return SIDE == 0 ? c + d : c - d;
}

template <int SIDE, int NW>
EWN_DEV void kill_at(const Geom &g, GState<NW> &s, int q)
{
    u32 a = alive_of<SIDE>(s);
    for (int k = 0; k < g.CN; k++)
        if (((a >> k) & 1u) && pos_of<SIDE>(s, k) == q) a &= ~(1u << k);
    alive_of<SIDE>(s) = a;
    if constexpr (NW != 3) occ_of<SIDE>(s) &= ~(1ull << q);
}

// execute_move / make_simulated_action for a move already known to stay on the board
// (envs/ewn.py:252-261, 393-404): vacate, capture whatever is there (own cube included), place.
template <int SIDE, int NW>
EWN_DEV void apply_move(const Geom &g, GState<NW> &s, int k, int dir)
{
    const int p = pos_of<SIDE>(s, k);
    const int q = dest_cell<SIDE>(g, p, dir);
    if constexpr (NW == 3) {
        kill_at<1 - SIDE>(g, s, q);   // at most one cube stands on q, of either side; the scans simply find nothing otherwise
        kill_at<SIDE>(g, s, q);
    } else {
        const u64 bq = 1ull << q;
        if (occ_of<1 - SIDE>(s) & bq) kill_at<1 - SIDE>(g, s, q);
        else if (occ_of<SIDE>(s) & bq) kill_at<SIDE>(g, s, q);
        occ_of<SIDE>(s) = (occ_of<SIDE>(s) & ~(1ull << p)) | bq;
    }
    set_pos_of<SIDE>(s, k, q);
}

// get_legal_actions, envs/ewn.py:338-375.  Visits the legal actions of SIDE in the
// reference's list order; f(flag, k, dir) returns false to stop early.
template <int SIDE, int NW, class F>
EWN_DEV int for_each_legal(const Geom &g, const GState<NW> &s, int dice, F &&f)
{
    const CubeSel cs = select_cubes(alive_of<SIDE>(s), dice);
    int n = 0;
    bool go = true;
    #pragma unroll
    for (int slot = 0; slot < 2; slot++) {
        bool have; int k, flag;
        if (slot == 0) { have = cs.exact || cs.has_up; k = cs.exact ? cs.k_exact : cs.k_up; flag = cs.exact ? 0 : 1; }
        else { have = !cs.exact && cs.has_down; k = cs.k_down; flag = 0; }
        if (have && go) {
            const int p = pos_of<SIDE>(s, k);
            for (int dir = 0; dir < 3 && go; dir++)
                if (dir_ok<SIDE>(g, p, dir)) { n++; go = f(flag, k, dir); }
        }
    }
    return n;
}

// ---------------------------------------------------------------- heuristics

// min over a side's cubes of max(S-1-row, S-1-col): BOTH sides are measured to the
// bottom-right corner (envs/minimax_ewn.py:67-76).
EWN_DEV int min_dist_br(const Geom &g, u64 m)
{
    int t = g.S - 1;
    while (t > 0 && (m & g.sq[t]) == 0) t--;
    return g.S - 1 - t;
}

// MinimaxEnv.evaluate, envs/minimax_ewn.py:29-213 (agent_player is TOP_LEFT in every
// policy's private env).  fp64 in the reference's operation order; compile with
// -ffp-contract=off so the multiply and the subtract stay two roundings.
template <int NW>
EWN_DEV double evaluate(const Geom &g, const GState<NW> &s, int heur)
{
    if constexpr (NW == 3) {
        // boards from 9x9: the same arithmetic from the cube positions (distance of a cube = max(S-1-row, S-1-col))
        if (top_left_won<3>(g, s)) return 10.0;                     // :42-44
        bool n_home = false;
        for (int k = 0; k < g.CN; k++) n_home |= ((s.aliveN >> k) & 1u) && pos_get<3>(s.posN, k) == 0;
        if (n_home || s.aliveP == 0) return -10.0;                  // :45-47
        const int np = __popc(s.aliveP), nn = __popc(s.aliveN);
        if (heur == 3) return (double)(-(np + nn));
        int p1 = 1 << 20, p2 = 1 << 20, n1 = 1 << 20, n2 = 1 << 20; // the two smallest distances of either side
        for (int k = 0; k < g.CN; k++) {
            #pragma unroll
            for (int side = 0; side < 2; side++) {
                if (!(((side ? s.aliveN : s.aliveP) >> k) & 1u)) continue;
                const int c = pos_get<3>(side ? s.posN : s.posP, k);
                const int row = (int)(((u32)c * g.div_magic) >> 16), col = c - row * g.S;
                const int d = max(g.S - 1 - row, g.S - 1 - col);
                int &a = side ? n1 : p1, &b = side ? n2 : p2;
                if (d < a) { b = a; a = d; } else if (d < b) b = d;
            }
        }
        if (heur == 2) return (double)((n1 + (nn >= 2 ? n2 : 0)) - (p1 + (np >= 2 ? p2 : 0)));   // two_min_dist :157-174
        if (heur == 1) return (double)((g.S - p1) - (g.S - n1));    // min_dist :126-127
        const double a = (double)(g.S - p1) * (1.0 / (double)np);   // hybrid :79-80
        const double b = (double)(g.S - n1) * (1.0 / (double)nn);   // :81-82
        return (0.0 + a) - b;
    } else {

    if ((s.occP & g.corner_br) || s.occN == 0) return 10.0;     // :42-44
    if ((s.occN & 1ull) || s.occP == 0) return -10.0;           // :45-47
    const int np = __popcll(s.occP), nn = __popcll(s.occN);
    if (heur == 3) return (double)(-(np + nn));                 // attk :212
    if (heur == 2) {                                            // two_min_dist :157-174
        int sp = 0, sn = 0, tp = 0, tn = 0;
        for (int t = g.S - 1; t >= 0; t--) {
            const u64 ring = g.sq[t] & ~(t + 1 < g.S ? g.sq[t + 1] : 0ull);
            int cp = __popcll(s.occP & ring), cn = __popcll(s.occN & ring);
            const int d = g.S - 1 - t;
            while (cp > 0 && tp < 2) { sp += d; tp++; cp--; }
            while (cn > 0 && tn < 2) { sn += d; tn++; cn--; }
        }
        return (double)(sn - sp);
    }
    const int mdp = min_dist_br(g, s.occP), mdn = min_dist_br(g, s.occN);
    if (heur == 1) return (double)((g.S - mdp) - (g.S - mdn));  // min_dist :126-127
    const double a = (double)(g.S - mdp) * (1.0 / (double)np);  // hybrid :79-80
    const double b = (double)(g.S - mdn) * (1.0 / (double)nn);  // :81-82
    return (0.0 + a) - b;
    }
}

// ---------------------------------------------------------------- expectiminimax

// classical_policies/minimax.py:19-73, the recursion unrolled at compile time.
// KIND 0 = MAX (TOP_LEFT moves), 1 = MIN (BOTTOM_RIGHT moves), 2 = CHANCE below MAX,
// 3 = CHANCE below MIN.  Alpha/beta are passed by value THROUGH chance nodes exactly
// as the reference does (unsound but behaviour-defining); depth also
// decrements at chance nodes; the dice loop is the hard-coded 1..6 (App. D5).
// The value of a leaf is `leaf(g, s)`: EvalLeaf = MinimaxEnv.evaluate(heuristic) for the four board heuristics; the
// 'sim_winrate' heuristic (random playouts, envs/minimax_ewn.py:215-238) brings its own functor (ewn_kernels.hip: SimLeaf).
template <int NW>
struct EvalLeaf {
    static constexpr bool outline_deep = false;
    int heur;
    EWN_DEV double operator()(const Geom &g, const GState<NW> &s) { return evaluate<NW>(g, s, heur); }
};

template <int NW, int DEPTH, int KIND, bool ROOT, class Leaf>
__device__ double search(const Geom &g, const GState<NW> &s, int dice, double alpha, double beta, Leaf &leaf, int &bflag, int &bdir);

// A leaf type with a large body (SimLeaf: a playout loop) asks for the depth-4 subtree of a deeper search to be a real function:
// for_each_legal unrolls its two cube slots, so every level inlined doubles the code below it, and at max_depth 5 / 6 the optimiser
// gives the unrolling up half way (with a warning per loop).  One call per subtree costs nothing next to 1 296 leaves x 100 playouts.
template <int NW, int DEPTH, int KIND, class Leaf>
__device__ __attribute__((noinline)) double search_outlined(const Geom &g, const GState<NW> &s, int dice, double alpha, double beta, Leaf &leaf)
{
    int f = 0, d = 0;
    return search<NW, DEPTH, KIND, false>(g, s, dice, alpha, beta, leaf, f, d);
}

template <int NW, int DEPTH, int KIND, bool ROOT, class Leaf>
__device__ double search(const Geom &g, const GState<NW> &s, int dice, double alpha, double beta, Leaf &leaf, int &bflag, int &bdir)
{
    if constexpr (DEPTH == 0) {
        return leaf(g, s);
    } else {
        if (is_win<NW>(g, s)) return leaf(g, s);
        if constexpr (KIND >= 2) {
            double expected = 0.0;
            for (int d = 1; d <= 6; d++) {
                double v;
                if constexpr (Leaf::outline_deep && DEPTH - 1 == 4) v = search_outlined<NW, DEPTH - 1, (KIND == 2 ? 1 : 0)>(g, s, d, alpha, beta, leaf);
                else v = search<NW, DEPTH - 1, (KIND == 2 ? 1 : 0), false>(g, s, d, alpha, beta, leaf, bflag, bdir);
                expected = expected + v / 6.0;
            }
            return expected;
        } else {
            constexpr int SIDE = KIND;
            double best = SIDE == 0 ? -__builtin_inf() : __builtin_inf();
            for_each_legal<SIDE, NW>(g, s, dice, [&](int flag, int k, int dir) -> bool {
                GState<NW> c = s;
                apply_move<SIDE, NW>(g, c, k, dir);
                double v;
                if constexpr (Leaf::outline_deep && DEPTH - 1 == 4) v = search_outlined<NW, DEPTH - 1, (SIDE == 0 ? 2 : 3)>(g, c, dice, alpha, beta, leaf);
                else v = search<NW, DEPTH - 1, (SIDE == 0 ? 2 : 3), false>(g, c, dice, alpha, beta, leaf, bflag, bdir);
                if constexpr (SIDE == 0) {
                    if (v > best) { best = v; if constexpr (ROOT) { bflag = flag; bdir = dir; } }
                    if (best > alpha) alpha = best;
                } else {
                    if (v < best) best = v;
                    if (best < beta) beta = best;
                }
                return !(beta <= alpha);
            });
            return best;
        }
    }
}

// ---------------------------------------------------------------- RNG

EWN_DEV void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 (&out)[4])
{
    #pragma unroll
    for (int r = 0; r < 10; r++) {
        const u32 h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const u32 h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const u32 n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Sequential u32 stream: word n = philox(ctr={n>>2, c1, c2, c3}, key)[n&3].
// The cached block is four scalars, not an array: an array indexed by (n & 3) ends up in scratch memory.
struct PhiloxStream {
    u32 c1, c2, c3, k0, k1, n;
    u32 b0, b1, b2, b3;
    u32 have; // block index + 1 currently cached (0 = none)
    EWN_DEV void init(u32 c1_, u32 c2_, u32 c3_, u64 key, u32 n_) { c1 = c1_; c2 = c2_; c3 = c3_; k0 = (u32)key; k1 = (u32)(key >> 32); n = n_; have = 0; b0 = b1 = b2 = b3 = 0; }
    // compute the block the next draw comes from now (e.g. while a load is in flight); next() will find it cached
    EWN_DEV void prime()
    {
        const u32 b = n >> 2;
        if (have != b + 1) { u32 o[4]; philox4x32_10(b, c1, c2, c3, k0, k1, o); b0 = o[0]; b1 = o[1]; b2 = o[2]; b3 = o[3]; have = b + 1; }
    }
    EWN_DEV u32 next()
    {
        const u32 b = n >> 2;
        if (have != b + 1) { u32 o[4]; philox4x32_10(b, c1, c2, c3, k0, k1, o); b0 = o[0]; b1 = o[1]; b2 = o[2]; b3 = o[3]; have = b + 1; }
        const u32 i = n;
        n++;
        const u32 lo = (i & 1u) ? b1 : b0, hi = (i & 1u) ? b3 : b2;
        return (i & 2u) ? hi : lo;
    }
};

EWN_DEV u32 mt_temper(u32 y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

EWN_DEV u32 mt_twist(u32 a, u32 b)
{
    const u32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// np.random.seed(seed) followed by the first W outputs of the legacy MT19937 stream
// (numpy mt19937_seed / mt19937_gen).  Output n (< 227) depends only on
// the seeded words s[n], s[n+1], s[n+397], so one pass of the seeding recurrence
// (397+W steps) yields the window without ever materialising the 624-word state.
// `win` doubles as the temporary for s[0..W-1].
EWN_DEV void mt_fill_window(u32 seed, int W, u32 *win)
{
    u32 s = seed;
    for (int i = 0; i < W; i++) { win[i] = s; s = 1812433253u * (s ^ (s >> 30)) + (u32)i + 1u; }
    const u32 sW = s; // s[W]
    for (int i = W; i < 397; i++) s = 1812433253u * (s ^ (s >> 30)) + (u32)i + 1u;
    // s == s[397]
    for (int n = 0; n < W; n++) {
        const u32 a = win[n], b = (n + 1 < W) ? win[n + 1] : sW;
        win[n] = mt_temper(s ^ mt_twist(a, b));
        s = 1812433253u * (s ^ (s >> 30)) + (u32)(397 + n) + 1u;
    }
}

// Output n of the same stream for W <= n < 454, memory-free (rare path: an episode
// that consumes more draws than the window holds).  For 227 <= n < 454 the word
// x[n+397] is itself a first-generation output of the recurrence.
EWN_DEV u32 mt_output_closed(u32 seed, u32 n)
{
    const u32 j = n >= 227u ? n - 227u : 0u;
    u32 s = seed, sn = 0, sn1 = 0, sj = 0, sj1 = 0, sj397 = 0, sn397 = 0;
    for (u32 i = 0; i < 624u; i++) {
        if (i == n) sn = s;
        if (i == n + 1u) sn1 = s;
        if (i == n + 397u) sn397 = s;
        if (i == j) sj = s;
        if (i == j + 1u) sj1 = s;
        if (i == j + 397u) sj397 = s;
        s = 1812433253u * (s ^ (s >> 30)) + i + 1u;
    }
    const u32 x397 = n >= 227u ? (sj397 ^ mt_twist(sj, sj1)) : sn397;
    return mt_temper(x397 ^ mt_twist(sn, sn1));
}

// One MT19937 window: np.random.seed(seed)'s first W outputs into dst.  The seeding recurrence (397+W dependent
// multiply-adds) runs in registers with its W+1 saved words in LDS ([word][thread], row stride 65: conflict-free).
EWN_DEV void mt_window_lds(u32 seed, u32 W, u32 *sm, int t, u32 *dst)
{
    u32 s = seed;
    for (u32 i = 0; i <= W; i++) { sm[i * 65 + t] = s; s = 1812433253u * (s ^ (s >> 30)) + i + 1u; }  // s[0..W]
    for (u32 i = W + 1; i < 397; i++) s = 1812433253u * (s ^ (s >> 30)) + i + 1u;                       // -> s[397]
    u32 a = sm[t];
    #pragma unroll 8
    for (u32 n = 0; n < W; n++) {   // output n = temper(s[397+n] ^ twist(s[n], s[n+1]))
        const u32 b = sm[(n + 1) * 65 + t];
        dst[n] = mt_temper(s ^ mt_twist(a, b));
        s = 1812433253u * (s ^ (s >> 30)) + (397u + n) + 1u;
        a = b;
    }
}

EWN_DEV u32 fmix32(u32 h) // MurmurHash3 finaliser
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

#define EWN_RNG_HDR 4 // header words per lane: seed, draw index, next_seed, flags
// flags word (MT kind): episode e of a lane lives in window slot e % 3; the two other slots hold, or are about to hold,
// the windows of episodes e+1 and e+2, so an auto-reset is a rotation, not a 500-step recurrence.
#define RNGF_OVERFLOW 1u                 // MT draw index >= 454 in one episode: unsupported (never observed)
#define RNGF_CUR(f) (((f) >> 4) & 3u)    // slot of the current episode
#define RNGF_READY(f) (((f) >> 8) & 3u)  // how many of the next episodes' windows are complete (0..2)
#define RNGF_X(f) (((f) >> 10) & 3u)     // slots freed during the previous step kernel: their refill is running now
#define RNGF_Y(f) (((f) >> 12) & 3u)     // slots freed two kernels ago: refilled during the previous kernel, usable now
EWN_DEV u32 rngf_make(u32 ovf, u32 cur, u32 ready, u32 x, u32 y) { return (ovf & 1u) | (cur << 4) | (ready << 8) | (x << 10) | (y << 12); }

// ewn_state.rng layout: N headers (uint4 each, one coalesced 16-byte access per lane); for the MT kind then N x 3
// windows of W tempered MT19937 outputs and N reset-epoch words (a lane's explicit reset invalidates queued refills).
EWN_DEV uint4 *rng_hdr_ptr(u32 *rng, int lane) { return (uint4 *)rng + lane; }
EWN_DEV u32 *rng_win_ptr(u32 *rng, int N, u32 W, int lane, u32 slot)
{
    return rng + (size_t)N * EWN_RNG_HDR + ((size_t)lane * 3 + slot) * W;
}
EWN_DEV u32 *rng_epoch_ptr(u32 *rng, int N, u32 W, int lane) { return rng + (size_t)N * EWN_RNG_HDR + (size_t)N * 3 * W + lane; }

// Window refills handed from one step launch to the next (k_step_d3): step block b of a launch parks its requests
// {lane | slot << 30, seed, epoch, 0} in ITS OWN region of list[phase] and stores how many in cnt[phase][b] -- plain
// stores, no atomics; the refill blocks of the next launch read list[phase ^ 1]; k_mtq_flip toggles phase in between.
struct MtQueue { u32 *ctrl; u32 *cnt; uint4 *list; int nblk, nb4, per_blk; };
EWN_DEV MtQueue mtq_make(void *scratch, int nblk, int per_blk)
{
    MtQueue q;
    q.nblk = nblk; q.nb4 = (nblk + 3) & ~3; q.per_blk = per_blk;
    q.ctrl = (u32 *)scratch;                   // [0] = phase
    q.cnt = q.ctrl + 4;                        // [2][nb4]
    q.list = (uint4 *)(q.cnt + 2 * q.nb4);     // [2][nblk][per_blk]
    return q;
}

// The fused stand-in agent's randomness (ewn_step_out.random_action): one 32-bit hash per (episode, draws so far, lane)
EWN_DEV u32 agent_hash(u32 seed, u32 draws, u32 lane_global, u64 key)
{
    return fmix32(seed ^ fmix32(draws * 0x9E3779B1u + lane_global) ^ fmix32((u32)key ^ 0x41474E54u) ^ ((u32)(key >> 32) * 0x85ebca6bu));
}

// k-th (0-based) set bit of a 6-bit legal mask (bits 0-2: first cube's dirs, 3-5: second cube's) -> slot index
EWN_DEV int nth_set_bit(u32 m, int k)
{
    for (int i = 0; i < k; i++) m &= m - 1;
    return __ffs((int)m) - 1;
}

// One lane's dice stream for the duration of a kernel.
//  kind 0 (MT19937): np.random.seed / legacy randint, bit-exact with the reference.
//  kind 1 (Philox):  word n of an episode = philox4x32-10(ctr={n>>2, seed, wraps of the seed so far, 'ENV1'}, key)[n&3].  A step starts on a
//                    block boundary and needs <= 3 words (opponent dice, random opponent's choice, next dice), so ONE
//                    Philox block serves a step; bounded draws use Lemire's multiply-shift (exact: the rare rejection
//                    re-draws); an episode's first dice is a hash of its seed, so reset() needs no second block.
struct LaneRng {
    int kind;         // 0 MT19937 window, 1 Philox
    u32 seed, n, next_seed, flags;
    u32 W;
    const u32 *win;
    PhiloxStream ps;
    // MT kind: the next 8 window words, fetched in one round trip when the lane is loaded, so the masked-rejection
    // loop of randint does not serialise a global load per iteration (scalars, not an array: no scratch)
    u32 pre_base, pre_lim, p0, p1, p2, p3, p4, p5, p6, p7;
    u32 q0, q1, q2, q3; // first words of the NEXT episode's window, fetched with the rest (prefetch_next)
    bool q_valid;
    EWN_DEV void prefetch()
    {
        pre_base = n; pre_lim = 8u;
        if (kind == 0 && n + 8u <= W) { p0 = win[n]; p1 = win[n + 1]; p2 = win[n + 2]; p3 = win[n + 3]; p4 = win[n + 4]; p5 = win[n + 5]; p6 = win[n + 6]; p7 = win[n + 7]; }
        else pre_base = 0xFFFFFFF0u; // nothing cached
    }
    EWN_DEV void load(int kind_, uint4 h, const u32 *win_, u32 W_, u64 key)
    {
        kind = kind_; seed = h.x; n = h.y; next_seed = h.z; flags = h.w; W = W_; win = win_;
        p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = 0; pre_base = 0xFFFFFFF0u; pre_lim = 8u; q0 = q1 = q2 = q3 = 0; q_valid = false;
        if (kind == 1) ps.init(seed, flags, 0x454E5631u, key, n); // Philox kind: the header's 4th word is the episode index's HIGH word
    }
    // The episode seed advances by seed_stride per auto-reset and wraps after 2^32 / stride episodes of a lane (np.random.seed is
    // 32-bit: inherent for the MT kind).  The Philox kind counts the wraps in the header's 4th word and feeds it to the counter
    // (c2) and to every hash keyed by the episode, so a lane's dice streams do not repeat before 2^64 / stride episodes.
    EWN_DEV u32 seed_mix() const { return kind == 1 ? seed ^ (flags * 0x9E3779B9u) : seed; }
    EWN_DEV uint4 header() const { return make_uint4(seed, kind == 1 ? ps.n : n, next_seed, flags); }
    EWN_DEV u32 draws() const { return kind == 1 ? ps.n : n; }
    EWN_DEV u32 next()
    {
        if (kind == 1) return ps.next();
        u32 v;
        const u32 j = n - pre_base;
        if (j < pre_lim) {
            const u32 a = (j & 1u) ? p1 : p0, b = (j & 1u) ? p3 : p2, c = (j & 1u) ? p5 : p4, d = (j & 1u) ? p7 : p6;
            const u32 ab = (j & 2u) ? b : a, cd = (j & 2u) ? d : c;
            v = (j & 4u) ? cd : ab;
        }
        else if (n < W) v = win[n];
        else if (n < 454u) v = mt_output_closed(seed, n);
        else { v = 0; flags |= RNGF_OVERFLOW; }
        n++;
        return v;
    }
    // np.random.randint(lo, hi) of the legacy RandomState: masked rejection on 32-bit
    // draws; a one-element range consumes no draw.
    // Once per step kernel: what was freed two kernels ago has been refilled by now.
    EWN_DEV void begin_kernel()
    {
        if (kind == 0) flags = rngf_make(flags, RNGF_CUR(flags), min(2u, RNGF_READY(flags) + RNGF_Y(flags)), 0u, RNGF_X(flags));
    }
    // speculative: the head of the next episode's window, so an auto-reset does not wait for a dependent load
    EWN_DEV void prefetch_next(u32 *rng, int N, int lane)
    {
        if (kind == 0 && RNGF_READY(flags) >= 1u) {
            const u32 *w = rng_win_ptr(rng, N, W, lane, (RNGF_CUR(flags) + 1u) % 3u);
            q0 = w[0]; q1 = w[1]; q2 = w[2]; q3 = w[3]; q_valid = true;
        }
    }
    // Start the next episode (seed = next_seed).  MT kind: rotate to the next window slot and ask for the freed slot to
    // be refilled with the window of episode e+3 -- through `pend` when the caller queues refills for the next kernel
    // (k_step_d3), else by leaving X > 0 for k_mt_refill.  If no future window is complete (cannot happen while refills
    // keep up; kept correct anyway) the current slot is rebuilt in place and both other slots are re-requested.
    struct Pending { u32 n, slot0, seed0, slot1, seed1; }; // scalars, not arrays: a runtime-indexed array would live in scratch
    EWN_DEV void next_episode(u32 *rng, int N, int lane, u32 stride, u64 key, Pending *pend)
    {
        const u32 s2 = next_seed;
        u32 f = flags;
        if (pend) { pend->n = 0; pend->slot0 = pend->slot1 = pend->seed0 = pend->seed1 = 0; }
        if (kind == 0) {
            const u32 cur = RNGF_CUR(f), ready = RNGF_READY(f);
            if (ready >= 1u) {
                f = rngf_make(0u, (cur + 1u) % 3u, ready - 1u, RNGF_X(f) + 1u, RNGF_Y(f));
                if (pend) { pend->n = 1; pend->slot0 = cur; pend->seed0 = s2 + 2u * stride; }
                const u32 a0 = q0, a1 = q1, a2 = q2, a3 = q3;
                const bool spec = q_valid;
                load(kind, make_uint4(s2, 0u, s2 + stride, f), rng_win_ptr(rng, N, W, lane, RNGF_CUR(f)), W, key);
                if (!spec) { prefetch(); return; }
                // the first dice of the new episode comes out of the speculatively fetched words; anything beyond them
                // (three masked rejections in a row) falls back to the window in memory
                pre_base = 0u; p0 = a0; p1 = a1; p2 = a2; p3 = a3; p4 = p5 = p6 = p7 = 0; pre_lim = 4u;
                return;
            } else {
                mt_fill_window(s2, (int)W, rng_win_ptr(rng, N, W, lane, cur));
                f = rngf_make(0u, cur, 0u, 2u, 0u);
                if (pend) { pend->n = 2; pend->slot0 = (cur + 1u) % 3u; pend->seed0 = s2 + stride;
                            pend->slot1 = (cur + 2u) % 3u; pend->seed1 = s2 + 2u * stride; }
            }
        } else f = flags + (s2 < seed ? 1u : 0u); // Philox kind: seed + stride wrapped past 2^32 -> next high word
        load(kind, make_uint4(s2, 0u, s2 + stride, f), rng_win_ptr(rng, N, W, lane, kind == 0 ? RNGF_CUR(f) : 0u), W, key);
        prefetch();
    }
    EWN_DEV void begin_step() { if (kind == 1) ps.n = (ps.n + 3u) & ~3u; }
    EWN_DEV int first_dice(int cube_num)
    {
        if (kind != 1) return randint(1, cube_num + 1);      // roll_dice, envs/ewn.py:90-91
        const u32 w = fmix32(seed_mix() ^ fmix32(ps.k0 ^ 0x454E5631u) ^ (ps.k1 * 0x9E3779B1u));
        return 1 + (int)__umulhi(w, (u32)cube_num);           // bias <= cube_num / 2^32
    }
    EWN_DEV int lemire(u32 range)
    {
        if (range <= 1u) return 0;
        u64 m = (u64)next() * range;
        if ((u32)m < range) {                                  // probability range / 2^32
            const u32 t = (0u - range) % range;
            int guard = 0;
            while ((u32)m < t && ++guard < 64) m = (u64)next() * range;
        }
        return (int)(m >> 32);
    }
    EWN_DEV int randint(int lo, int hi)
    {
        if (kind == 1) return lo + lemire((u32)(hi - lo));
        const u32 rng = (u32)(hi - lo - 1);
        if (rng == 0) return lo;
        u32 mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        u32 v;
        int guard = 0;
        do { v = next() & mask; } while (v > rng && ++guard < 4096);
        return lo + (int)(v > rng ? 0u : v);
    }
};
