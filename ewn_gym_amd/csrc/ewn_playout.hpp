// Random playouts for cube_layer <= 3 (six cubes a side): classical_policies/mcts.py:21-45 (flat Monte-Carlo "MCTS") and
// envs/minimax_ewn.py:215-238 (MinimaxEnv.simulate).
//
// Measured on MI355X before this file existed (7x7, 400 playouts per root move, 32 768 observations): the generic
// playout (u64 occupancy + 6-bit position fields + masked-rejection draws) retired ~265 VALU wave-instructions per ply
// and the launch sat at the integer-issue limit of the chip (SQ_ACTIVE_INST_VALU ~ 100 % of SIMD cycles), with a third
// of every wave idle waiting for its longest playout.  This version is built for that limit:
//   * a game is 12 bytes in 4 registers: one byte per cube = its cell in row*8+col numbering, bit 6 set once the
//     cube is off the board.  A capture (either colour, envs/ewn.py:254-258) is one SWAR byte-compare per register;
//   * dice -> cube(s) to move (find_near_cube, envs/ewn.py:144-176), legal directions by cell and "k-th legal move" by
//     (mask, k) are three small LDS tables built by the block itself; the first one returns v_perm byte selectors;
//   * one 32-bit draw per ply yields both the dice and the move index;
//   * a lane plays its share of a root move's playouts back to back, restarting on its own, so a wave's length is the
//     longest SUM of several playouts rather than several times the longest playout (mean/max 0.69 -> 0.87 on 7x7);
//   * a root move's playouts belong to an aligned group of 8..64 lanes and a block visits only the root moves that
//     exist (about half of the 6 slots), so waves do not carry dead lanes.
#pragma once
#include "ewn_core.hpp"

// Randomness of the playouts of one observation: ONE Philox block (ctr = {0, c1, 0, tag}) gives a word; playout number x
// (root move i, playout r: x = i * total + r) starts a 32-bit LCG at fmix32(word + x * 0x9E3779B9).  A ply draws once: the
// top 24 bits of the xor-folded state, times six, give the dice in bits 24-26 and a 24-bit fraction that picks the move.
// The reference's `random.randint` is never seeded (mcts.py:29-32): parity is statistical either way; the oracle mirrors
// this generator, so HIP-vs-oracle is bit-exact whichever lane plays which playout.
struct PlayoutRng {
    u32 s;
    EWN_DEV static u32 obs_word(u32 c1, u32 tag, u64 key)
    {
        u32 o[4];
        philox4x32_10(0u, c1, 0u, tag, (u32)key, (u32)(key >> 32), o);
        return o[0];
    }
    EWN_DEV void seed(u32 word, u32 x) { s = fmix32(word + x * 0x9E3779B9u); }
    EWN_DEV void draw(u32 &d, u32 &frac) // d in 0..5 (random.randint(1, 6) - 1, mcts.py:29), frac in [0, 2^24)
    {
        s = s * 0x2C9277B5u + 0xAC564B05u;
        const u32 t = ((s ^ (s >> 15)) >> 8) * 6u;
        d = t >> 24; frac = t & 0xFFFFFFu;
    }
    EWN_DEV static u32 pick(u32 frac, u32 n) { return (frac * n) >> 24; } // uniform index into n <= 6 legal moves
};

struct PlayTab {
    uint16_t sel[512];     // [cubes on board (6 bits) * 8 + dice - 1]: v_perm byte selectors of the cube(s) the dice allows
    uint8_t legal[2][128]; // [side][cell]: bit 0 sideways, bit 1 along the column, bit 2 diagonal stays on the board
    uint8_t nth[512];      // [mask * 8 + k]: k-th set bit of a 6-bit legal mask as (second cube ? 0x80 : 0) | cell delta
};

// byte of cube k (0-based) in the register pair {hi, lo}: even k in lo, odd k in hi
EWN_DEV u32 playout_byte_of(int k) { return (u32)((k & 1) * 4 + (k >> 1)); }

EWN_DEV void playtab_build(PlayTab *T, int S) // every thread of the block; the caller synchronises
{
    for (int e = threadIdx.x; e < 512; e += blockDim.x) {
        const u32 alive = (u32)e >> 3;
        const int d = (e & 7) + 1;
        u32 r = 0x0303u; // byte 3 of lo: permanently off the board, its "cell" 0x40 has no legal direction
        if (alive != 0u && d <= 6) {
            const CubeSel cs = select_cubes(alive, d);
            // legal list order (envs/ewn.py:338-375): the dice cube, else the larger neighbour's moves then the smaller's
            const int first = cs.exact ? cs.k_exact : (cs.has_up ? cs.k_up : cs.k_down);
            const bool two = !cs.exact && cs.has_up && cs.has_down;
            r = playout_byte_of(first) | ((two ? playout_byte_of(cs.k_down) : 3u) << 8);
        }
        T->sel[e] = (uint16_t)r;
    }
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
        const int side = e >> 7, c = e & 127, row = c >> 3, col = c & 7;
        u32 m = 0;
        if (c < 64 && row < S && col < S) {
            const bool cok = side == 0 ? col < S - 1 : col > 0, rok = side == 0 ? row < S - 1 : row > 0;
            m = (cok ? 1u : 0u) | (rok ? 2u : 0u) | ((cok && rok) ? 4u : 0u);
        }
        T->legal[side][c] = (uint8_t)m;
    }
    for (int e = threadIdx.x; e < 512; e += blockDim.x) {
        u32 m = (u32)e >> 3;
        for (int i = 0; i < (e & 7); i++) m &= m - 1;
        u32 r = 0;
        if (m) {
            const int j = __ffs((int)m) - 1, slot = j >= 3 ? 1 : 0, dir = j - 3 * slot;
            r = ((u32)slot << 7) | (dir == 0 ? 1u : (dir == 1 ? 8u : 9u));
        }
        T->nth[e] = (uint8_t)r;
    }
}

// plo/nlo: cubes 1, 3, 5 (bytes 0..2), phi/nhi: cubes 2, 4, 6; byte 3 of every word is a permanent "off the board"
struct PState { u32 plo, phi, nlo, nhi; };

// bit k = cube k (0-based) is on the board: bit 6 of the bytes, gathered with one 24-bit multiply
EWN_DEV u32 alive_bits(u32 lo, u32 hi)
{
    const u32 a = ~lo & 0x40404040u, b = ~hi & 0x40404040u;
    const u32 z = (a | (b << 1)) >> 6;                 // cube k at bit 8 * (k >> 1) + (k & 1)
    return (__umul24(z, 0x1041u) >> 12) & 0x3Fu;       // bit pairs 0-1, 8-9, 16-17 land on 12-13, 14-15, 16-17
}

// The 2^gl lanes of a group (one wave at most) read one board together: lane t looks at cells t, t + 2^gl, ... and xors
// the cubes it finds into the group's four LDS words gw[0..3]; every lane returns the finished position.  LDS operations of
// one wave execute in program order, so no barrier is needed.  (A cube number that appears twice garbles that byte; the
// tables are sized so that any byte value is a safe index.)
EWN_DEV PState pstate_load(const Geom &g, const int8_t *board, int lane, int tc, u32 *gw)
{
    if (lane < 4) gw[lane] = 0x40404040u;
    __builtin_amdgcn_wave_barrier();
    const u32 magic = 65536u / (u32)g.S + 1u; // (c * magic) >> 16 == c / S for c < 64, S in 3..8
    for (int c = lane; c < g.cells; c += tc) {
        const int v = board[c], k = (v > 0 ? v : -v) - 1;
        if (v != 0 && k < 6) {
            const u32 row = ((u32)c * magic) >> 16, cell8 = row * 8u + ((u32)c - row * (u32)g.S);
            atomicXor(&gw[(v < 0 ? 2 : 0) + (k & 1)], (0x40u ^ cell8) << (8 * (k >> 1)));
        }
    }
    __builtin_amdgcn_wave_barrier();
    const PState st = { gw[0], gw[1], gw[2], gw[3] };
    __builtin_amdgcn_wave_barrier();
    return st;
}

// some byte of w equals cell q (q < 64)
EWN_DEV bool pstate_on(u32 w, u32 q)
{
    const u32 q7 = q ^ 0x7Fu, qb = __builtin_amdgcn_perm(q7, q7, 0u);
    return (((w ^ qb) + 0x01010101u) & 0x80808080u) != 0u;
}

// check_win, envs/ewn.py:131-142; top_left_won as mcts.py:39-41 reads it
EWN_DEV bool pstate_is_win(const PState &st, int S, bool &top_left_won)
{
    const u32 goal = 9u * (u32)(S - 1);
    const bool p_home = pstate_on(st.plo, goal) || pstate_on(st.phi, goal), n_home = pstate_on(st.nlo, 0u) || pstate_on(st.nhi, 0u);
    const bool p_none = alive_bits(st.plo, st.phi) == 0u, n_none = alive_bits(st.nlo, st.nhi) == 0u;
    top_left_won = p_home || n_none;
    return p_home || n_home || p_none || n_none;
}

// One uniformly random legal move of SIDE (mcts.py:29-35).  A: in = SIDE's cubes on the board, out = the other side's
// after the move.  Returns true when the move ends the game, which the mover then has won: a move can reach only the
// mover's own goal corner and can empty only the other side (the mover itself stays), envs/ewn.py:131-142.
template <int SIDE, class Pick>
EWN_DEV bool playout_move(const PlayTab *T, PState &st, u32 &A, u32 d, Pick &&pick_of, u32 goal)
{
    u32 &mlo = SIDE == 0 ? st.plo : st.nlo, &mhi = SIDE == 0 ? st.phi : st.nhi;
    const u32 sel = T->sel[A * 8u + d];
    const u32 pp = __builtin_amdgcn_perm(mhi, mlo, sel | 0x0C0C0000u); // cells of the first and the second candidate cube
    const u32 p0 = pp & 0xFFu, p1 = pp >> 8;                             // no second cube: p1 = 0x40, legal[..][0x40] = 0
    const u32 okm = (u32)T->legal[SIDE][p0] | ((u32)T->legal[SIDE][p1] << 3);
    const u32 r = T->nth[okm * 8u + pick_of((u32)__popc(okm))];
    const bool second = r > 127u;
    const u32 p = second ? p1 : p0, byte = second ? sel >> 8 : sel & 0xFFu, delta = r & 0x7Fu;
    const u32 q = SIDE == 0 ? p + delta : p - delta;
    // whatever stands on q leaves the board: byte == q  <=>  bit 7 of ((byte ^ q ^ 0x7F) + 1); bytes stay below 0x80
    const u32 q7 = q ^ 0x7Fu, qb = __builtin_amdgcn_perm(q7, q7, 0u); // byte 0 into all four bytes
    st.plo |= (((st.plo ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    st.phi |= (((st.phi ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    st.nlo |= (((st.nlo ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    st.nhi |= (((st.nhi ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    const u64 mv = (u64)(p ^ q) << (byte * 8u);
    mlo ^= (u32)mv;
    mhi ^= (u32)(mv >> 32);
    A = SIDE == 0 ? alive_bits(st.nlo, st.nhi) : alive_bits(st.plo, st.phi);
    return q == goal || A == 0u;
}

template <int SIDE>
EWN_DEV bool playout_ply(const PlayTab *T, PState &st, u32 &A, PlayoutRng &ps, u32 goal)
{
    u32 d, frac;
    ps.draw(d, frac);
    return playout_move<SIDE>(T, st, A, d, [&](u32 n) { return PlayoutRng::pick(frac, n); }, goal);
}

// TOP_LEFT's idx-th legal action for this dice (get_legal_actions order, envs/ewn.py:338-375): the root move of mcts.py:21-24
EWN_DEV bool playout_root_move(const PlayTab *T, PState &st, int S, int dice, int idx)
{
    u32 A = alive_bits(st.plo, st.phi);
    return playout_move<0>(T, st, A, (u32)(dice - 1), [&](u32) { return (u32)idx; }, 9u * (u32)(S - 1));
}

// The playouts x0 .. x0 + total - 1 of one start position b0, FIRST to move, shared by the 2^gl lanes of a group: lane t
// starts with playout t and takes the next unplayed one from *next (an LDS counter of the group, preset to 2^gl) whenever
// it finishes, so the lanes of a wave stay busy until the position runs out of playouts.  Which lane plays which playout
// does not matter: playout x always starts from seed(word, x).  Returns how many of this lane's playouts TOP_LEFT won
// (mcts.py:39-41).
template <int FIRST>
EWN_DEV int run_playouts(const PlayTab *T, const PState &b0, int S, u32 word, u32 x0, int lane, int total, int *next)
{
    const u32 goal0 = 9u * (u32)(S - 1);
    const u32 A0 = FIRST == 0 ? alive_bits(b0.plo, b0.phi) : alive_bits(b0.nlo, b0.nhi);
    PState st = b0;
    u32 A = A0;
    PlayoutRng ps;
    ps.seed(word, x0 + (u32)lane);
    bool busy = lane < total;
    int w = 0;
    // a cube only ever moves towards its goal corner: a playout is at most 12 * 2 * (S - 1) plies; the cap is a guard only
    for (int it = 0; it < total * 256 && busy; it++) {
        bool fin = playout_ply<FIRST>(T, st, A, ps, FIRST == 0 ? goal0 : 0u);
        bool second_won = false;
        if (!fin) fin = second_won = playout_ply<1 - FIRST>(T, st, A, ps, FIRST == 0 ? 0u : goal0);
        if (fin) {
            w += ((FIRST == 0) != second_won) ? 1 : 0; // the side that made the last move won
            const int r = atomicAdd(next, 1);
            busy = r < total;
            st = b0; A = A0;
            ps.seed(word, x0 + (u32)r);
        }
    }
    return w;
}

// lanes per group of playouts sharing one start position: 8..64, about PLAYOUT_PER_LANE playouts or more per lane
#ifndef PLAYOUT_PER_LANE
#define PLAYOUT_PER_LANE 6
#endif
static inline int playout_group_log2(int total)
{
    int l = 3;
    while (l < 6 && (total >> (l + 1)) >= PLAYOUT_PER_LANE) l++;
    return l;
}
