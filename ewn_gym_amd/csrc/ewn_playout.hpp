// Random playouts for cube_layer <= 3 (six cubes a side): classical_policies/mcts.py:21-45 (flat Monte-Carlo "MCTS") and
// envs/minimax_ewn.py:215-238 (MinimaxEnv.simulate).
//
// Measured on MI355X before this file existed (7x7, 400 playouts per root move, 32 768 observations): the generic
// playout (u64 occupancy + 6-bit position fields + masked-rejection draws) retired ~265 VALU wave-instructions per ply
// and the launch sat at the integer-issue limit of the chip, with a third of every wave idle waiting for its longest
// playout.  This version is built for that limit:
//   * a game is 12 bytes in 4 registers: one byte per cube = its cell in row*8+col numbering, bit 6 set once the
//     cube is off the board.  A capture (either colour, envs/ewn.py:254-258) is one SWAR byte-compare per register;
//   * the set of cubes still on the board is read off bit 6 of those bytes, laid out so that "nearest larger / smaller
//     number" (find_near_cube, envs/ewn.py:144-176) is one ffbl / ffbh;
//   * legal directions by cell and "k-th legal move" by (mask, k) are two small LDS tables built by the block itself;
//   * one 32-bit draw per ply yields both the dice and the move index;
//   * a lane plays PLAYOUT_CHAIN playouts back to back, restarting on its own, so a wave's length is the longest SUM of
//     eight playouts rather than eight times the longest playout (mean/max 0.69 -> 0.87 on 7x7).
#pragma once
#include "ewn_core.hpp"

#define PLAYOUT_CHAIN 8

// Randomness of playout r of a (observation, root move) cell: chunk c = r / PLAYOUT_CHAIN shares ONE Philox block (ctr =
// {0, c1, c2 + c * PLAYOUT_CHAIN, tag}); playout j = r % PLAYOUT_CHAIN of the chunk starts a 32-bit LCG at
// fmix32(block[0] + j * 0x9E3779B9).  A ply draws once: the top 24 bits of the xor-folded state, times six, give the dice
// in bits 24-26 and a 24-bit fraction that picks the move.  The reference's `random.randint` is never seeded
// (mcts.py:29-32): parity is statistical either way; the oracle mirrors this generator so HIP-vs-oracle is bit-exact.
struct PlayoutRng {
    u32 s;
    EWN_DEV static u32 chunk_word(u32 c1, u32 c2, u32 tag, u64 key)
    {
        u32 o[4];
        philox4x32_10(0u, c1, c2, tag, (u32)key, (u32)(key >> 32), o);
        return o[0];
    }
    EWN_DEV void seed(u32 word, u32 j) { s = fmix32(word + j * 0x9E3779B9u); }
    EWN_DEV void draw(u32 &d, u32 &frac) // d in 0..5 (random.randint(1, 6) - 1, mcts.py:29), frac in [0, 2^24)
    {
        s = s * 0x2C9277B5u + 0xAC564B05u;
        const u32 t = ((s ^ (s >> 15)) >> 8) * 6u;
        d = t >> 24; frac = t & 0xFFFFFFu;
    }
    EWN_DEV static u32 pick(u32 frac, u32 n) { return (frac * n) >> 24; } // uniform index into n <= 6 legal moves
};

struct PlayTab {
    uint8_t legal[2][128]; // [side][cell]: bit 0 sideways, bit 1 along the column, bit 2 diagonal stays on the board
    uint8_t nth[512];      // [mask * 8 + k]: k-th set bit of a 6-bit legal mask as (second cube ? 0x80 : 0) | cell delta
};

EWN_DEV void playtab_build(PlayTab *T, int S) // every thread of the block; the caller synchronises
{
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

// cube k (0-based) -> bit 8 * (k >> 1) + (k & 1): increasing in k, so ffbl / ffbh find the neighbours
EWN_DEV u32 alive_bits(u32 lo, u32 hi)
{
    const u32 a = ~lo & 0x40404040u, b = ~hi & 0x40404040u;
    return (a | (b << 1)) >> 6;
}

EWN_DEV PState pstate_from(const Geom &g, const GState<1> &s)
{
    u32 w[4] = { 0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u };
    const u32 magic = 65536u / (u32)g.S + 1u; // (c * magic) >> 16 == c / S for c < 64, S in 3..8
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        #pragma unroll
        for (int side = 0; side < 2; side++) {
            const u32 alive = side == 0 ? s.aliveP : s.aliveN;
            const u32 c = (u32)(side == 0 ? pos_of<0>(s, k) : pos_of<1>(s, k));
            const u32 row = (c * magic) >> 16, cell8 = row * 8u + (c - row * (u32)g.S);
            const int sh = 8 * (k >> 1);
            if ((alive >> k) & 1u) w[side * 2 + (k & 1)] = (w[side * 2 + (k & 1)] & ~(0xFFu << sh)) | (cell8 << sh);
        }
    }
    return PState{ w[0], w[1], w[2], w[3] };
}

// One uniformly random legal move of SIDE (mcts.py:29-35).  A: in = bits of SIDE's cubes on the board, out = the other
// side's after the move.  Returns true when the move ends the game, which the mover then has won: a move can reach only
// the mover's own goal corner and can empty only the other side (the mover itself stays), envs/ewn.py:131-142.
template <int SIDE>
EWN_DEV bool playout_ply(const PlayTab *T, PState &st, u32 &A, PlayoutRng &ps, u32 goal)
{
    u32 &mlo = SIDE == 0 ? st.plo : st.nlo, &mhi = SIDE == 0 ? st.phi : st.nhi;
    u32 d, frac;
    ps.draw(d, frac);
    // find_near_cube on the bit layout of alive_bits()
    const u32 fpos = ((d >> 1) << 3) | (d & 1u), bit = 1u << fpos;
    const u32 upm = A & (0xFFFFFFFEu << fpos), dnm = A & (bit - 1u);
    const bool ex = (A & bit) != 0u;
    const u32 pos_up = (u32)__ffs((int)upm) - 1u, pos_dn = 31u - (u32)__clz((int)dnm);
    const u32 c0 = ex ? fpos : (upm ? pos_up : pos_dn), c1 = pos_dn; // legal list: larger neighbour's moves, then smaller's
    const bool has1 = !ex && upm != 0u && dnm != 0u;
    const u32 p0 = (((c0 & 1u) ? mhi : mlo) >> (c0 & 24u)) & 0x7Fu;
    const u32 p1 = (((c1 & 1u) ? mhi : mlo) >> (c1 & 24u)) & 0x7Fu; // junk < 128 when there is no second cube: masked below
    const u32 m0 = T->legal[SIDE][p0], m1 = has1 ? (u32)T->legal[SIDE][p1] : 0u;
    const u32 okm = m0 | (m1 << 3);
    const u32 r = T->nth[okm * 8u + PlayoutRng::pick(frac, (u32)__popc(okm))];
    const bool second = r > 127u;
    const u32 p = second ? p1 : p0, c = second ? c1 : c0, delta = r & 0x7Fu;
    const u32 q = SIDE == 0 ? p + delta : p - delta;
    // whatever stands on q leaves the board: byte == q  <=>  bit 7 of ((byte ^ q ^ 0x7F) + 1); bytes stay below 0x80
    const u32 qb = (q ^ 0x7Fu) * 0x01010101u;
    st.plo |= (((st.plo ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    st.phi |= (((st.phi ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    st.nlo |= (((st.nlo ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    st.nhi |= (((st.nhi ^ qb) + 0x01010101u) >> 1) & 0x40404040u;
    const u32 mv = (p ^ q) << (c & 24u);
    mlo ^= (c & 1u) ? 0u : mv;
    mhi ^= (c & 1u) ? mv : 0u;
    A = SIDE == 0 ? alive_bits(st.nlo, st.nhi) : alive_bits(st.plo, st.phi);
    return q == goal || A == 0u;
}

// nj playouts from b0 with FIRST to move; returns how many TOP_LEFT won (mcts.py:39-41)
template <int FIRST>
EWN_DEV int run_playouts(const PlayTab *T, const PState &b0, int S, u32 word, int nj)
{
    const u32 goal0 = 9u * (u32)(S - 1);
    const u32 A0 = FIRST == 0 ? alive_bits(b0.plo, b0.phi) : alive_bits(b0.nlo, b0.nhi);
    PState st = b0;
    u32 A = A0;
    PlayoutRng ps;
    ps.seed(word, 0u);
    int j = 0, w = 0;
    // a cube only ever moves towards its goal corner: a playout is at most 12 * 2 * (S - 1) plies; the cap is a guard only
    for (int it = 0; it < PLAYOUT_CHAIN * 256 && j < nj; it++) {
        bool fin = playout_ply<FIRST>(T, st, A, ps, FIRST == 0 ? goal0 : 0u);
        bool second_won = false;
        if (!fin) fin = second_won = playout_ply<1 - FIRST>(T, st, A, ps, FIRST == 0 ? 0u : goal0);
        if (fin) {
            w += ((FIRST == 0) != second_won) ? 1 : 0; // the side that made the last move won
            j++;
            st = b0; A = A0;
            ps.seed(word, (u32)j);
        }
    }
    return w;
}
