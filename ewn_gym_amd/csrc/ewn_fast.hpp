// ewn_fast.hpp -- search tables of the specialised depth-3 'hybrid' expectiminimax (built on the host by
// build_fast_tables, read from LDS by d3_search in ewn_step_d3.hpp) and the reasoning behind them.
//
// What classical_policies/minimax.py:19-73 computes at max_depth=3:
//   for each root move a_i (<= 6, reference order):   B1 = apply(B, a_i)
//     win(B1)            -> v_i = evaluate(B1) = 10
//     else v_i = sum_{d=1..6} w_d / 6,  w_d = min over BOTTOM_RIGHT's replies to dice d of
//                        evaluate(B2), the scan stopping early once w_d <= alpha
//   best = first maximal v_i, alpha = running best.
//
// How it is computed here, per lane, with no recursion, no fp64 division, no divergence
// between the lanes of a wavefront (every loop has a fixed trip count and is predicated):
//  * cells are renumbered in "ring order" (ascending min(row,col)): the hybrid
//    heuristic's min-distance of a side (both sides are measured to the bottom-right
//    corner, envs/minimax_ewn.py:67-76) is then a function of the HIGHEST set bit of its
//    occupancy mask -- one v_ffbh + one LDS byte instead of a scan;
//  * a leaf value depends only on (level, count) of either side: <= (8S)^2 cases.  The host
//    tabulates them in fp64 exactly as Python evaluates them, sorts the distinct values
//    and hands the kernel an integer RANK per case, so every `val < worst` /
//    `worst <= alpha` comparison of the reference becomes an integer comparison with the
//    identical outcome, and w_d/6 is a table read of the host-computed quotient;
//  * the <= 216 leaves collapse to <= 108 distinct positions (a reply is a (cube, dir)
//    pair; which two cubes a dice value selects only decides which of those are scanned);
//  * the reference's early exit "w <= alpha" is replayed in closed form: along a cube's replies
//    the running minimum is non-increasing, so the loop stops at the LARGEST prefix minimum
//    that is <= alpha (`cut`, 0 if none); for the (larger-neighbour F, smaller-neighbour G)
//    cube pair a dice selects the result is  cutF ? cutF : cutG ? cutG : min(p2F, p2G).
#pragma once
#include "ewn_core.hpp"

#define FAST_NV 1024   // distinct leaf values (incl. +-10) get ranks 1..nv in 10 bits; 0 = "none", 1023 = "no such reply"
// Ranks travel through the search as BYTE OFFSETS into val[] / val6[] (8 x rank): order and equality are those of the ranks,
// and a value lookup needs no address arithmetic (measured round 2: every u16 / f64 table read cost one half-rate VALU shift).
#define FAST_NONE8 (1023u * 8u)             // "no such reply": val = +inf
#define FAST_KNONE (0x8000u | FAST_NONE8)   // key (see d3_search) of a cube that is not on the board

template <int S> struct MaskOf { typedef u32 type; };
template <> struct MaskOf<6> { typedef u64 type; };
template <> struct MaskOf<7> { typedef u64 type; };
template <> struct MaskOf<8> { typedef u64 type; };

template <int S>
struct FastTab {
    static constexpr int CELLS = S * S;
    // one side's (level t, count n) is the 6-bit index t*8 + n (t <= 7, n <= 6); a leaf is the pair (ix of the P side, iy of the
    // N side).  rank[(ix << 6) | iy] = 8 x the rank of (0 + x) - y among the table's distinct values: 1 = lowest ... nv = +10
    uint16_t rank[64 * 64];
    double val[FAST_NV];                         // rank -> leaf value (ascending); val[0] = -inf (rank 0 = "none"), unused ranks = +inf
    double val6[FAST_NV];                        // rank -> value / 6.0 (IEEE quotient, host-computed)
    uint8_t lvl_guard[8];                        // lvl[-1]: v_ffbh_u32 answers -1 for an empty 32-bit mask; level 0 + count 0 = the "-10" row
    uint8_t lvl[72];                             // clz(mask) -> level << 3 (entry for an empty mask: 0)
    uint8_t ri[64];                              // canonical row-major cell -> ring index
    // position bytes (ewn_step_d3.hpp: one byte per cube = its ring index, bit 6 set once it is off the board) index the
    // next five tables directly, so they have 128 entries: 64..127 = "no such cube" -> off board / no legal direction
    uint8_t nbp[3][128];                         // mover   (TOP_LEFT):     ring index -> destination ring index, 255 = off board
    uint8_t nbn[3][128];                         // replier (BOTTOM_RIGHT): ring index -> destination ring index, 255 = off board
    uint8_t lgp[128], lgn[128];                  // ring index -> legal directions of a TOP_LEFT / BOTTOM_RIGHT cube there (bits 0..2)
    int32_t nv, ri_origin, m10, pad1;            // number of ranks (= rank of +10); ring index of cell (0,0); rank of -10
    // for the fused step kernel (ewn_step_d3.hpp), which keeps the game in canonical ring space:
    uint8_t real_of_ring[64];                    // ring index -> REAL row-major cell (canonical cell = S*S-1 - real cell)
    uint64_t init_P, init_N, init_posP, init_posN; // the start position (envs/ewn.py:94-107): opponent = P side, agent = N side
    // find_near_cube / get_legal_actions (envs/ewn.py:144-176, 338-375) as a table: [cubes on board (bit k = cube k+1) * 8 +
    // dice - 1] -> first | second << 8 | first_is_larger_neighbour << 15, where first / second are the (0-based) cubes whose
    // moves make up the legal list in the reference's order; 6 = none (byte 6 of a position word is permanently off board)
    uint16_t sel[512];
    uint8_t nth[512];                            // [6-bit legal mask * 8 + k] -> index of its k-th set bit (0 when there is none)
    // ring index -> ring index of the same square seen from the other side (board rotated by 180 degrees, envs/ewn.py:289-296);
    // an involution.  The rollout kernel uses it to hand the AGENT's position to the same search (ewn_rollout.hpp).
    uint8_t rot[64];
    // ring index -> max(row, col) of the canonical square = the REAL square's distance to the real bottom-right corner, which is
    // what the shaped training env's evaluate() measures for both sides (envs/training_ewn.py:94-96, envs/minimax_ewn.py:67-76)
    uint8_t dtl[64];
    // the replier's (BOTTOM_RIGHT's) moves for d3_search's per-search setup, [direction][position byte] (bytes 64..127 = no such cube):
    // the destination's bit in the occupancy mask (0: the move leaves the board), and what steers the leaf's rank address --
    // keep | fixed << 16 with (address & keep) | fixed: a normal reply keeps the address, a reply that does not exist reads rank[0]
    // ("no such reply"), a reply onto the origin rank[1] (-10, envs/minimax_ewn.py:45-47).  Two LDS reads where the setup spent
    // eight VALU instructions per reply; the slot-task rollout kernel runs that setup once per THREE roots.
    uint32_t rkf[3][128];
    typename MaskOf<S>::type rsetT[3][128];
};

// table images per board size: [heuristic image][variant]; heuristic images: 'hybrid', 'min_dist', 'attk' (functions of each side's
// (level, count); 'two_min_dist' needs the second-nearest cube as well and stays on the generic kernels)
// The fourth image serves 'two_min_dist' (sum of each side's two smallest distances, envs/minimax_ewn.py:133-178): a side's index is
// that SUM (0 .. 2(S-1)) instead of (level, count), the image's `lvl` table holds distances instead of levels, and the search
// looks at a mask's two highest bits (template parameter H2 of d3_search / d5_search: the stateless predict kernel and the fused
// step kernel k_step_d3 are instantiated with it, the K-step rollout kernels are not).
#define FAST_HEUR_IMAGES 4
static inline int fast_heur_image(int heur) { return heur == 0 ? 0 : (heur == 1 ? 1 : (heur == 3 ? 2 : (heur == 2 ? 3 : -1))); }
static inline bool fast_heur_lean(int heur) { return heur == 0 || heur == 1 || heur == 3; } // images k_step_d3 / k_rollout_d3 can run
static inline bool fast_heur_step(int heur) { return fast_heur_lean(heur) || heur == 2; }   // images k_step_d3 can run ('two_min_dist': its own instances)

// LDS / device image size: the struct padded to 4 KiB so the LDS-DMA copy needs no tail handling
#define FAST_TAB_BYTES(S) ((int)((sizeof(FastTab<S>) + 4095) / 4096 * 4096))


EWN_DEV int clz_m(u32 m) { return __clz((int)m); }       // 32 for m == 0
EWN_DEV int clz_m(u64 m) { return __clzll((long long)m); } // 64 for m == 0
// the same without the clamp, for a mask that is non-zero wherever the result matters: the raw instruction answers -1 for 0 and the
// table read that follows then lands two bytes in front of the table, inside LDS; the caller masks its value off.  Inline
// assembly on purpose: __builtin_clz(0) is undefined behaviour the optimiser is free to act on.
EWN_DEV int clz_nz(u32 m) { int r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(m)); return r; }
EWN_DEV int clz_nz(u64 m)
{
    int hi, lo;
    asm("v_ffbh_u32 %0, %1" : "=v"(hi) : "v"((u32)(m >> 32)));
    asm("v_ffbh_u32 %0, %1" : "=v"(lo) : "v"((u32)m));
    return (u32)(m >> 32) ? hi : 32 + lo;
}
EWN_DEV int popc_m(u32 m) { return __popc(m); }
EWN_DEV int popc_m(u64 m) { return __popcll(m); }
// index into FastTab::lvl: the raw v_ffbh of a 32-bit mask (-1 for an empty one: lvl[-1] is the guard byte, inside the table
// image; an out-of-range LDS read would answer 0 as well), the clamped count of a 64-bit one
EWN_DEV int lvl_index(u32 m) { return clz_nz(m); }
EWN_DEV int lvl_index(u64 m) { return clz_m(m); }
// one side's 6-bit (level, count) index: one byte read + v_bcnt_u32_b32 with its accumulate operand
template <int S> EWN_DEV u32 ft_side(const FastTab<S> *Tb, typename MaskOf<S>::type m) { return (u32)Tb->lvl[lvl_index(m)] + (u32)popc_m(m); }
// 'two_min_dist' image: the side's index is the sum of the distances of its two highest cells (one cell: its distance alone;
// none: 0, the row that answers -10)
EWN_DEV u32 drop_top(u32 m) { return m ? m & ~(0x80000000u >> __clz((int)m)) : 0u; }
EWN_DEV u64 drop_top(u64 m) { return m ? m & ~(0x8000000000000000ull >> __clzll((long long)m)) : 0ull; }
template <int S> EWN_DEV u32 ft_side2(const FastTab<S> *Tb, typename MaskOf<S>::type m)
{
    return (u32)Tb->lvl[lvl_index(m)] + (u32)Tb->lvl[lvl_index(drop_top(m))];
}
template <int S, bool H2> EWN_DEV u32 ft_side_h(const FastTab<S> *Tb, typename MaskOf<S>::type m)
{
    if constexpr (H2) return ft_side2<S>(Tb, m); else return ft_side<S>(Tb, m);
}
// byte address of a leaf's entry inside rank[]
EWN_DEV u32 ft_addr(u32 ix, u32 iy) { return (ix << 7) | (iy << 1); }
template <int S> EWN_DEV u32 ft_rank8(const FastTab<S> *Tb, u32 addr) { return *(const uint16_t *)((const char *)Tb->rank + addr); }
template <int S> EWN_DEV double ft_val(const FastTab<S> *Tb, u32 r8) { return *(const double *)((const char *)Tb->val + r8); }
template <int S> EWN_DEV double ft_val6(const FastTab<S> *Tb, u32 r8) { return *(const double *)((const char *)Tb->val6 + r8); }

// ---------------------------------------------------------------- host: table construction
#include <algorithm>
#include <cstring>
#include <vector>

// variant 0: max_depth 3 (a leaf is evaluate()).  variant 1: max_depth 4, where the search goes one chance node and one
// depth-0 max node further before evaluating (classical_policies/minimax.py:19-73): a non-terminal leaf is then
// sum over six dice of evaluate()/6, added in order -- the same tables with that value in place of evaluate().
// heur: EWN_H_HYBRID (0), EWN_H_MIN_DIST (1) or EWN_H_ATTK (3): the leaf value as envs/minimax_ewn.py:29-213 computes it
template <int S>
static int build_fast_tables(FastTab<S> *T, int variant = 0, int heur = 0)
{
    constexpr int IXN = 64;
    memset(T, 0, sizeof(*T));
    // ring order: ascending level t = min(row, col); (0,0) first, (S-1,S-1) last
    int ring_of_rm[64], rm_of_ring[64], n = 0;
    for (int t = 0; t < S; t++)
        for (int i = 0; i < S; i++)
            for (int j = 0; j < S; j++)
                if ((i < j ? i : j) == t) { ring_of_rm[i * S + j] = n; rm_of_ring[n] = i * S + j; n++; }
    int level_of_ring[64];
    for (int q = 0; q < S * S; q++) { const int c = rm_of_ring[q]; level_of_ring[q] = (c / S < c % S) ? c / S : c % S; }
    for (int c = 0; c < 64; c++) T->ri[c] = c < S * S ? (uint8_t)ring_of_rm[c] : 0;
    T->ri_origin = ring_of_rm[0];
    for (int d = 0; d < 3; d++)
        for (int q = 0; q < 128; q++) {
            T->nbp[d][q] = 255; T->nbn[d][q] = 255;
            if (q >= S * S) continue;
            const int c = rm_of_ring[q], i = c / S, j = c % S;
            const int di = d == 0 ? 0 : 1, dj = d == 1 ? 0 : 1;
            if (i + di < S && j + dj < S) T->nbp[d][q] = (uint8_t)ring_of_rm[(i + di) * S + (j + dj)];
            if (i - di >= 0 && j - dj >= 0) T->nbn[d][q] = (uint8_t)ring_of_rm[(i - di) * S + (j - dj)];
        }
    for (int d = 0; d < 3; d++)
        for (int q = 0; q < 128; q++) {
            const int dn = T->nbn[d][q];
            const bool ok = dn != 255, home = ok && dn == T->ri_origin;
            T->rsetT[d][q] = ok ? (typename MaskOf<S>::type)1 << dn : (typename MaskOf<S>::type)0;
            T->rkf[d][q] = ((ok && !home) ? 0xFFFFu : 0u) | ((home ? 2u : 0u) << 16);
        }
    for (int q = 0; q < 128; q++) {
        T->lgp[q] = T->lgn[q] = 0;
        for (int d = 0; d < 3; d++) {
            if (T->nbp[d][q] != 255) T->lgp[q] |= (uint8_t)(1u << d);
            if (T->nbn[d][q] != 255) T->lgn[q] |= (uint8_t)(1u << d);
        }
    }
    for (int e = 0; e < 512; e++) {
        const unsigned alive = (unsigned)e >> 3;
        const int d = e & 7; // dice - 1; 6 and 7 cannot be rolled but keep find_near_cube's answer for them
        int up = -1, down = -1;
        for (int k = d + 1; k < 6; k++) if ((alive >> k) & 1u) { up = k; break; }
        for (int k = (d < 6 ? d : 6) - 1; k >= 0; k--) if ((alive >> k) & 1u) { down = k; break; }
        int first = 6, second = 6, larger = 0;
        if (d < 6 && ((alive >> d) & 1u)) first = d;
        else if (up >= 0) { first = up; larger = 1; if (down >= 0) second = down; }
        else if (down >= 0) first = down;
        T->sel[e] = (uint16_t)(first | (second << 8) | (larger << 15));
        unsigned m = (unsigned)e >> 3;
        for (int i = 0; i < (e & 7); i++) m &= m - 1;
        int j = 0;
        while (m && !((m >> j) & 1u)) j++;
        T->nth[e] = (uint8_t)(m ? j : 0);
    }
    for (int q = 0; q < 64; q++) T->real_of_ring[q] = q < S * S ? (uint8_t)(S * S - 1 - rm_of_ring[q]) : 0;
    for (int q = 0; q < 64; q++) T->rot[q] = q < S * S ? (uint8_t)ring_of_rm[S * S - 1 - rm_of_ring[q]] : 0;
    for (int q = 0; q < 64; q++) { const int c = q < S * S ? rm_of_ring[q] : 0; T->dtl[q] = (uint8_t)((c / S > c % S) ? c / S : c % S); }
    T->init_posP = T->init_posN = 0x4040ull << 48; // bytes 6 and 7: no such cube
    {
        int cnt = 1;
        for (int i = 1; i <= 3; i++)
            for (int j = 0; j < i; j++) {
                const int cpos = j * S + (i - j - 1), cneg = (S - 1 - j) * S + (S - i + j); // real cells of +cnt / -cnt
                const int rN = ring_of_rm[S * S - 1 - cpos], rP = ring_of_rm[S * S - 1 - cneg];
                T->init_N |= 1ull << rN; T->init_posN |= (uint64_t)rN << (8 * (cnt - 1));
                T->init_P |= 1ull << rP; T->init_posP |= (uint64_t)rP << (8 * (cnt - 1));
                cnt++;
            }
    }
    const int W = S <= 5 ? 32 : 64;
    for (int z = 0; z <= W; z++) {
        const int h = W - 1 - z; // highest set bit
        const bool none = z == W || h >= S * S;
        const int t = none ? 0 : level_of_ring[h];
        T->lvl[z] = heur == 2 ? (uint8_t)(none ? 0 : S - 1 - t) : (uint8_t)(t << 3); // 'two_min_dist': the distance itself
    }
    // leaf values: score = 0; score += (L - mdP) * (1 / nP); score -= (L - mdN) * (1 / nN)   (envs/minimax_ewn.py:79-82)
    // with L = S, md = S - 1 - t.  volatile keeps every rounding step a separate fp64 operation.
    std::vector<double> all;
    std::vector<double> e((size_t)IXN * IXN, 0.0);
    std::vector<char> used_e((size_t)IXN * IXN, 0);
    if (heur == 2) { // index = sum of the two smallest distances of a side, 1 .. 2(S-1); value = sum(BOTTOM_RIGHT) - sum(TOP_LEFT), :176
        for (int sp = 1; sp <= 2 * (S - 1); sp++)
            for (int sn = 0; sn <= 2 * (S - 1); sn++) { // 0: BOTTOM_RIGHT's only cube still sits on the far corner (both sides are measured to it)
                double ev = (double)(sn - sp);
                if (variant == 1) { volatile double sixth = ev / 6.0, acc = 0.0; for (int d = 0; d < 6; d++) acc = acc + sixth; ev = acc; }
                e[(size_t)sp * IXN + sn] = ev; used_e[(size_t)sp * IXN + sn] = 1;
                all.push_back(ev);
            }
    } else
    for (int tp = 0; tp < S; tp++) for (int np_ = 1; np_ <= 6; np_++)
        for (int tn = 0; tn < S; tn++) for (int nn = 1; nn <= 6; nn++) {
            volatile double rp = 1.0 / (double)np_, rn_ = 1.0 / (double)nn;
            volatile double x = (double)(tp + 1) * rp, y = (double)(tn + 1) * rn_;
            volatile double s0 = 0.0 + x;
            volatile double s1 = s0 - y;
            double ev = s1;
            // the integer heuristics return Python ints; every later use (val / 6, comparisons, the returned root value) sees the
            // same number as a double.  min_dist: (L - mdP) - (L - mdN) with L - md = t + 1 (:126-127); attk: -(nP + nN) (:212)
            if (heur == 1) ev = (double)((tp + 1) - (tn + 1));
            if (heur == 3) ev = (double)(-(np_ + nn));
            if (variant == 1) { volatile double sixth = ev / 6.0, acc = 0.0; for (int d = 0; d < 6; d++) acc = acc + sixth; ev = acc; }
            // level S-1 is the far corner alone: TOP_LEFT standing there has won, evaluate() = +10 whatever else is on the board
            // (envs/minimax_ewn.py:41-44; no search reads these rows for anything else -- a root or a leaf that wins)
            if (tp == S - 1) ev = 10.0;
            e[(size_t)(tp * 8 + np_) * IXN + (tn * 8 + nn)] = ev; used_e[(size_t)(tp * 8 + np_) * IXN + (tn * 8 + nn)] = 1;
            all.push_back(ev);
        }
    all.push_back(-10.0);
    all.push_back(10.0);
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    if ((int)all.size() > FAST_NV - 3) return -1;
    T->nv = (int)all.size();
    T->m10 = (int)(1 + (std::lower_bound(all.begin(), all.end(), -10.0) - all.begin()));
    for (int i = 0; i < FAST_NV; i++) {
        const double v = i == 0 ? -__builtin_inf() : (i <= T->nv ? all[i - 1] : __builtin_inf()); // unused ranks compare above every alpha
        volatile double q = v / 6.0;
        T->val[i] = v; T->val6[i] = q;
    }
    // the rank of -10: 1 for 'hybrid' and 'min_dist' (nothing lies below it), not for 'attk' (-(6 + 6) = -12 does)
    const uint16_t m10 = (uint16_t)(1 + (std::lower_bound(all.begin(), all.end(), -10.0) - all.begin()));
    for (int ix = 0; ix < IXN; ix++)
        for (int iy = 0; iy < IXN; iy++) {
            const bool used = used_e[(size_t)ix * IXN + iy] != 0;
            // a slot no position maps to (count 0 = the side has lost its last cube, envs/minimax_ewn.py:45-47) answers -10
            // ... and column 0 of the other rows (BOTTOM_RIGHT has no cube left: TOP_LEFT has won, :41-44) answers +10: the leaves of
            // the max_depth 5 / 6 search (ewn_search_d5.hpp) that take the last cube land there
            const uint16_t unused_rank = (heur != 2 && iy == 0 && ix != 0) ? (uint16_t)T->nv : m10;
            T->rank[ix * IXN + iy] = (uint16_t)(8 * (used ? (uint16_t)(1 + (std::lower_bound(all.begin(), all.end(), e[(size_t)ix * IXN + iy]) - all.begin())) : unused_rank));
        }
    T->rank[0] = (uint16_t)FAST_NONE8; T->rank[1] = (uint16_t)(8 * m10); // unused (level 0, count 0) slots that d3_search's leaf index is steered to: "no such reply", -10
    return 0;
}
