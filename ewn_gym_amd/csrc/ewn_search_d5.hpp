// ewn_search_d5.hpp -- ExpectiMinimaxAgent.predict at max_depth 5 (and, on the second table variant, 6) in CLOSED FORM:
// no recursion, no break, no alpha-beta window -- every loop has a fixed trip count and the lanes of a wave never diverge.
//
// What classical_policies/minimax.py:19-73 computes at max_depth 5 (searcher = canonical TOP_LEFT = P, replier = N):
//   root max over P's moves (alpha = best root so far)
//     chance: v = sum_{d1} M3(d1) / 6
//       M3 = min over N's replies to dice d1 (list order: the dice cube, or its larger neighbour F then its smaller neighbour G,
//            envs/ewn.py:338-375), scan stopped once worst <= alpha; beta = worst so far is handed down
//         chance: C2 = sum_{d2} X1(d2) / 6
//           X1 = max over P's moves for dice d2, scan stopped once best >= beta  (alpha < beta always holds down here)
//             leaf = evaluate()
// and how it collapses (derivation pinned on the CPU by tests/test_d5_closed_form_model.py against the oracle's recursion):
//  * the value of a reply depends on the scan in front of it only through beta.  A cube's three replies AS F start from
//    beta = +inf, so they are the same for every dice that selects the cube as F.  AS G the chain continues from its F partner's
//    minimum, and a cube's F partner is always the next cube above it that is on the board -- so that chain, too, exists once.
//    Per root: <= 6 cubes x (3 + 3) reply evaluations instead of 6 dice x 6 replies;
//  * the leaves under a reply are <= 6 cubes x 3 directions = 18 distinct positions whatever d2 is; per cube the three prefix
//    maxima p0 <= p1 <= p2 of its leaves are kept (table RANKS: ewn_fast.hpp).  X1 for a dice that selects (F', G') is
//    cutF' ? cutF' : cutG' ? cutG' : max(p2F', p2G')  with cut = the first prefix maximum >= beta;
//  * a direction that leaves the board is replaced by one of the cube's directions that does not: repeating a leaf changes
//    neither the prefix maxima that matter nor the value at which the scan stops, and no per-leaf "exists" flag is needed;
//  * "P reaches the far corner" and "N has no cube left" (leaf value +10) are rows / a column of the rank table.
// Work per root and game: <= 18 x 18 leaves and <= 36 reply evaluations, against 6^4 leaves visited with data-dependent breaks
// (d5_search, ewn_step_d3.hpp, kept for the 'two_min_dist' image).  T = 1 or 2 lanes per game: lane `sub` owns the replier's
// cubes sub, sub + T, ... (walked from the highest down, so that a cube's F partner is finished before its G chain starts).
#pragma once

#define D5_CUT 0x8000u    // key flag: this cube's scan stops inside its leaves, the low bits hold the value it stops at
#define D5_CUTF 0x10000u  // the same, on the copy that travels as "nearest cube above" (it has to win the max against a G cut)

// One reply's chance node: sum over the six dice of the depth-1 max node's value / 6, in dice order (minimax.py:66-72).
// pw[k][i] = i-th prefix maximum (byte-offset rank) of cube k's leaves, 0 for a cube that is not on the board.
// NOCUT: beta = +inf (a cube's first reply as F): nothing stops early.
template <int S, bool NOCUT>
__device__ __forceinline__ double d5c_reply_value(const FastTab<S> *Tb, const u32 (&pw)[6][3], double beta)
{
    u32 key[6];
    if constexpr (NOCUT) {
        #pragma unroll
        for (int k = 0; k < 6; k++) key[k] = pw[k][2];
    } else {
        #pragma unroll
        for (int h = 0; h < 6; h += 3) {               // nine value reads in flight at a time
            double v[3][3];
            #pragma unroll
            for (int k = 0; k < 3; k++) {
                #pragma unroll
                for (int i = 0; i < 3; i++) v[k][i] = ft_val<S>(Tb, pw[h + k][i]); // val[0] = -inf: a missing cube never stops the scan
            }
            D3_STAGE_FENCE();
            #pragma unroll
            for (int k = 0; k < 3; k++) {
                // the first prefix maximum >= beta (three independent selects, the last one wins)
                u32 q = pw[h + k][2];
                q = v[k][2] >= beta ? (pw[h + k][2] | D5_CUT) : q;
                q = v[k][1] >= beta ? (pw[h + k][1] | D5_CUT) : q;
                q = v[k][0] >= beta ? (pw[h + k][0] | D5_CUT) : q;
                key[h + k] = q;
            }
        }
    }
    // which cubes a dice value selects (find_near_cube): the nearest cube on the board above / below, carried along
    u32 up[6], down[6];
    {
        u32 cur = 0;
        #pragma unroll
        for (int d = 5; d >= 0; d--) {
            up[d] = cur;
            const u32 kf = NOCUT ? key[d] : (key[d] | ((key[d] & D5_CUT) << 1));
            cur = key[d] != 0 ? kf : cur;
        }
        cur = 0;
        #pragma unroll
        for (int d = 0; d < 6; d++) { down[d] = cur; cur = key[d] != 0 ? key[d] : cur; }
    }
    double q6[6];
    #pragma unroll
    for (int d = 0; d < 6; d++) {
        // the dice cube alone if it is on the board; else F = up, G = down: F's cut value if F cuts, else G's if G cuts, else the
        // larger of the two maxima -- one max, because a cut F outranks everything and a cut G outranks an uncut F
        const u32 t = max(up[d], down[d]);
        const u32 x = (key[d] != 0 ? key[d] : t) & 0x7FFFu;
        q6[d] = ft_val6<S>(Tb, x);
    }
    D3_STAGE_FENCE();
    double val = 0.0;
    #pragma unroll
    for (int d = 0; d < 6; d++) val = val + q6[d];
    return val;
}

// bit k = P cube k is on the board after whatever stood on ring cell q (< 64) has been taken
EWN_DEV u32 d5c_alive_after(u64 posP, u32 q)
{
    const u32 q7 = q ^ 0x7Fu, qb = __builtin_amdgcn_perm(q7, q7, 0u);
    return pk_alive(pk_capture(posP, qb));
}

// PERLANE: one call = the three roots of one root cube (slotL), best / bflag / bdir carried in and out -- see d3_search.
template <int S, int T, bool PERLANE = false>
__device__ __forceinline__ double d5c_search(const FastTab<S> *Tb, const RState<S> &c, int dice, int sub, int &bflag, int &bdir,
                                             int slotL = 0, double best_in = 0.0, bool *have_second = nullptr)
{
    static_assert(T == 1 || T == 2, "one or two lanes per game");
    typedef typename MaskOf<S>::type M;
    constexpr int KPT = 6 / T;
    const M one = 1;
    const double inf = __builtin_inf();
    double best = -inf;
    const u32 e0 = pk_sel<S>(Tb, c.posP, dice), pp0 = pk_pair(c.posP, e0);
    if constexpr (PERLANE) {
        *have_second = !((pp0 >> 8) & PK_OFF);
        if (slotL == 0) { bflag = 0; bdir = 0; } else best = best_in;
    } else { bflag = 0; bdir = 0; }

    #pragma unroll 1
    for (int r = 0; r < (PERLANE ? 3 : 6); r++) {
        const int slot = PERLANE ? slotL : (r >= 3 ? 1 : 0), dir = PERLANE ? r : r - 3 * slot;
        const int cube = (int)((slot ? e0 >> 8 : e0) & 7u), rb = (int)((slot ? pp0 >> 8 : pp0) & 0xFFu);
        const int dest = Tb->nbp[dir][rb];
        const bool valid = dest != 255;                // no such cube (byte 6) or off the board
        if (__builtin_amdgcn_ballot_w64(valid) == 0) continue; // nobody in the wave has this root (uniform: the lanes stay together)
        RState<S> s1 = c;
        if (valid) rs_move<S, true>(s1, cube, dest);
        const bool term = dest == FastTab<S>::CELLS - 1 || s1.N == 0;
        const double alpha = best;

        // the searcher's 18 leaf moves from s1: they are the same under every reply (a reply only removes a cube)
        M setL[6][3], clrL[6];
        #pragma unroll
        for (int k = 0; k < 6; k++) {
            const int b = pk_get(s1.posP, k);
            const int d0 = Tb->nbp[0][b], d1 = Tb->nbp[1][b], d2 = Tb->nbp[2][b];
            const bool on = !(b & PK_OFF);
            const int q0 = d0 != 255 ? d0 : d1, q1 = d1 != 255 ? d1 : d0, q2 = d2 != 255 ? d2 : q0;
            setL[k][0] = on ? one << (q0 & 63) : (M)0;
            setL[k][1] = on ? one << (q1 & 63) : (M)0;
            setL[k][2] = on ? one << (q2 & 63) : (M)0;
            clrL[k] = on ? ~(one << (b & 63)) : ~(M)0;
        }

        // per cube of mine (slot i): A = the min node's result with the cube as F (cut value or minimum), B = with it as G
        double Ares[KPT], Bres[KPT];
        u32 cutbits = 0, alivebits = 0;               // bit i: my cube i cuts as F / is on the board
        #pragma unroll
        for (int i = 0; i < KPT; i++) { Ares[i] = inf; Bres[i] = inf; }
        // nearest cube on the board above the one(s) being worked on: its minimum as F, whether it cut, whether there is one
        double cMin = inf;
        bool cCut = false, cEx = false, aboveDead = false; // aboveDead: the cube directly above lane T-1's current cube is off the board

        #pragma unroll 1
        for (int i = 0; i < KPT; i++) {
            const int k = (6 - T + sub) - T * i;       // T = 2: lane 1 walks 5, 3, 1, lane 0 walks 4, 2, 0
            const int nb = pk_get(s1.posN, k);
            const bool on = !(nb & PK_OFF);
            const M rclr = ~(one << (nb & 63));
            M rset[3];
            bool ex[3], tm[3];
            u32 al2[3];
            u32 pw[3][6][3];
            #pragma unroll
            for (int j = 0; j < 3; j++) {
                const int dn = Tb->nbn[j][nb];          // 255: off the board, or no such cube
                ex[j] = dn != 255;
                rset[j] = ex[j] ? one << (dn & 63) : (M)0;
                const M P2 = s1.P & ~rset[j];
                tm[j] = dn == Tb->ri_origin || P2 == 0;  // check_win after the reply: evaluate() = -10
                al2[j] = d5c_alive_after(s1.posP, (u32)dn & 63u);
            }
            #pragma unroll
            for (int j = 0; j < 3; j++) {
                const M N2 = (s1.N & rclr) | rset[j], P2 = s1.P & ~rset[j];
                #pragma unroll
                for (int h = 0; h < 6; h += 3) {        // nine leaves at a time: levels, then ranks
                    D3_PRIO_HI();
                    M P3[3][3], N3[3][3];
                    u32 lp[3][3], ln[3][3], a[3][3];
                    #pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        #pragma unroll
                        for (int d = 0; d < 3; d++) {
                            P3[kk][d] = (P2 & clrL[h + kk]) | setL[h + kk][d];
                            N3[kk][d] = N2 & ~setL[h + kk][d];
                            lp[kk][d] = Tb->lvl[lvl_index(P3[kk][d])];
                            ln[kk][d] = Tb->lvl[lvl_index(N3[kk][d])];
                        }
                    }
                    D3_STAGE_FENCE();
                    #pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        #pragma unroll
                        for (int d = 0; d < 3; d++)
                            a[kk][d] = ft_rank8<S>(Tb, ft_addr(lp[kk][d] + (u32)popc_m(P3[kk][d]), ln[kk][d] + (u32)popc_m(N3[kk][d])));
                    }
                    D3_STAGE_FENCE();
                    D3_PRIO_LO();
                    #pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        const u32 m = 0u - ((al2[j] >> (h + kk)) & 1u); // a cube the reply took (or that was gone before) has no leaves
                        const u32 p0 = a[kk][0], p1 = max(p0, a[kk][1]), p2 = max(p1, a[kk][2]);
                        pw[j][h + kk][0] = p0 & m; pw[j][h + kk][1] = p1 & m; pw[j][h + kk][2] = p2 & m;
                    }
                }
            }
            // the cube as F: beta starts at +inf
            double w0, w1, w2;
            {
                double v = d5c_reply_value<S, true>(Tb, pw[0], inf);
                v = tm[0] ? -10.0 : v; w0 = ex[0] ? v : inf;
                v = d5c_reply_value<S, false>(Tb, pw[1], w0);
                v = tm[1] ? -10.0 : v; v = ex[1] ? v : inf; w1 = v < w0 ? v : w0;
                v = d5c_reply_value<S, false>(Tb, pw[2], w1);
                v = tm[2] ? -10.0 : v; v = ex[2] ? v : inf; w2 = v < w1 ? v : w1;
            }
            // the running minimum never rises: the scan stops at the first w <= alpha (minimax.py:59-61)
            double A = w2;
            A = w1 <= alpha ? w1 : A;
            A = w0 <= alpha ? w0 : A;
            const bool cutF = on && w2 <= alpha;
            // who is the nearest cube above mine?  T = 2: for lane 0 it is lane 1's cube of this step if that is on the board
            double uMin = cMin; bool uCut = cCut, uEx = cEx, upDead = aboveDead;
            double hiMin = w2; bool hiCut = cutF, hiOn = on;       // lane T-1's cube of this step
            double loMin = w2; bool loCut = cutF, loOn = on;       // lane 0's
            if constexpr (T == 2) {
                const u32 fl = (on ? 1u : 0u) | (cutF ? 2u : 0u);
                const u32 f0 = dpp_u32<Bcast<2, 0>::CTRL>(fl), f1 = dpp_u32<Bcast<2, 1>::CTRL>(fl);
                loMin = dpp_f64<Bcast<2, 0>::CTRL>(w2); hiMin = dpp_f64<Bcast<2, 1>::CTRL>(w2);
                loOn = f0 & 1u; loCut = (f0 >> 1) & 1u; hiOn = f1 & 1u; hiCut = (f1 >> 1) & 1u;
                if (sub == 0) { uMin = hiOn ? hiMin : cMin; uCut = hiOn ? hiCut : cCut; uEx = hiOn || cEx; upDead = !hiOn; }
            }
            // the cube as G: only a dice whose own cube is gone selects a pair, so the cube directly above must be off the board,
            // and the F partner must not have cut (else the scan never reaches G)
            const bool needG = on && upDead && uEx && !uCut;
            double B = inf;
            if (__builtin_amdgcn_ballot_w64(needG) != 0) {
                double v = d5c_reply_value<S, false>(Tb, pw[0], uMin);
                v = tm[0] ? -10.0 : v; v = ex[0] ? v : inf; const double g0 = v < uMin ? v : uMin;
                v = d5c_reply_value<S, false>(Tb, pw[1], g0);
                v = tm[1] ? -10.0 : v; v = ex[1] ? v : inf; const double g1 = v < g0 ? v : g0;
                v = d5c_reply_value<S, false>(Tb, pw[2], g1);
                v = tm[2] ? -10.0 : v; v = ex[2] ? v : inf; const double g2 = v < g1 ? v : g1;
                B = g2;
                B = g1 <= alpha ? g1 : B;
                B = g0 <= alpha ? g0 : B;
            }
            #pragma unroll
            for (int n = 0; n < KPT; n++) { Ares[n] = i == n ? A : Ares[n]; Bres[n] = i == n ? B : Bres[n]; }
            cutbits |= (cutF ? 1u : 0u) << i;
            alivebits |= (on ? 1u : 0u) << i;
            // carry for the next step: the lowest cube on the board so far
            cMin = loOn ? loMin : (hiOn ? hiMin : cMin);
            cCut = loOn ? loCut : (hiOn ? hiCut : cCut);
            cEx = loOn || hiOn || cEx;
            aboveDead = !loOn;
        }

        // all six cubes' results in every lane, indexed by cube number (slot i of lane j is cube 6 - T + j - T * i)
        double A6[6], B6[6];
        u32 on6 = 0, cut6 = 0;
        #pragma unroll
        for (int i = 0; i < KPT; i++) {
            if constexpr (T == 1) {
                A6[5 - i] = Ares[i]; B6[5 - i] = Bres[i];
                on6 |= ((alivebits >> i) & 1u) << (5 - i); cut6 |= ((cutbits >> i) & 1u) << (5 - i);
            } else {
                A6[4 - 2 * i] = dpp_f64<Bcast<2, 0>::CTRL>(Ares[i]); A6[5 - 2 * i] = dpp_f64<Bcast<2, 1>::CTRL>(Ares[i]);
                B6[4 - 2 * i] = dpp_f64<Bcast<2, 0>::CTRL>(Bres[i]); B6[5 - 2 * i] = dpp_f64<Bcast<2, 1>::CTRL>(Bres[i]);
            }
        }
        if constexpr (T == 2) {
            const u32 a0 = dpp_u32<Bcast<2, 0>::CTRL>(alivebits), a1 = dpp_u32<Bcast<2, 1>::CTRL>(alivebits);
            const u32 c0 = dpp_u32<Bcast<2, 0>::CTRL>(cutbits), c1 = dpp_u32<Bcast<2, 1>::CTRL>(cutbits);
            #pragma unroll
            for (int i = 0; i < KPT; i++) {
                on6 |= (((a0 >> i) & 1u) << (4 - 2 * i)) | (((a1 >> i) & 1u) << (5 - 2 * i));
                cut6 |= (((c0 >> i) & 1u) << (4 - 2 * i)) | (((c1 >> i) & 1u) << (5 - 2 * i));
            }
        }
        // the six min nodes of the root's chance node: the dice cube's A if it is on the board; else F = the nearest cube above,
        // G = the nearest below: A_F if F cut or there is no G, else B_G; no cube above: A of the one below
        double upA[6], downA[6], downB[6];
        bool upE[6], upC[6], downE[6];
        {
            double ca = inf; bool ce = false, cc = false;
            #pragma unroll
            for (int d = 5; d >= 0; d--) {
                upA[d] = ca; upE[d] = ce; upC[d] = cc;
                const bool o = (on6 >> d) & 1u;
                ca = o ? A6[d] : ca; cc = o ? (bool)((cut6 >> d) & 1u) : cc; ce = ce || o;
            }
            double da = inf, db = inf; bool de = false;
            #pragma unroll
            for (int d = 0; d < 6; d++) {
                downA[d] = da; downB[d] = db; downE[d] = de;
                const bool o = (on6 >> d) & 1u;
                da = o ? A6[d] : da; db = o ? B6[d] : db; de = de || o;
            }
        }
        double v = 0.0;
        #pragma unroll
        for (int d = 0; d < 6; d++) {
            const bool o = (on6 >> d) & 1u;
            const double pairv = (upC[d] || !downE[d]) ? upA[d] : downB[d];
            const double w = o ? A6[d] : (upE[d] ? pairv : downA[d]);
            v = v + w / 6.0;                           // expected_val += val / 6, minimax.py:72
        }
        v = term ? 10.0 : v;
        if (valid && v > best) { best = v; bflag = slot == 0 ? (int)(e0 >> 15) : 0; bdir = dir; }
    }
    D3_PRIO_LO();
    return best;
}
