// ewn_a2c3.hpp -- k_a2c_grad3: the A2C gradient pass (ewn_a2c.hpp: same inputs, same partial-gradient output) on the bf16 matrix
// pipe at fp32 accuracy (ewn_mlp3.hpp), with NO LDS transposes and no barriers inside the step loop.
//
// One wave owns a tile of 32 samples and the whole net's gradient (one wave per SIMD: the register file is the gradient buffer).
// Two register layouts of a [64 units] x [32 samples] quantity occur:
//   S-layout  sample on the lane, units in the registers -- what the forward chain produces and what a product that sums over UNITS
//             wants as an operand (layer 2, dh1 = W2^T g2);
//   U-layout  unit on the lane, samples in the registers -- what a product that sums over SAMPLES wants (dW = g a^T).
// k_a2c_grad / k_a2c_grad2 went from one to the other through LDS ([sample][unit] tiles, written and re-read every step, the two
// waves of a team meeting four times per step).  Here the matrix pipe does it: an S-layout operand times an identity operand IS the
// U-layout tile (D[sample][unit] = sum_k X[sample][k] I[k][unit], three exact products for the three bf16 parts, exact fp32 sum), six
// 32-cycle MFMAs per 32 units on a pipe that otherwise idles, and eight consecutive registers of the result are again a k-block --
// of samples this time.  dh1 is computed directly in U-layout (operands swapped), so g1 never exists in S-layout at all.
//
// Per tile and step: ~270 MFMAs (8.6 k matrix cycles, overlapped) and ~2.5 k VALU instructions, most of them the operand splits.
#pragma once
#include "ewn_a2c.hpp"
#include "ewn_mlp3.hpp"

template <int S> struct A2c3Geo {
    using G = MlpGeo<S>;
    using Q = Mlp3Geo<S>;
    static constexpr int FT = (G::F + 31) / 32;                    // 32-wide feature tiles of dW1
    static constexpr int N_W2T = 2 * 4 * 64, N_WH = 2 * 64, N_WF = 4 * 64;     // entries per part
    static constexpr int O_W2T = Q::FWD_BYTES, O_WH = O_W2T + 3 * N_W2T * 16, O_WF = O_WH + 3 * N_WH * 16, O_GI = O_WF + 3 * N_WF * 16;
    static constexpr int NET_PARAMS = G::BODY + MLP_NA * MLP_H + MLP_NA;
    static constexpr size_t lds_bytes() { return (size_t)O_GI + ((size_t)NET_PARAMS + 8) * 4; }
    static_assert(Q::FWD_BYTES % 16 == 0, "image alignment");
};

#define A2C3_FENCE() __builtin_amdgcn_sched_barrier(0)
// -DA2C3_STAMPS: shader-clock time per region of the step (block 0, wave 0), left in the 64 spare floats behind the scratch's stats
// (diagnostic build only: every stamp waits for the LDS queue)
#ifdef A2C3_STAMPS
#define A2C3_T(i) do { const u64 now_ = __builtin_amdgcn_s_memtime(); tacc[i] += (u32)(now_ - tlast); tlast = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A2C3_T(i) do { } while (0)
#endif

// identity operands: slot jj of lane (n, h) is 1.0 where the slot's k is column n
EWN_DEV u32x4 a2c3_identity(int lane, bool unit_slots, int c)
{
    const int n = lane & 31, h = lane >> 5;
    u32x4 o = { 0u, 0u, 0u, 0u };
    #pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        const int k = unit_slots ? mlp_row(8 * c + jj, h) : 16 * c + 8 * h + jj;   // unit held in that slot / feature (or head row) of it
        if (k == n) o[jj >> 1] |= 0x3F80u << (16 * (jj & 1));
    }
    return o;
}

// S-layout k-block operand (as A: rows = samples) -> U-layout tile accumulator: exact
EWN_DEV f32x16 a2c3_transpose_add(f32x16 acc, const Mlp3Op &x, u32x4 id)
{
    acc = MLP3_MFMA(x.p[2], id, acc); acc = MLP3_MFMA(x.p[1], id, acc); acc = MLP3_MFMA(x.p[0], id, acc);
    return acc;
}

// what a step reads from global memory (loaded one step ahead: a lone wave per SIMD has nothing to hide the latency under)
template <int S> struct A2c3Ld { uint2 xb[Mlp3Geo<S>::KB1]; int dice; A2cStepIn in; };

template <int S, int NET>
EWN_DEV A2c3Ld<S> a2c3_load(const A2cCfg &c, const A2cBuf &B, int t, int gc, int h)
{
    constexpr int CELLS = S * S, STR = RecGeo<S>::STR;
    A2c3Ld<S> L;
    const uint8_t *rrow = B.rec + ((size_t)t * c.N + gc) * STR;
    #pragma unroll
    for (int kb = 0; kb < Mlp3Geo<S>::KB1; kb++) L.xb[kb] = *(const uint2 *)(rrow + 16 * kb + 8 * h);
    L.dice = (int8_t)rrow[CELLS];
    if (t < c.K) L.in = a2c_step_in<NET>(c, B, rrow + (size_t)c.N * STR, CELLS, t, gc);   // row t + 1: action a_t, flags of step t
    else L.in = A2cStepIn{ 0, 0, false, 0.0f, 0.0f };
    return L;
}

// features: record bytes 16 kb + 8 h .. + 7 of my sample (bytes past the board masked off), the dice one-hot
template <int S>
EWN_DEV void a2c3_features(const A2c3Ld<S> &L, int h, u32x4 (&xop)[Mlp3Geo<S>::KB1])
{
    constexpr int CELLS = S * S;
    #pragma unroll
    for (int kb = 0; kb < Mlp3Geo<S>::KB1; kb++) {
        u32 lo = L.xb[kb].x, hi = L.xb[kb].y;
        if (16 * kb + 15 >= CELLS) {
            const int nv = CELLS - 16 * kb - 8 * h;                      // valid bytes of my eight
            const unsigned long long m = nv >= 8 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << (8 * nv)) - 1ull));
            lo &= (u32)m; hi &= (u32)(m >> 32);
        }
        xop[kb] = mlp3_bytes_operand(lo, hi);
        if (16 * kb + 15 >= CELLS && 16 * kb < CELLS + 7) xop[kb] = mlp3_onehot(xop[kb], CELLS + L.dice - 1 - (16 * kb + 8 * h));
    }
}

// NET 0: policy body + action head; NET 1: value body + value head.  256 threads: four waves, one per SIMD.
template <int S, int NET>
__global__ __launch_bounds__(256, 1) void k_a2c_grad3(A2cCfg c, A2cBuf B)
{
    using G = MlpGeo<S>;
    using Q = Mlp3Geo<S>;
    using A = A2c3Geo<S>;
    constexpr int CELLS = S * S, NT = 256, NWV = 4, NOUT = NET ? 1 : MLP_NA, KB1 = Q::KB1, FT = A::FT;
    extern __shared__ __attribute__((aligned(16))) int8_t lds3[];
    int8_t *img = lds3;
    const u32x4 *I1 = (const u32x4 *)(img + Q::O_W1), *I2 = (const u32x4 *)(img + Q::O_W2);
    u32x4 *IW2T = (u32x4 *)(img + A::O_W2T), *IWH = (u32x4 *)(img + A::O_WH), *IWF = (u32x4 *)(img + A::O_WF);
    const float *Lf = (const float *)(img + Q::O_F);
    float *GI = (float *)(img + A::O_GI);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;

    mlp3_pack_fwd<S>(img, B.params, NET, threadIdx.x, NT);
    {
        const float *W2 = B.params + (NET ? G::O_VF : G::O_PI) + MLP_H * G::F + MLP_H;
        const float *Wh = B.params + (NET ? G::O_VW : G::O_AW);
        for (int e = threadIdx.x; e < A::N_W2T; e += NT) {        // [col tile][k-block][lane]: W2[unit(kb, h, jj)][32 nt + (lane & 31)]
            const int l = e & 63, kb = (e >> 6) & 3, nt = e >> 8;
            float v[8];
            #pragma unroll
            for (int jj = 0; jj < 8; jj++) v[jj] = W2[mlp3_unit(kb, l >> 5, jj) * MLP_H + 32 * nt + (l & 31)];
            mlp3_store(IW2T, A::N_W2T, e, v);
        }
        for (int e = threadIdx.x; e < A::N_WH; e += NT) {         // [unit tile][lane]: Wh[8 h + jj][32 mt + (lane & 31)] (rows past the head's: zero)
            const int l = e & 63, mt = e >> 6;
            float v[8];
            #pragma unroll
            for (int jj = 0; jj < 8; jj++) { const int a = 8 * (l >> 5) + jj; v[jj] = a < NOUT ? Wh[a * MLP_H + 32 * mt + (l & 31)] : 0.0f; }
            mlp3_store(IWH, A::N_WH, e, v);
        }
        for (int e = threadIdx.x; e < A::N_WF; e += NT) {         // [k-block][lane]: Wh[lane & 31][unit(kb, h, jj)] (the head as an MFMA row tile)
            const int l = e & 63, kb = e >> 6;
            float v[8];
            #pragma unroll
            for (int jj = 0; jj < 8; jj++) v[jj] = (l & 31) < NOUT ? Wh[(l & 31) * MLP_H + mlp3_unit(kb, l >> 5, jj)] : 0.0f;
            mlp3_store(IWF, A::N_WF, e, v);
        }
    }
    __syncthreads();

    const u32x4 idu[2] = { a2c3_identity(lane, true, 0), a2c3_identity(lane, true, 1) };     // unit slots of k-block parity c -> column n
    const u32x4 idf[2] = { a2c3_identity(lane, false, 0), a2c3_identity(lane, false, 1) };   // feature / head-row slots 16 c + 8 h + jj -> column n

    // gradient accumulators of everything this wave sees
    f32x16 dW2[2][2], dW1[2][FT], dWh[2];
    #pragma unroll
    for (int a = 0; a < 2; a++) {
        #pragma unroll
        for (int b = 0; b < 2; b++) dW2[a][b] = (f32x16)(0.0f);
        #pragma unroll
        for (int b = 0; b < FT; b++) dW1[a][b] = (f32x16)(0.0f);
        dWh[a] = (f32x16)(0.0f);
    }
    float db2a[2] = { 0.0f, 0.0f };                 // U-layout partials: unit 32 nt + n, my half's samples
    float dbh[MLP_NA] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    float st_pl = 0.0f, st_vl = 0.0f, st_en = 0.0f;

#ifdef A2C3_STAMPS
    u32 tacc[16] = { 0 };
    u64 tlast = __builtin_amdgcn_s_memtime();
#endif
    const int tiles = (c.N + 31) / 32;
    #pragma unroll 1
    for (int tile = (int)blockIdx.x * NWV + wave; tile < tiles; tile += (int)gridDim.x * NWV) {
        const int game = tile * 32 + j;
        const bool valid = game < c.N;
        const int gc = valid ? game : c.N - 1;
        float Rn = 0.0f;
        A2c3Ld<S> cur = a2c3_load<S, NET>(c, B, NET ? c.K : c.K - 1, gc, h);
        if constexpr (NET == 1) {       // the bootstrap value V(s_K): a forward pass of its own (the step loop below stays branch-free)
            const A2c3Ld<S> nxt = a2c3_load<S, NET>(c, B, c.K - 1, gc, h);
            u32x4 xop[KB1];
            a2c3_features<S>(cur, h, xop);
            f32x16 h1[2], h2[2];
            float out[1];
            mlp3_forward<S, 1>(img, lane, [&](int kb) { return xop[kb]; }, h1, h2, out);
            Rn = out[0];
            cur = nxt;
            A2C3_FENCE();
        }
        // t = K-1 .. 0: forward + backward of step t
        #pragma unroll 1
        for (int t = c.K - 1; t >= 0; t--) {
            const A2c3Ld<S> nxt = a2c3_load<S, NET>(c, B, t > 0 ? t - 1 : 0, gc, h);
            // ---- layer 1's weight operands are asked for first: the feature decode below covers their LDS latency
            Mlp3Op w1[2][KB1];
            #pragma unroll
            for (int kb = 0; kb < KB1; kb++) { w1[0][kb] = mlp3_load(I1, Q::N_W1, kb * 64 + lane); w1[1][kb] = mlp3_load(I1, Q::N_W1, (KB1 + kb) * 64 + lane); }
            u32x4 xop[KB1];
            a2c3_features<S>(cur, h, xop);
            A2C3_FENCE();
            A2C3_T(0);
            // The step is a chain of fenced regions (A2C3_FENCE = sched_barrier).  Two reasons.  Registers: a lone wave per SIMD has 256
            // architectural registers for everything the VALU touches, and an unfenced schedule hoists every operand load and split to
            // the top (measured: 300 registers spilled to scratch, 27 us per tile and step).  Overlap: a wave issues in order, so its
            // MFMAs run under its own VALU work only when the two alternate IN PROGRAM ORDER -- each region therefore pairs the MFMAs
            // of one k-block with the operand split of the NEXT one (or of a later product), which do not depend on each other, and the
            // scheduler interleaves inside the region.  An operand lives for one k-block; results wait in the accumulation registers.
            // ---- layer 1 (layer 2's first weight operands are asked for under it)
            f32x16 h1[2], h2[2];
            Mlp3Op wa = mlp3_load(I2, Q::N_W2, lane), wb = mlp3_load(I2, Q::N_W2, 4 * 64 + lane);
            {
                f32x16 a0 = mlp_bias_acc(Lf + Q::F_B1, h), a1 = mlp_bias_acc(Lf + Q::F_B1 + 32, h);
                #pragma unroll
                for (int kb = 0; kb < KB1; kb++) { a0 = mlp3_mac_ax(a0, w1[0][kb], xop[kb]); a1 = mlp3_mac_ax(a1, w1[1][kb], xop[kb]); }
                h1[0] = mlp_tanh16(a0); h1[1] = mlp_tanh16(a1);
            }
            // ---- layer 2, and h1 in U-layout for the backward pass (the same operand, times the identity)
            f32x16 h1U[2] = { (f32x16)(0.0f), (f32x16)(0.0f) };
            Mlp3Op wf = wa;                                // the MFMA head's / dh2's first weight operand, asked for a region ahead
            {
                f32x16 c0 = mlp_bias_acc(Lf + Q::F_B2, h), c1 = mlp_bias_acc(Lf + Q::F_B2 + 32, h);
                Mlp3Op u = mlp3_operand(h1[0], 0);
                A2C3_FENCE();
                A2C3_T(1);
                #pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    Mlp3Op un = u, wan = wa, wbn = wb;
                    if (kb + 1 < 4) { wan = mlp3_load(I2, Q::N_W2, (kb + 1) * 64 + lane); wbn = mlp3_load(I2, Q::N_W2, (4 + kb + 1) * 64 + lane); }
                    else if (NET == 0) wf = mlp3_load(IWF, A::N_WF, lane);
                    if (kb + 1 < 4) un = mlp3_operand(h1[(kb + 1) >> 1], (kb + 1) & 1);
                    c0 = mlp3_mac(c0, wa, u);
                    c1 = mlp3_mac(c1, wb, u);
                    h1U[kb >> 1] = a2c3_transpose_add(h1U[kb >> 1], u, idu[kb & 1]);
                    A2C3_FENCE();
                    u = un; wa = wan; wb = wbn;
                }
                A2C3_T(2);
                h2[0] = mlp_tanh16(c0); h2[1] = mlp_tanh16(c1);
            }
            // ---- the head
            float out[NOUT];
            f32x16 h2U[2] = { (f32x16)(0.0f), (f32x16)(0.0f) };
            if constexpr (NET == 1) {
                mlp3_head<S, 1>(img, lane, h2, out);
                A2C3_T(3);
            } else {
                // the five logits as rows 0-4 of an MFMA tile (h2's operand split is needed for its U-layout anyway)
                f32x16 lg = (f32x16)(0.0f);
                Mlp3Op u = mlp3_operand(h2[0], 0);
                A2C3_FENCE();
                A2C3_T(3);
                #pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    Mlp3Op un = u, wfn = wf;
                    if (kb + 1 < 4) { wfn = mlp3_load(IWF, A::N_WF, (kb + 1) * 64 + lane); un = mlp3_operand(h2[(kb + 1) >> 1], (kb + 1) & 1); }
                    lg = mlp3_mac(lg, wf, u);
                    h2U[kb >> 1] = a2c3_transpose_add(h2U[kb >> 1], u, idu[kb & 1]);
                    A2C3_FENCE();
                    u = un; wf = wfn;
                }
                A2C3_T(4);
                // rows 0-3 sit in registers 0-3 of lane half 0, row 4 in register 0 of half 1
                const float o0 = mlp_other_half(lg[0], lane), o1 = mlp_other_half(lg[1], lane), o2 = mlp_other_half(lg[2], lane), o3 = mlp_other_half(lg[3], lane);
                out[0] = (h ? o0 : lg[0]) + Lf[Q::F_BH]; out[1] = (h ? o1 : lg[1]) + Lf[Q::F_BH + 1];
                out[2] = (h ? o2 : lg[2]) + Lf[Q::F_BH + 2]; out[3] = (h ? o3 : lg[3]) + Lf[Q::F_BH + 3];
                out[4] = (h ? lg[0] : o0) + Lf[Q::F_BH + 4];
            }
            A2C3_FENCE();
            A2C3_T(5);
            // ---- the loss of step t and its gradient w.r.t. the head outputs (both lane halves hold the same numbers); dh2's weight
            // operands are asked for under it
            const Mlp3Op wh0 = mlp3_load(IWH, A::N_WH, lane), wh1 = mlp3_load(IWH, A::N_WH, 64 + lane);
            float d[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
            a2c_loss_grad<NET>(c, B, cur.in, t, game, valid, h == 0, out, Rn, d, st_pl, st_vl, st_en);
            Mlp3Op dop;                              // d as a k-block: slot (h, jj) = head row 8 h + jj
            {
                float v[8];
                #pragma unroll
                for (int jj = 0; jj < 8; jj++) v[jj] = (jj < NOUT && h == 0) ? d[jj] : 0.0f;
                dop = mlp3_operand(v);
            }
            A2C3_FENCE();
            A2C3_T(6);
            // ---- dh2 = Wh^T d (S-layout) and d in U-layout on the matrix pipe, under them the first splits of h2 in U-layout (policy) /
            // the value head's gradient (value)
            f32x16 g2[2], dU = (f32x16)(0.0f);
            Mlp3Op h2k[2][2];                        // [unit tile][sample k-block]
            {
                g2[0] = mlp3_mac((f32x16)(0.0f), wh0, dop); g2[1] = mlp3_mac((f32x16)(0.0f), wh1, dop);
                if constexpr (NET == 0) {
                    dU = a2c3_transpose_add(dU, dop, idf[0]);
                    #pragma unroll
                    for (int nt = 0; nt < 2; nt++) { h2k[nt][0] = mlp3_operand(h2U[nt], 0); h2k[nt][1] = mlp3_operand(h2U[nt], 1); }
                    if (h == 0) { for (int i = 0; i < 5; i++) dbh[i] += d[i]; }
                } else {
                    #pragma unroll
                    for (int mt = 0; mt < 2; mt++) {
                        #pragma unroll
                        for (int r = 0; r < 16; r++) dWh[mt][r] += d[0] * h2[mt][r];     // per-lane partial of dWv[unit] = sum_s dV_s h2[unit][s]
                    }
                    if (h == 0) dbh[0] += d[0];
                }
            }
            A2C3_FENCE();
            A2C3_T(7);
            // g2 = dh2 (1 - h2^2)
            #pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                #pragma unroll
                for (int r = 0; r < 16; r++) g2[mt][r] *= 1.0f - h2[mt][r] * h2[mt][r];
            }
            Mlp3Op dk[2];
            if constexpr (NET == 0) { dk[0] = mlp3_operand(dU, 0); dk[1] = mlp3_operand(dU, 1); }
            Mlp3Op wt0 = mlp3_load(IW2T, A::N_W2T, lane), wt1 = mlp3_load(IW2T, A::N_W2T, 4 * 64 + lane);   // dh1's first weight operands
            A2C3_FENCE();
            A2C3_T(8);
            // ---- g2 in U-layout; dh1 = W2^T g2 computed in U-layout directly (operands swapped); the action head's gradient
            // dWa[a][unit] = sum_s d[a][s] h2[unit][s] (both operands in U-layout; d: head row on the lane); under them the splits of h1
            // in U-layout that dW2 wants
            f32x16 g2U[2] = { (f32x16)(0.0f), (f32x16)(0.0f) }, g1U[2] = { (f32x16)(0.0f), (f32x16)(0.0f) };
            Mlp3Op h1k[2][2];                        // [unit tile][sample k-block]
            {
                Mlp3Op u = mlp3_operand(g2[0], 0);
                A2C3_FENCE();
                A2C3_T(9);
                #pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    Mlp3Op un = u, wt0n = wt0, wt1n = wt1;
                    if (kb + 1 < 4) {
                        wt0n = mlp3_load(IW2T, A::N_W2T, (kb + 1) * 64 + lane); wt1n = mlp3_load(IW2T, A::N_W2T, (4 + kb + 1) * 64 + lane);
                        un = mlp3_operand(g2[(kb + 1) >> 1], (kb + 1) & 1);
                    }
                    h1k[kb >> 1][kb & 1] = mlp3_operand(h1U[kb >> 1], kb & 1);
                    g2U[kb >> 1] = a2c3_transpose_add(g2U[kb >> 1], u, idu[kb & 1]);
                    g1U[0] = mlp3_mac(g1U[0], u, wt0); g1U[1] = mlp3_mac(g1U[1], u, wt1);
                    if constexpr (NET == 0) dWh[kb >> 1] = mlp3_mac(dWh[kb >> 1], dk[kb & 1], h2k[kb >> 1][kb & 1]);
                    A2C3_FENCE();
                    u = un; wt0 = wt0n; wt1 = wt1n;
                }
                A2C3_T(10);
            }
            // g1 = dh1 (1 - h1^2); db2 = the sum of g2 over the samples
            #pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                float sb = 0.0f;
                #pragma unroll
                for (int r = 0; r < 16; r++) { g1U[nt][r] *= 1.0f - h1U[nt][r] * h1U[nt][r]; sb += g2U[nt][r]; }
                db2a[nt] += sb;
            }
            A2C3_FENCE();
            A2C3_T(11);
            // ---- dW2 += g2 . h1^T, then dW1 += g1 . x^T (the bias gradient db1 is the sum of its seven one-hot dice columns: taken
            // at the end); under dW2's MFMAs the splits of g1 and the features in U-layout
            {
                Mlp3Op ka[2] = { mlp3_operand(g2U[0], 0), mlp3_operand(g2U[1], 0) };
                Mlp3Op g1k[2][2];
                f32x16 xU[FT];
                A2C3_FENCE();
                A2C3_T(12);
                #pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    Mlp3Op kan[2] = { ka[0], ka[1] };
                    if (kb == 0) { kan[0] = mlp3_operand(g2U[0], 1); kan[1] = mlp3_operand(g2U[1], 1); }
                    g1k[0][kb] = mlp3_operand(g1U[0], kb); g1k[1][kb] = mlp3_operand(g1U[1], kb);
                    if (kb == 0) {
                        #pragma unroll
                        for (int ft = 0; ft < FT; ft++) {    // features 32 ft + n of the tile's samples (small integers: one bf16 part)
                            xU[ft] = (f32x16)(0.0f);
                            #pragma unroll
                            for (int cc = 0; cc < 2; cc++) { if (2 * ft + cc < KB1) xU[ft] = MLP3_MFMA(xop[2 * ft + cc], idf[cc], xU[ft]); }
                        }
                    }
                    #pragma unroll
                    for (int mt = 0; mt < 2; mt++) {
                        #pragma unroll
                        for (int nt = 0; nt < 2; nt++) dW2[mt][nt] = mlp3_mac(dW2[mt][nt], ka[mt], h1k[nt][kb]);
                    }
                    A2C3_FENCE();
                    ka[0] = kan[0]; ka[1] = kan[1];
                }
                A2C3_T(13);
                #pragma unroll
                for (int ft = 0; ft < FT; ft++) {
                    #pragma unroll
                    for (int kb = 0; kb < 2; kb++) {
                        u32x4 xk;
                        #pragma unroll
                        for (int q = 0; q < 4; q++) xk[q] = mlp3_pack(__float_as_uint(xU[ft][8 * kb + 2 * q]), __float_as_uint(xU[ft][8 * kb + 2 * q + 1]));
                        #pragma unroll
                        for (int mt = 0; mt < 2; mt++) dW1[mt][ft] = mlp3_mac_ax(dW1[mt][ft], g1k[mt][kb], xk);
                    }
                }
                A2C3_FENCE();
                A2C3_T(14);
            }
            cur = nxt;
        }
    }

    // ---- the block's gradient image in LDS: waves add in a fixed order (bit-reproducible), then one coalesced copy out
    constexpr int I_W1 = 0, I_B1 = I_W1 + MLP_H * G::F, I_W2 = I_B1 + MLP_H, I_B2 = I_W2 + MLP_H * MLP_H, I_WH = I_B2 + MLP_H,
                  I_BH = I_WH + NOUT * MLP_H, I_END = I_BH + NOUT;
    #pragma unroll
    for (int nt = 0; nt < 2; nt++) db2a[nt] += __shfl_xor(db2a[nt], 32, 64);     // the other half's samples
    if constexpr (NET == 1) {
        #pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            #pragma unroll
            for (int r = 0; r < 16; r++) dWh[mt][r] = a2c_sum32(dWh[mt][r]);
        }
    }
    #pragma unroll
    for (int i = 0; i < NOUT; i++) { dbh[i] = a2c_sum32(dbh[i]); dbh[i] += __shfl_xor(dbh[i], 32, 64); }
    st_pl = a2c_sum32(st_pl); st_vl = a2c_sum32(st_vl); st_en = a2c_sum32(st_en);
    #pragma unroll 1
    for (int w = 0; w < NWV; w++) {
        if (wave == w) {
            const bool first = w == 0;
            #pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                #pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = 32 * mt + mlp_row(r, h);
                    #pragma unroll
                    for (int nt = 0; nt < 2; nt++) { float *p = GI + I_W2 + row * MLP_H + 32 * nt + j; *p = (first ? 0.0f : *p) + dW2[mt][nt][r]; }
                    #pragma unroll
                    for (int ft = 0; ft < FT; ft++) {
                        const int col = 32 * ft + j;
                        if (col < G::F) { float *p = GI + I_W1 + row * G::F + col; *p = (first ? 0.0f : *p) + dW1[mt][ft][r]; }
                    }
                    if (NET == 1 && j == 0) { float *p = GI + I_WH + row; *p = (first ? 0.0f : *p) + dWh[mt][r]; }
                }
                if (h == 0) { float *p = GI + I_B2 + 32 * mt + j; *p = (first ? 0.0f : *p) + db2a[mt]; }
            }
            if constexpr (NET == 0) {    // dWa tile: row a = mlp_row(r, h) (a < 5), unit = 32 nt + j
                #pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    #pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int a = mlp_row(r, h);
                        if (a < MLP_NA) { float *p = GI + I_WH + a * MLP_H + 32 * nt + j; *p = (first ? 0.0f : *p) + dWh[nt][r]; }
                    }
                }
            }
            if (lane == 0) {
                #pragma unroll
                for (int i = 0; i < NOUT; i++) { float *p = GI + I_BH + i; *p = (first ? 0.0f : *p) + dbh[i]; }
                float *sp = GI + I_END;   // four floats of loss sums behind the image
                sp[0] = (first ? 0.0f : sp[0]) + st_pl; sp[1] = (first ? 0.0f : sp[1]) + st_vl; sp[2] = (first ? 0.0f : sp[2]) + st_en; sp[3] = 0.0f;
            }
        }
        __syncthreads();
    }
    // db1[row] = sum of dW1[row][CELLS .. CELLS + 6] (exactly one dice feature is 1 in every sample)
    for (int row = threadIdx.x; row < MLP_H; row += NT) {
        float sacc = 0.0f;
        for (int dd = 0; dd < 7; dd++) sacc += GI[I_W1 + row * G::F + CELLS + dd];
        GI[I_B1 + row] = sacc;
    }
    __syncthreads();
    float *dst = B.partial + (size_t)blockIdx.x * G::P;
    const int o_body = NET ? G::O_VF : G::O_PI, o_hw = NET ? G::O_VW : G::O_AW;
    for (int e = threadIdx.x; e < G::BODY; e += NT) dst[o_body + e] = GI[e];
    for (int e = threadIdx.x; e < NOUT * MLP_H + NOUT; e += NT) dst[o_hw + e] = GI[I_WH + e];   // head W then b: contiguous in both layouts
    if (threadIdx.x < 4) B.stats[((size_t)blockIdx.x * 2 + NET) * 4 + threadIdx.x] = GI[I_END + threadIdx.x];
#ifdef A2C3_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 0; i < 16; i++) B.stats[256 * 8 + NET * 16 + i] = (float)tacc[i]; }
#endif
}
