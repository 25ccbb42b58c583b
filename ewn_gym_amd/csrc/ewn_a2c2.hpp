// ewn_a2c2.hpp -- k_a2c_grad2: the A2C gradient pass of ewn_a2c.hpp with TWO waves per 32-sample tile, 5x5 boards.
//
// Why: k_a2c_grad keeps a whole net's gradient in one wave's registers (160 accumulators + the working tiles = 450 registers), so it
// runs one wave per SIMD -- and measured (tools/mfma_probe.hip) the f32-input MFMA overlaps with nothing on its SIMD, so a lone wave
// pays 64 cycles per MFMA, ~5 cycles per other instruction and every LDS / MFMA-result latency in full: 35 k cycles per tile and
// step for 16.5 k cycles of matrix work.  Here a tile belongs to a TEAM of two waves that split the hidden units: wave m owns units
// 32 m .. 32 m + 31 of both layers -- their rows of W1 / W2, their activations, their rows of every weight gradient.  Per wave that is
// half the MFMAs, half the elementwise work and half the accumulators (80 registers: the kernel fits 256, two waves per SIMD), and
// what one wave waits for the other issues under.  What crosses between the two waves goes through the LDS transposes that the
// sample-contracting products need anyway: T(h1) and T(g2) ([sample][unit], both halves) double as the other half's B operands of
// layer 2 and of dh1 = W2^T g2 (lane (j, h) reads unit kcol(ks, h) of its own sample j: one conflict-free ds_read per k-step).
// Four meetings of the team's two waves per step order the phases (a2c_team_sync: per-team LDS flags, not block barriers).
#pragma once
#include "ewn_a2c.hpp"

// The two waves of a team meet; nobody else is involved.  A block barrier would do (all teams run the same sequence), but it ties the
// four teams of a block -- and with them the two waves that share a SIMD, which belong to different teams -- into one phase: measured
// (PMC) the SIMDs then sat idle 27 % of the kernel, both of their waves waiting.  Per-team flags let the teams drift apart, so that one
// wave's MFMA chain runs while its SIMD neighbour waits for its own partner.  flag[m] = how many meetings wave m has reached; LDS
// operations of a wave complete in order, so the partner that sees my count also sees everything I stored before it.
EWN_DEV void a2c_team_sync(volatile int *flag, int m, int lane, int &epoch)
{
    epoch++;
    asm volatile("" ::: "memory");
    if (lane == 0) flag[m] = epoch;
    while (flag[1 - m] < epoch) __builtin_amdgcn_s_sleep(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int S> struct A2c2Geo {
    using G = MlpGeo<S>;
    using A = A2cGeo<S>;
    static_assert(S == 5, "row-split layout written for 32 features (one feature tile); 7x7 runs k_a2c_grad");
    static constexpr int XS = 33;                                   // [sample][feature] row stride
    // per team: XT x 2 (double-buffered: features of step t are written while the other wave may still read step t + 1's) | TA | TB | Dt | PX
    static constexpr int O_XT = 0, O_TA = O_XT + 2 * 32 * XS, O_TB = O_TA + 32 * A2C_TS, O_DT = O_TB + 32 * A2C_TS, O_PX = O_DT + 32 * 8;
    static constexpr int O_FLAG = O_PX + 2 * 8 * 32;                // two meeting counters (a2c_team_sync)
    static constexpr int TEAM_FLOATS = O_FLAG + 4;
    static constexpr int TEAMS = 4;
    static constexpr size_t lds_bytes() { return ((size_t)A::L_NET_END + (size_t)TEAMS * TEAM_FLOATS) * 4; }
};

// NET 0: policy body + action head; NET 1: value body + value head.  512 threads: 4 teams x 2 waves.
template <int S, int NET>
__global__ __launch_bounds__(512, 2) void k_a2c_grad2(A2cCfg c, A2cBuf B)
{
    using G = MlpGeo<S>;
    using A = A2cGeo<S>;
    using Q = A2c2Geo<S>;
    constexpr int CELLS = S * S, STR = RecGeo<S>::STR, NT = 512, NOUT = NET ? 1 : MLP_NA, KS1 = G::KS1;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *L = lds_f;
    float *W2T = L + A::L_W2T, *WhT = L + A::L_WHT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int team = wave >> 1, m = wave & 1;        // my tile of hidden units: 32 m .. 32 m + 31
    float *TM = L + A::L_NET_END + team * Q::TEAM_FLOATS;
    float *TA = TM + Q::O_TA, *TB = TM + Q::O_TB, *Dt = TM + Q::O_DT, *PX = TM + Q::O_PX;
    volatile int *flag = (volatile int *)(TM + Q::O_FLAG);
    int epoch = 0;
    float *GI = L + A::L_NET_END;                    // at the end: the block's gradient image (over the team areas)

    mlp_pack_net<S>(L, B.params, NET, threadIdx.x, NT);
    {
        const float *W2 = B.params + (NET ? G::O_VF : G::O_PI) + MLP_H * G::F + MLP_H;
        const float *Wh = B.params + (NET ? G::O_VW : G::O_AW);
        for (int e = threadIdx.x; e < 2 * 32 * 64; e += NT) {
            const int l = e & 63, ks = (e >> 6) & 31, mt = e >> 11;
            W2T[e] = W2[mlp_kcol(ks, l >> 5) * MLP_H + 32 * mt + (l & 31)];
        }
        for (int e = threadIdx.x; e < 2 * 3 * 64; e += NT) {
            const int l = e & 63, s = (e >> 6) % 3, mt = (e >> 6) / 3, row = 2 * s + (l >> 5);
            WhT[e] = row < NOUT ? Wh[row * MLP_H + 32 * mt + (l & 31)] : 0.0f;
        }
    }
    for (int e = m * 64 + lane; e < 32 * 8; e += 128) Dt[e] = 0.0f;
    if (lane < 2 && m == 0) flag[lane] = 0;
    __syncthreads();

    // my rows of the gradient: dW2 [32 m + row][64], dW1 [32 m + row][32 features], the head's columns 32 m .. (pi: a 32 x 32 MFMA tile
    // whose rows 0-4 count; vf: per-lane partials), the layer-2 bias partials
    f32x16 dW2[2] = { (f32x16)(0.0f), (f32x16)(0.0f) }, dW1 = (f32x16)(0.0f), dWh = (f32x16)(0.0f), db2l = (f32x16)(0.0f);
    float dbh[MLP_NA] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    float st_pl = 0.0f, st_vl = 0.0f, st_en = 0.0f;

    const int tiles = (c.N + 31) / 32, stride = (int)gridDim.x * Q::TEAMS, iters = (tiles + stride - 1) / stride;
    int parity = 0;
    #pragma unroll 1
    for (int it = 0; it < iters; it++) {            // the same trip count for both waves of a team (their meetings must pair up)
        const int tile = it * stride + (int)blockIdx.x * Q::TEAMS + team;
        const int game = tile * 32 + j;
        const bool valid = tile < tiles && game < c.N;
        const int gc = valid ? game : c.N - 1;
        float Rn = 0.0f;
        #pragma unroll 1
        for (int t = NET ? c.K : c.K - 1; t >= 0; t--) {
            // ---- features of observation t: lane (j, h) of wave m turns record bytes 16 m + 8 h .. + 7 of sample j into floats
            float *XT = TM + Q::O_XT + parity * 32 * Q::XS;
            parity ^= 1;
            const uint8_t *rrow = B.rec + ((size_t)t * c.N + gc) * STR;
            {
                // all eight bytes unconditionally (a per-byte `if (k < CELLS)` on a lane-dependent k cost an exec-mask dance per store: a
                // fifth of the loop's instructions): the lane with bytes 24 .. 31 stores the meta bytes as "features" 25 .. 31 too and
                // then overwrites them with the dice one-hot itself (same lane, program order)
                const uint2 v = *(const uint2 *)(rrow + 16 * m + 8 * h);
                const u32 w[2] = { v.x, v.y };
                float *xk = XT + j * Q::XS + 16 * m + 8 * h;
                #pragma unroll
                for (int i = 0; i < 8; i++) xk[i] = (float)(int)(int8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                if (m == 1 && h == 1) {             // bytes 24 .. 31: cell 24, then the dice at byte 25
                    const int dice = (int)(int8_t)((w[0] >> 8) & 0xFFu);
                    #pragma unroll
                    for (int dd = 0; dd < 7; dd++) xk[1 + dd] = (dd == dice - 1) ? 1.0f : 0.0f;
                }
            }
            a2c_team_sync(flag, m, lane, epoch);                        // (1) the tile's features are complete
            // ---- layer 1, my 32 units
            const int hh = h;
            f32x16 a = mlp_bias_acc(L + G::L_B1 + 32 * m, hh);
            #pragma unroll
            for (int s = 0; s < KS1; s++) {
                a = MLP_MFMA(L[G::L_W1 + (m * KS1 + s) * 64 + lane], XT[j * Q::XS + 2 * s + h], a);
                if ((s & 7) == 7) MLP_SCHED_FENCE();
            }
            const f32x16 h1 = mlp_tanh16(a);
            #pragma unroll
            for (int r = 0; r < 16; r++) TB[j * A2C_TS + 32 * m + mlp_row(r, h)] = h1[r];   // T(h1), my columns
            a2c_team_sync(flag, m, lane, epoch);                        // (2) T(h1) complete: the other half's units are layer 2's remaining B operands
            // ---- layer 2, my 32 units: k-steps of my own tile from registers, the other tile's from T(h1)
            f32x16 cacc = mlp_bias_acc(L + G::L_B2 + 32 * m, hh);
            #pragma unroll
            for (int r = 0; r < 16; r++) {          // my own tile's units: straight from the registers
                cacc = MLP_MFMA(L[G::L_W2 + (m * 32 + 16 * m + r) * 64 + lane], h1[r], cacc);
                if ((r & 7) == 7) MLP_SCHED_FENCE();
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) {          // the other wave's tile: its columns of T(h1)
                cacc = MLP_MFMA(L[G::L_W2 + (m * 32 + 16 * (1 - m) + r) * 64 + lane], TB[j * A2C_TS + 32 * (1 - m) + mlp_row(r, h)], cacc);
                if ((r & 7) == 7) MLP_SCHED_FENCE();
            }
            const f32x16 h2 = mlp_tanh16(cacc);
            // ---- the head on the VALU: my 16 units per lane, the other lane half, the other wave
            float out[NOUT];
            {
                const float *wh = L + G::L_WH + h * 8 * 32 + m * 16;
                #pragma unroll
                for (int q = 0; q < NOUT; q++) {
                    float acc = 0.0f;
                    #pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const float4 w = *(const float4 *)(wh + q * 32 + 4 * r4);
                        acc = fmaf(w.x, h2[4 * r4], acc); acc = fmaf(w.y, h2[4 * r4 + 1], acc); acc = fmaf(w.z, h2[4 * r4 + 2], acc); acc = fmaf(w.w, h2[4 * r4 + 3], acc);
                    }
                    const float o = mlp_other_half(acc, lane);
                    out[q] = h ? o + acc : acc + o;           // half 0 first: the same bits in both halves
                }
                if (h == 0) { for (int q = 0; q < NOUT; q++) PX[(m * 8 + q) * 32 + j] = out[q]; }
            }
            a2c_team_sync(flag, m, lane, epoch);                        // (3) both waves' partial head outputs are in PX
            #pragma unroll
            for (int q = 0; q < NOUT; q++) {
                const float p0 = PX[q * 32 + j], p1 = PX[(8 + q) * 32 + j];
                out[q] = (p0 + p1) + L[G::L_BH + q];
            }
            if (NET == 1 && t == c.K) { Rn = out[0]; continue; }   // V(s_K): the bootstrap value (uniform branch: every wave takes it)
            // ---- the loss of step t and its gradient w.r.t. the head outputs (the same numbers in both waves and lane halves)
            // (tried: these loads and the next step's record bytes issued a phase / a step ahead behind an LDS-only barrier -- no gain: the
            // global latency is not what the step waits for)
            const A2cStepIn sin = a2c_step_in<NET>(c, B, B.rec + ((size_t)(t + 1) * c.N + gc) * STR, CELLS, t, gc);   // row t + 1: action a_t, flags of step t
            float d[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
            a2c_loss_grad<NET>(c, B, sin, t, game, valid, m == 0 && h == 0, out, Rn, d, st_pl, st_vl, st_en);
            // ---- head gradients, my columns
            if constexpr (NET == 1) {
                #pragma unroll
                for (int r = 0; r < 16; r++) dWh[r] += d[0] * h2[r];          // per-lane partial of dWv[unit] = sum_s dV_s h2[unit][s]
                if (m == 0 && h == 0) dbh[0] += d[0];
            } else {
                #pragma unroll
                for (int r = 0; r < 16; r++) TA[j * A2C_TS + 32 * m + mlp_row(r, h)] = h2[r];   // T(h2), my columns
                if (h == 0) { *(float4 *)(Dt + j * 8) = make_float4(d[0], d[1], d[2], d[3]); Dt[j * 8 + 4] = d[4]; }   // both waves: the same values
                __builtin_amdgcn_wave_barrier();
                #pragma unroll
                for (int s = 0; s < 16; s++) {
                    const float av = (j < 8) ? Dt[(2 * s + h) * 8 + (j & 7)] : 0.0f;
                    dWh = MLP_MFMA(av, TA[(2 * s + h) * A2C_TS + 32 * m + j], dWh);
                    if ((s & 7) == 7) MLP_SCHED_FENCE();
                }
                if (m == 0 && h == 0) { for (int i = 0; i < 5; i++) dbh[i] += d[i]; }
            }
            // ---- dh2 = Wh^T d (my units), g2 = dh2 (1 - h2^2)
            f32x16 g2 = (f32x16)(0.0f);
            #pragma unroll
            for (int s = 0; s < (NET ? 1 : 3); s++) g2 = MLP_MFMA(WhT[(m * 3 + s) * 64 + lane], h ? d[2 * s + 1] : d[2 * s], g2);
            #pragma unroll
            for (int r = 0; r < 16; r++) { g2[r] *= 1.0f - h2[r] * h2[r]; db2l[r] += g2[r]; }
            // ---- dW2[my rows] += g2 . h1^T: A = T(g2) (my columns, just written), B = T(h1) (both halves, complete since (2))
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int r = 0; r < 16; r++) TA[j * A2C_TS + 32 * m + mlp_row(r, h)] = g2[r];
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int s = 0; s < 16; s++) {
                const float av = TA[(2 * s + h) * A2C_TS + 32 * m + j];
                dW2[0] = MLP_MFMA(av, TB[(2 * s + h) * A2C_TS + j], dW2[0]);
                dW2[1] = MLP_MFMA(av, TB[(2 * s + h) * A2C_TS + 32 + j], dW2[1]);
                if ((s & 3) == 3) MLP_SCHED_FENCE();
            }
            a2c_team_sync(flag, m, lane, epoch);                        // (4) T(g2) complete, and nobody reads T(h1) any more
            // ---- dh1 = W2^T g2 (my units): my own g2 from registers, the other half's from T(g2); g1 = dh1 (1 - h1^2)
            f32x16 g1 = (f32x16)(0.0f);
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                g1 = MLP_MFMA(W2T[(m * 32 + 16 * m + r) * 64 + lane], g2[r], g1);
                if ((r & 7) == 7) MLP_SCHED_FENCE();
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                g1 = MLP_MFMA(W2T[(m * 32 + 16 * (1 - m) + r) * 64 + lane], TA[j * A2C_TS + 32 * (1 - m) + mlp_row(r, h)], g1);
                if ((r & 7) == 7) MLP_SCHED_FENCE();
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) g1[r] *= 1.0f - h1[r] * h1[r];
            // ---- dW1[my rows] += g1 . x^T: T(g1) goes where T(h1) was (my columns); db1 = the dice columns' sum, taken at the end
            #pragma unroll
            for (int r = 0; r < 16; r++) TB[j * A2C_TS + 32 * m + mlp_row(r, h)] = g1[r];
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int s = 0; s < 16; s++) {
                dW1 = MLP_MFMA(TB[(2 * s + h) * A2C_TS + 32 * m + j], XT[(2 * s + h) * Q::XS + j], dW1);
                if ((s & 7) == 7) MLP_SCHED_FENCE();
            }
        }
    }

    // ---- the block's gradient image: teams add in a fixed order, the two waves of a team own disjoint rows
    constexpr int I_W1 = 0, I_B1 = I_W1 + MLP_H * G::F, I_W2 = I_B1 + MLP_H, I_B2 = I_W2 + MLP_H * MLP_H, I_WH = I_B2 + MLP_H,
                  I_BH = I_WH + NOUT * MLP_H, I_END = I_BH + NOUT;
    #pragma unroll
    for (int r = 0; r < 16; r++) { db2l[r] = a2c_sum32(db2l[r]); if (NET == 1) dWh[r] = a2c_sum32(dWh[r]); }
    #pragma unroll
    for (int i = 0; i < NOUT; i++) { dbh[i] = a2c_sum32(dbh[i]); dbh[i] += __shfl_xor(dbh[i], 32, 64); }
    st_pl = a2c_sum32(st_pl); st_vl = a2c_sum32(st_vl); st_en = a2c_sum32(st_en);
    __syncthreads();
    #pragma unroll 1
    for (int w = 0; w < Q::TEAMS; w++) {
        if (team == w) {
            const bool first = w == 0;
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32 * m + mlp_row(r, h);
                #pragma unroll
                for (int nt = 0; nt < 2; nt++) { float *p = GI + I_W2 + row * MLP_H + 32 * nt + j; *p = (first ? 0.0f : *p) + dW2[nt][r]; }
                { float *p = GI + I_W1 + row * G::F + j; *p = (first ? 0.0f : *p) + dW1[r]; }
                if (j == 0) { float *p = GI + I_B2 + row; *p = (first ? 0.0f : *p) + db2l[r]; }
                if (NET == 1 && j == 0) { float *p = GI + I_WH + row; *p = (first ? 0.0f : *p) + dWh[r]; }
                if constexpr (NET == 0) {        // dWa tile: row a = mlp_row(r, h) (a < 5), unit = 32 m + j
                    const int aa = mlp_row(r, h);
                    if (aa < MLP_NA) { float *p = GI + I_WH + aa * MLP_H + 32 * m + j; *p = (first ? 0.0f : *p) + dWh[r]; }
                }
            }
            if (m == 0 && lane == 0) {
                #pragma unroll
                for (int i = 0; i < NOUT; i++) { float *p = GI + I_BH + i; *p = (first ? 0.0f : *p) + dbh[i]; }
                float *sp = GI + I_END;
                sp[0] = (first ? 0.0f : sp[0]) + st_pl; sp[1] = (first ? 0.0f : sp[1]) + st_vl; sp[2] = (first ? 0.0f : sp[2]) + st_en; sp[3] = 0.0f;
            }
        }
        __syncthreads();
    }
    for (int row = threadIdx.x; row < MLP_H; row += NT) {
        float sacc = 0.0f;
        for (int dd = 0; dd < 7; dd++) sacc += GI[I_W1 + row * G::F + CELLS + dd];
        GI[I_B1 + row] = sacc;
    }
    __syncthreads();
    float *dst = B.partial + (size_t)blockIdx.x * G::P;
    const int o_body = NET ? G::O_VF : G::O_PI, o_hw = NET ? G::O_VW : G::O_AW;
    for (int e = threadIdx.x; e < G::BODY; e += NT) dst[o_body + e] = GI[e];
    for (int e = threadIdx.x; e < NOUT * MLP_H + NOUT; e += NT) dst[o_hw + e] = GI[I_WH + e];
    if (threadIdx.x < 4) B.stats[((size_t)blockIdx.x * 2 + NET) * 4 + threadIdx.x] = GI[I_END + threadIdx.x];
}
