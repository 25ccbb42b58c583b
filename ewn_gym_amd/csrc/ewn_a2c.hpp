// ewn_a2c.hpp -- the A2C update of the reference's trainer (train.py:35-63, 148: stable_baselines3 A2C.train on the n-step
// rollout: policy-gradient loss + vf_coef * value MSE + ent_coef * entropy bonus, one RMSprop step, max_grad_norm clipping) as
// three kernels over the trajectory records ewn_step_k_policy writes:
//   k_a2c_grad<S, 1>  value body: bootstrap V(s_K), n-step returns (GAE with lambda = 1, SB3's A2C default), forward + backward of
//                     the value loss, advantages R_t - V(s_t) left in a scratch column for ...
//   k_a2c_grad<S, 0>  policy body: forward (recomputed: same parameters as the rollout, nothing was stored), policy-gradient and
//                     entropy terms, backward
//   k_a2c_reduce      sums the per-block partial gradients into the flat gradient (the bucket a multi-GPU job all-reduces), and
//   k_a2c_apply       clips by the global norm and takes the RMSprop step on the flat parameter vector.
// Forward and backward run on the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32, ewn_mlp.hpp): 32 samples per tile with the
// sample on the lane.  The products that sum over SAMPLES (dW = dpre . h^T) need both operands with the unit on the lane instead:
// those tiles go through a per-wave LDS transpose ([sample][unit], odd row stride: conflict-free both ways).  Weight gradients
// accumulate in MFMA accumulators for all the samples a wave sees (one wave per SIMD: the register file is the gradient buffer),
// bias gradients as per-lane partial sums; a block reduces its waves in a fixed order, so results are bit-reproducible.
#pragma once
#include "ewn_mlp.hpp"
#include "ewn_rollout.hpp"

struct A2cCfg { int N, K; float gamma, vf_coef, ent_coef, inv_batch; };
struct A2cBuf {
    const uint8_t *rec;      // [K + 1][N][STR] trajectory records, row 0 = the observation before step 0
    const double *reward;    // [K][N]
    const float *params;     // [P]
    float *adv;              // [K][N] advantages: written by the value pass, read by the policy pass
    float *partial;          // [blocks][P] per-block gradient sums (each pass writes its own body's and head's entries)
    float *stats;            // [blocks][2][4] per block and pass: loss sums {policy, value, entropy, -}
};

#define A2C_TS 65            // row stride (floats) of the [sample][unit] transpose tiles

template <int S> struct A2cGeo {
    using G = MlpGeo<S>;
    static constexpr int FT = (G::FP + 31) / 32;        // 32-wide feature tiles of dW1
    static constexpr int XS = FT * 32 + 1;              // row stride of the [sample][feature] tile
    static constexpr int L_W2T = G::L_END, L_WHT = L_W2T + 2 * 32 * 64, L_NET_END = L_WHT + 2 * 3 * 64; // the block's weight images
    static constexpr int WAVE_FLOATS = 32 * XS + 2 * 32 * A2C_TS + 32 * 8;   // XT | TA | TB | Dt per wave
    static constexpr int NET_PARAMS = G::BODY + MLP_NA * MLP_H + MLP_NA;     // the larger of the two nets' parameter counts (pi)
};

template <int S, int NWV>
constexpr size_t a2c_lds_bytes()
{
    using A = A2cGeo<S>;
    const size_t wave_area = (size_t)NWV * A::WAVE_FLOATS, grad_img = (size_t)A::NET_PARAMS + 8;
    return ((size_t)A::L_NET_END + (wave_area > grad_img ? wave_area : grad_img)) * 4;
}

// MFMA-layout activations (unit 32 tile + mlp_row(r, h) of sample j) -> LDS [sample][unit]
EWN_DEV void a2c_tile_to_lds(float *T, const f32x16 (&v)[2], int j, int h)
{
    #pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) T[j * A2C_TS + 32 * mt + mlp_row(r, h)] = v[mt][r];
    }
}

// sum over the 32 sample lanes of a half (lanes with the same lane >> 5): every lane of the half ends up with the total
EWN_DEV float a2c_sum32(float x)
{
    #pragma unroll
    for (int m = 1; m < 32; m <<= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// The loss of one sample at step t and its gradient d[] w.r.t. the head's outputs (SB3 A2C.train: policy_loss = -(adv * log_prob).mean(),
// value_loss = mse(returns, values), entropy_loss = -entropy.mean(); loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss).
// NET 1 (value pass): the n-step return R_t = r_t + gamma (1 - done_t) R_{t+1} (GAE with lambda 1), the advantage R_t - V(s_t) left in
// B.adv for the policy pass.  NET 0 (policy pass): two categoricals (flag: logits 0-1, direction: logits 2-4).  stat_lane: this lane
// writes the advantage and counts the loss sums (one lane per sample).
// what a step's loss needs besides the head outputs
struct A2cStepIn { int a0, a1; bool term; float rew, adv; };
template <int NET>
EWN_DEV A2cStepIn a2c_step_in(const A2cCfg &c, const A2cBuf &B, const uint8_t *nrow, int cells, int t, int gc)
{
    A2cStepIn in;
    in.a0 = nrow[cells + 1]; in.a1 = nrow[cells + 2]; in.term = nrow[cells + 3] != 0;
    in.rew = NET == 1 ? (float)B.reward[(size_t)t * c.N + gc] : 0.0f;
    in.adv = NET == 0 ? B.adv[(size_t)t * c.N + gc] : 0.0f;
    return in;
}

template <int NET>
EWN_DEV void a2c_loss_grad(const A2cCfg &c, const A2cBuf &B, const A2cStepIn &in, int t, int game, bool valid, bool stat_lane,
                           const float *out, float &Rn, float (&d)[6], float &st_pl, float &st_vl, float &st_en)
{
    if constexpr (NET == 1) {
        const float R = in.rew + (in.term ? 0.0f : c.gamma * Rn);
        Rn = R;
        const float V = out[0];
        if (valid) {
            if (stat_lane) { B.adv[(size_t)t * c.N + game] = R - V; st_vl += (R - V) * (R - V); }
            d[0] = 2.0f * c.vf_coef * (V - R) * c.inv_batch;
        }
    } else {
        const float *lg = out;
        const int a0 = in.a0, a1 = in.a1;
        const float adv = in.adv;
        const float m0 = fmaxf(lg[0], lg[1]), m1 = fmaxf(lg[2], fmaxf(lg[3], lg[4]));
        const float e0 = __expf(lg[0] - m0), e1 = __expf(lg[1] - m0), e2 = __expf(lg[2] - m1), e3 = __expf(lg[3] - m1), e4 = __expf(lg[4] - m1);
        const float z0 = e0 + e1, z1 = e2 + e3 + e4, lz0 = __logf(z0), lz1 = __logf(z1);
        const float pr[5] = { e0 / z0, e1 / z0, e2 / z1, e3 / z1, e4 / z1 };
        const float lp[5] = { lg[0] - m0 - lz0, lg[1] - m0 - lz0, lg[2] - m1 - lz1, lg[3] - m1 - lz1, lg[4] - m1 - lz1 };
        const float H0 = -(pr[0] * lp[0] + pr[1] * lp[1]), H1 = -(pr[2] * lp[2] + pr[3] * lp[3] + pr[4] * lp[4]);
        const float logp = (a0 ? lp[1] : lp[0]) + (a1 == 0 ? lp[2] : (a1 == 1 ? lp[3] : lp[4]));
        if (valid) {
            if (stat_lane) { st_pl += -adv * logp; st_en += H0 + H1; }
            #pragma unroll
            for (int i = 0; i < 5; i++) {
                const float oh = (i < 2 ? (a0 == i) : (a1 == i - 2)) ? 1.0f : 0.0f;
                const float Hh = i < 2 ? H0 : H1;
                d[i] = c.inv_batch * (-adv * (oh - pr[i]) + c.ent_coef * pr[i] * (lp[i] + Hh));   // d(-H)/dl_i = p_i (log p_i + H)
            }
        }
    }
}

// NET 0: policy body + action head; NET 1: value body + value head.  NWV waves per block, one per SIMD.
template <int S, int NET, int NWV>
__global__ __launch_bounds__(NWV * 64, 1) void k_a2c_grad(A2cCfg c, A2cBuf B)
{
    using G = MlpGeo<S>;
    using A = A2cGeo<S>;
    constexpr int CELLS = S * S, STR = RecGeo<S>::STR, NT = NWV * 64, NOUT = NET ? 1 : MLP_NA;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *L = lds_f;                               // forward image of this net (ewn_mlp.hpp)
    float *W2T = L + A::L_W2T;                      // [tile][k-step][lane]: W2[kcol(ks, h)][32 tile + i]: dh1 = W2^T dpre2
    float *WhT = L + A::L_WHT;                      // [tile][3][lane]: Wh[2 s + h][32 tile + i] (rows past the head's are zero)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    float *XT = L + A::L_NET_END + wave * A::WAVE_FLOATS;   // [sample][feature]
    float *TA = XT + 32 * A::XS, *TB = TA + 32 * A2C_TS, *Dt = TB + 32 * A2C_TS;
    float *GI = L + A::L_NET_END;                   // at the end: the block's gradient image (over the per-wave areas)

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
    for (int e = lane; e < 32 * A::XS; e += 64) XT[e] = 0.0f;   // padding features stay zero
    for (int e = lane; e < 32 * 8; e += 64) Dt[e] = 0.0f;
    __syncthreads();

    // gradient accumulators of everything this wave sees
    f32x16 dW2[2][2], dW1[2][A::FT], dWh[2], db2l[2];
    #pragma unroll
    for (int a = 0; a < 2; a++) {
        #pragma unroll
        for (int b = 0; b < 2; b++) dW2[a][b] = (f32x16)(0.0f);
        #pragma unroll
        for (int b = 0; b < A::FT; b++) dW1[a][b] = (f32x16)(0.0f);
        dWh[a] = (f32x16)(0.0f); db2l[a] = (f32x16)(0.0f);
    }
    float dbh[MLP_NA] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    float st_pl = 0.0f, st_vl = 0.0f, st_en = 0.0f;

    const int tiles = (c.N + 31) / 32;
    #pragma unroll 1
    for (int tile = (int)blockIdx.x * NWV + wave; tile < tiles; tile += (int)gridDim.x * NWV) {
        const int game = tile * 32 + j;
        const bool valid = game < c.N;
        const int gc = valid ? game : c.N - 1;
        float Rn = 0.0f;
        // t = K: the bootstrap value V(s_K) (value pass only); t = K-1 .. 0: forward + backward of step t
        #pragma unroll 1
        for (int t = NET ? c.K : c.K - 1; t >= 0; t--) {
            // ---- features of observation t: the board cells of record row t as floats, one-hot dice
            const uint8_t *rrow = B.rec + ((size_t)t * c.N + gc) * STR;
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int c0 = 0; c0 < RecGeo<S>::NCH; c0 += 2) {
                const int ch = c0 + h;
                if (ch < RecGeo<S>::NCH) {
                    const uint4 v = *(const uint4 *)(rrow + 16 * ch);
                    const u32 w[4] = { v.x, v.y, v.z, v.w };
                    #pragma unroll
                    for (int i = 0; i < 16; i++) {
                        if (16 * c0 + i < CELLS) {
                            const int k = 16 * ch + i;
                            if (k < CELLS) XT[j * A::XS + k] = (float)(int)(int8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                        }
                    }
                }
            }
            if (h == 0) {
                const int dice = (int8_t)rrow[CELLS];
                #pragma unroll
                for (int d = 0; d < 7; d++) XT[j * A::XS + CELLS + d] = (d == dice - 1) ? 1.0f : 0.0f;
            }
            __builtin_amdgcn_wave_barrier();
            f32x16 h1[2], h2[2];
            float out[NOUT];                     // the head's outputs of my sample, the same in both lane halves
            const float *xrow = XT + j * A::XS + h;
            mlp_forward<S, NOUT>(L, lane, [&](int s) { return xrow[2 * s]; }, h1, h2, out);
            if (NET == 1 && t == c.K) { Rn = out[0]; continue; }   // V(s_K)
            // ---- the loss of step t and its gradient w.r.t. the head outputs (both lane halves hold the same numbers)
            const A2cStepIn sin = a2c_step_in<NET>(c, B, B.rec + ((size_t)(t + 1) * c.N + gc) * STR, CELLS, t, gc);   // row t + 1: action a_t, flags of step t
            float d[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
            a2c_loss_grad<NET>(c, B, sin, t, game, valid, h == 0, out, Rn, d, st_pl, st_vl, st_en);
            // ---- head gradients
            if constexpr (NET == 1) {
                #pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    #pragma unroll
                    for (int r = 0; r < 16; r++) dWh[mt][r] += d[0] * h2[mt][r];     // per-lane partial of dWv[unit] = sum_s dV_s h2[unit][s]
                }
                if (h == 0) dbh[0] += d[0];
            } else {
                __builtin_amdgcn_wave_barrier();
                a2c_tile_to_lds(TA, h2, j, h);
                if (h == 0) { *(float4 *)(Dt + j * 8) = make_float4(d[0], d[1], d[2], d[3]); Dt[j * 8 + 4] = d[4]; }
                __builtin_amdgcn_wave_barrier();
                // dWa[a][unit] = sum_s d[a][s] h2[unit][s]: A = d^T (row a on the lane: rows past 7 are zero), B = h2^T
                #pragma unroll
                for (int s = 0; s < 16; s++) {
                    const float av = (j < 8) ? Dt[(2 * s + h) * 8 + (j & 7)] : 0.0f;
                    dWh[0] = MLP_MFMA(av, TA[(2 * s + h) * A2C_TS + j], dWh[0]);
                    dWh[1] = MLP_MFMA(av, TA[(2 * s + h) * A2C_TS + 32 + j], dWh[1]);
                }
                if (h == 0) { for (int i = 0; i < 5; i++) dbh[i] += d[i]; }
            }
            // ---- dh2 = Wh^T d, dpre2 = dh2 (1 - h2^2)
            f32x16 g2[2] = { (f32x16)(0.0f), (f32x16)(0.0f) };
            #pragma unroll
            for (int s = 0; s < (NET ? 1 : 3); s++) {
                const float bv = h ? d[2 * s + 1] : d[2 * s];
                g2[0] = MLP_MFMA(WhT[s * 64 + lane], bv, g2[0]);
                g2[1] = MLP_MFMA(WhT[(3 + s) * 64 + lane], bv, g2[1]);
            }
            #pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                #pragma unroll
                for (int r = 0; r < 16; r++) { g2[mt][r] *= 1.0f - h2[mt][r] * h2[mt][r]; db2l[mt][r] += g2[mt][r]; }
            }
            // ---- dW2 += dpre2 . h1^T (sum over the tile's samples: both operands through the LDS transpose)
            __builtin_amdgcn_wave_barrier();
            a2c_tile_to_lds(TA, g2, j, h);
            a2c_tile_to_lds(TB, h1, j, h);
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int s = 0; s < 16; s++) {
                const float *ta = TA + (2 * s + h) * A2C_TS + j, *tb = TB + (2 * s + h) * A2C_TS + j;
                const float a0v = ta[0], a1v = ta[32], b0v = tb[0], b1v = tb[32];
                dW2[0][0] = MLP_MFMA(a0v, b0v, dW2[0][0]); dW2[0][1] = MLP_MFMA(a0v, b1v, dW2[0][1]);
                dW2[1][0] = MLP_MFMA(a1v, b0v, dW2[1][0]); dW2[1][1] = MLP_MFMA(a1v, b1v, dW2[1][1]);
                if ((s & 3) == 3) MLP_SCHED_FENCE();
            }
            // ---- dh1 = W2^T dpre2 (dpre2 straight from the registers), dpre1 = dh1 (1 - h1^2)
            f32x16 g1[2] = { (f32x16)(0.0f), (f32x16)(0.0f) };
            #pragma unroll
            for (int ks = 0; ks < 32; ks++) {
                const float bv = g2[ks >> 4][ks & 15];
                g1[0] = MLP_MFMA(W2T[ks * 64 + lane], bv, g1[0]);
                g1[1] = MLP_MFMA(W2T[(32 + ks) * 64 + lane], bv, g1[1]);
                if ((ks & 7) == 7) MLP_SCHED_FENCE();
            }
            #pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                #pragma unroll
                for (int r = 0; r < 16; r++) g1[mt][r] *= 1.0f - h1[mt][r] * h1[mt][r];
            }
            // ---- dW1 += dpre1 . x^T (the bias gradient db1 is the sum of its seven one-hot dice columns: taken at the end)
            __builtin_amdgcn_wave_barrier();
            a2c_tile_to_lds(TA, g1, j, h);
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int s = 0; s < 16; s++) {
                const float *ta = TA + (2 * s + h) * A2C_TS + j;
                const float a0v = ta[0], a1v = ta[32];
                #pragma unroll
                for (int ft = 0; ft < A::FT; ft++) {
                    const float bv = XT[(2 * s + h) * A::XS + 32 * ft + j];
                    dW1[0][ft] = MLP_MFMA(a0v, bv, dW1[0][ft]);
                    dW1[1][ft] = MLP_MFMA(a1v, bv, dW1[1][ft]);
                }
                if ((s & 3) == 3) MLP_SCHED_FENCE();
            }
        }
    }

    // ---- the block's gradient image in LDS: waves add in a fixed order (bit-reproducible), then one coalesced copy out
    constexpr int I_W1 = 0, I_B1 = I_W1 + MLP_H * G::F, I_W2 = I_B1 + MLP_H, I_B2 = I_W2 + MLP_H * MLP_H, I_WH = I_B2 + MLP_H,
                  I_BH = I_WH + NOUT * MLP_H, I_END = I_BH + NOUT;
    #pragma unroll
    for (int mt = 0; mt < 2; mt++) {     // bias partials: sum over the 32 sample lanes of my half
        #pragma unroll
        for (int r = 0; r < 16; r++) { db2l[mt][r] = a2c_sum32(db2l[mt][r]); if (NET == 1) dWh[mt][r] = a2c_sum32(dWh[mt][r]); }
    }
    #pragma unroll
    for (int i = 0; i < NOUT; i++) { dbh[i] = a2c_sum32(dbh[i]); dbh[i] += __shfl_xor(dbh[i], 32, 64); }
    st_pl = a2c_sum32(st_pl); st_vl = a2c_sum32(st_vl); st_en = a2c_sum32(st_en);
    __syncthreads();                     // every wave is done with its transpose tiles: the area becomes the gradient image
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
                    for (int ft = 0; ft < A::FT; ft++) {
                        const int col = 32 * ft + j;
                        if (col < G::F) { float *p = GI + I_W1 + row * G::F + col; *p = (first ? 0.0f : *p) + dW1[mt][ft][r]; }
                    }
                    if (j == 0) { float *p = GI + I_B2 + row; *p = (first ? 0.0f : *p) + db2l[mt][r]; }
                    if (NET == 1 && j == 0) { float *p = GI + I_WH + row; *p = (first ? 0.0f : *p) + dWh[mt][r]; }
                }
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
}

// ---- partials -> flat gradient (+ loss sums and the squared norm's per-block pieces)
struct A2cRedBuf { const float *partial; const float *stats; float *grad; int blocks, P; };

#define A2C_RED_E 32     // elements per block of k_a2c_reduce
__global__ __launch_bounds__(256) void k_a2c_reduce(A2cRedBuf B)
{
    // grad[i] = sum over blocks of partial[b][i]; grad[P .. P + 7] = loss sums {policy, value, entropy, 0} x {pi pass, vf pass}.
    // 32 elements per block, eight threads per element (each sums every eighth block with independent loads in flight, then the
    // eight partial sums are added in a fixed order): the first version walked all blocks in one thread per element, a chain of
    // 256 dependent-latency loads on 13 k threads (88 us for 13 MB).
    __shared__ float part[8][A2C_RED_E];
    const int e = (int)threadIdx.x & (A2C_RED_E - 1), q = (int)threadIdx.x / A2C_RED_E, i = (int)blockIdx.x * A2C_RED_E + e;
    float s = 0.0f;
    if (i < B.P) {
        const float *p = B.partial + i;
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        int b = q;
        for (; b + 24 < B.blocks; b += 32) {
            #pragma unroll
            for (int u = 0; u < 4; u++) acc[u] += p[(size_t)(b + 8 * u) * B.P];
        }
        for (; b < B.blocks; b += 8) acc[0] += p[(size_t)b * B.P];
        s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    } else if (i < B.P + 8) {
        for (int b = q; b < B.blocks; b += 8) s += B.stats[(size_t)b * 8 + (i - B.P)];
    }
    part[q][e] = s;
    __syncthreads();
    if (q == 0 && i < B.P + 8)
        B.grad[i] = ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) + ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e]));
}

// The same sum with 8-byte loads and every load of a thread in flight at once: 16 element PAIRS per block, sixteen threads per pair, each
// summing every sixteenth block (16 independent loads for the 256 blocks of a full launch), then the sixteen partial sums in a fixed
// order.  Needs an even P and an 8-byte aligned `partial` (the launcher checks; k_a2c_reduce above is the fallback).  11.2 -> measured
// in profiles/r03/a2c/kernel_stats.csv.
#define A2C_RED2_E 16
__global__ __launch_bounds__(256) void k_a2c_reduce2(A2cRedBuf B)
{
    __shared__ float2 part[16][A2C_RED2_E];
    const int e = (int)threadIdx.x & (A2C_RED2_E - 1), q = (int)threadIdx.x / A2C_RED2_E, i = 2 * ((int)blockIdx.x * A2C_RED2_E + e);
    float2 s = make_float2(0.0f, 0.0f);
    if (i < B.P) {
        const float2 *p = (const float2 *)(B.partial + i);
        const size_t row = (size_t)(B.P >> 1);
        float2 v[16];
        int b = q;
        for (; b + 240 < B.blocks; b += 256) {
            #pragma unroll
            for (int u = 0; u < 16; u++) v[u] = p[(size_t)(b + 16 * u) * row];
            #pragma unroll
            for (int u = 0; u < 16; u += 4) {   // a fixed tree: bit-reproducible
                s.x += (v[u].x + v[u + 1].x) + (v[u + 2].x + v[u + 3].x);
                s.y += (v[u].y + v[u + 1].y) + (v[u + 2].y + v[u + 3].y);
            }
        }
        for (; b < B.blocks; b += 16) { const float2 w = p[(size_t)b * row]; s.x += w.x; s.y += w.y; }
    } else if (i < B.P + 8) {
        for (int b = q; b < B.blocks; b += 16) { s.x += B.stats[(size_t)b * 8 + (i - B.P)]; s.y += B.stats[(size_t)b * 8 + (i - B.P) + 1]; }
    }
    part[q][e] = s;
    __syncthreads();
    if (q == 0 && i < B.P + 8) {
        float2 t = make_float2(0.0f, 0.0f);
        #pragma unroll
        for (int k = 0; k < 16; k += 4) {
            t.x += (part[k][e].x + part[k + 1][e].x) + (part[k + 2][e].x + part[k + 3][e].x);
            t.y += (part[k][e].y + part[k + 1][e].y) + (part[k + 2][e].y + part[k + 3][e].y);
        }
        B.grad[i] = t.x; B.grad[i + 1] = t.y;
    }
}

// ---- clip_grad_norm_(max_grad_norm) + RMSprop(alpha, eps) step (torch.optim.RMSprop as SB3's A2C configures it: centered = False,
// momentum 0, weight_decay 0): sq = alpha sq + (1 - alpha) g^2; p -= lr g / (sqrt(sq) + eps).  ONE block (13 k parameters): the norm
// needs every element, and a second launch would cost more than the arithmetic.
struct A2cApplyCfg { int P; float lr, alpha, eps, max_norm, grad_scale; };

// The same update with every load in flight at once (16-byte aligned buffers, P <= 1024 x 4 x A2C_APPLY_V): a thread's float4 pieces
// of the gradient are loaded together and stay in registers between the norm and the step, then its pieces of sq_avg and params are
// loaded together -- three memory round trips in all.  The plain loop below makes 2 x 13 dependent ones: the kernel is one block,
// nothing else hides them (11.4 us -> measured below).
#define A2C_APPLY_V 4
__global__ __launch_bounds__(1024) void k_a2c_apply_v4(A2cApplyCfg c, float *params, float *sq_avg, const float *grad, float *norm_out)
{
    __shared__ float red[16];
    const int n4 = c.P >> 2, tail = c.P & 3, tid = (int)threadIdx.x;
    float4 g[A2C_APPLY_V];
    #pragma unroll
    for (int v = 0; v < A2C_APPLY_V; v++) { const int i = tid + v * 1024; g[v] = i < n4 ? ((const float4 *)grad)[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    const float gt = tid < tail ? grad[4 * n4 + tid] * c.grad_scale : 0.0f;
    float4 q[A2C_APPLY_V], w[A2C_APPLY_V];
    #pragma unroll
    for (int v = 0; v < A2C_APPLY_V; v++) {
        const int i = tid + v * 1024;
        if (i < n4) { q[v] = ((const float4 *)sq_avg)[i]; w[v] = ((const float4 *)params)[i]; }
    }
    float ss = gt * gt;
    #pragma unroll
    for (int v = 0; v < A2C_APPLY_V; v++) {
        g[v].x *= c.grad_scale; g[v].y *= c.grad_scale; g[v].z *= c.grad_scale; g[v].w *= c.grad_scale;
        ss += (g[v].x * g[v].x + g[v].y * g[v].y) + (g[v].z * g[v].z + g[v].w * g[v].w);
    }
    #pragma unroll
    for (int m = 1; m < 64; m <<= 1) ss += __shfl_xor(ss, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
    #pragma unroll
    for (int k = 0; k < 16; k++) tot += red[k];
    const float norm = sqrtf(tot);
    const float clip = c.max_norm > 0.0f ? fminf(1.0f, c.max_norm / (norm + 1e-6f)) : 1.0f;   // torch.nn.utils.clip_grad_norm_
    auto step = [&](float gg, float &sq, float &pp) {
        gg *= clip;
        sq = c.alpha * sq + (1.0f - c.alpha) * gg * gg;
        pp -= c.lr * gg / (sqrtf(sq) + c.eps);
    };
    #pragma unroll
    for (int v = 0; v < A2C_APPLY_V; v++) {
        const int i = tid + v * 1024;
        if (i < n4) {
            step(g[v].x, q[v].x, w[v].x); step(g[v].y, q[v].y, w[v].y); step(g[v].z, q[v].z, w[v].z); step(g[v].w, q[v].w, w[v].w);
            ((float4 *)sq_avg)[i] = q[v]; ((float4 *)params)[i] = w[v];
        }
    }
    if (tid < tail) { float sq = sq_avg[4 * n4 + tid], pp = params[4 * n4 + tid]; step(gt, sq, pp); sq_avg[4 * n4 + tid] = sq; params[4 * n4 + tid] = pp; }
    if (tid == 0 && norm_out) *norm_out = norm;
}

__global__ __launch_bounds__(1024) void k_a2c_apply(A2cApplyCfg c, float *params, float *sq_avg, const float *grad, float *norm_out)
{
    __shared__ float red[16];
    float ss = 0.0f;
    for (int i = threadIdx.x; i < c.P; i += 1024) { const float g = grad[i] * c.grad_scale; ss += g * g; }
    #pragma unroll
    for (int m = 1; m < 64; m <<= 1) ss += __shfl_xor(ss, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
    #pragma unroll
    for (int w = 0; w < 16; w++) tot += red[w];
    const float norm = sqrtf(tot);
    const float clip = c.max_norm > 0.0f ? fminf(1.0f, c.max_norm / (norm + 1e-6f)) : 1.0f;   // torch.nn.utils.clip_grad_norm_
    for (int i = threadIdx.x; i < c.P; i += 1024) {
        const float g = grad[i] * c.grad_scale * clip;
        const float sq = c.alpha * sq_avg[i] + (1.0f - c.alpha) * g * g;
        sq_avg[i] = sq;
        params[i] -= c.lr * g / (sqrtf(sq) + c.eps);
    }
    if (threadIdx.x == 0 && norm_out) *norm_out = norm;
}
