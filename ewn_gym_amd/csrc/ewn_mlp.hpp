// ewn_mlp.hpp -- the actor-critic of the reference's trainer (train.py:35-63: stable_baselines3 "MultiInputPolicy" with
// activation_fn=Tanh, i.e. SB3's default two SEPARATE 64-64 tanh bodies for policy and value, a MultiDiscrete([2, 3]) action
// head of 5 logits and a scalar value head) evaluated on the matrix cores, shared by the policy-driven rollout kernel
// (k_rollout_mlp) and the fused A2C gradient kernels (k_a2c_grad).
//
// This is the one GEMM-shaped piece of the repo, so it is the one place MFMA applies: exact-f32 `v_mfma_f32_32x32x2_f32`
// (a k-ordered fmaf chain, bit for bit), 32 samples (games) per tile.  Orientation: Y[unit][sample] = W[unit][k] X[k][sample],
// i.e. the WEIGHTS are the A operand (read from an LDS image pre-arranged in operand order) and the activations the B operand.
// The 32x32 result has its sample on the lane (lane & 31) and its units in the 16 registers, which is exactly what the next
// layer's B operand wants (it sums over units = registers): a layer's output feeds the next MFMA with no lane movement and no
// LDS round trip.  Register r of lane half h = lane >> 5 holds unit row mlp_row(r, h); a k-step of the next layer takes
// register r from BOTH halves as its two k values, so the weight image lists the columns in that order.
#pragma once
#include "ewn_core.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MLP_H 64     // hidden width of both bodies (SB3 default net_arch)
#define MLP_NA 5     // logits of MultiDiscrete([2, 3]) (envs/ewn.py:59)

// features: the S*S board cells as floats, then one_hot(dice_roll - 1) of width cube_num + 1 = 7 (the observation space's
// Discrete(cube_num + 1, start=1), envs/ewn.py:66-68; cube_layer 3)
template <int S> struct MlpGeo {
    static constexpr int F = S * S + 7;
    static constexpr int KS1 = (F + 1) / 2;      // k-steps of layer 1 (two features per MFMA)
    static constexpr int FP = 2 * KS1;
    // flat fp32 parameter vector, in the order of a2c.ActorCritic.parameters(): body pi (W1 [64][F], b1, W2 [64][64], b2),
    // body vf (same), action head W [5][64], b [5], value head W [1][64], b [1]
    static constexpr int BODY = MLP_H * F + MLP_H + MLP_H * MLP_H + MLP_H;
    static constexpr int O_PI = 0, O_VF = BODY, O_AW = 2 * BODY, O_AB = O_AW + MLP_NA * MLP_H, O_VW = O_AB + MLP_NA, O_VB = O_VW + MLP_H;
    static constexpr int P = O_VB + 1;
    // LDS image of ONE net (body + its head) in MFMA A-operand order, in floats
    // (the head is evaluated on the VALU: [lane half][output, padded to 8][my 32 units in register order], then 8 biases)
    static constexpr int L_W1 = 0, L_B1 = L_W1 + 2 * KS1 * 64, L_W2 = L_B1 + 64, L_B2 = L_W2 + 2 * 32 * 64, L_WH = L_B2 + 64, L_BH = L_WH + 2 * 8 * 32;
    static constexpr int L_END = L_BH + 8;
};

// unit row of a 32x32 MFMA result held in register r of a lane of half h (lane >> 5)
EWN_DEV constexpr int mlp_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// ... and the column (unit of the previous layer) the k-step `ks` of a 64-wide layer takes from lane half h: register ks & 15 of tile ks >> 4
EWN_DEV constexpr int mlp_kcol(int ks, int h) { return 32 * (ks >> 4) + mlp_row(ks & 15, h); }

// one net's parameters, PyTorch layout in global memory -> the LDS image.  net: 0 policy (5 logits), 1 value (1 output).
template <int S>
EWN_DEV void mlp_pack_net(float *L, const float *P, int net, int tid, int nthreads)
{
    using G = MlpGeo<S>;
    const float *W1 = P + (net ? G::O_VF : G::O_PI), *b1 = W1 + MLP_H * G::F, *W2 = b1 + MLP_H, *b2 = W2 + MLP_H * MLP_H;
    const float *Wh = P + (net ? G::O_VW : G::O_AW), *bh = P + (net ? G::O_VB : G::O_AB);
    const int nout = net ? 1 : MLP_NA;
    for (int e = tid; e < 2 * G::KS1 * 64; e += nthreads) {       // [tile][k-step][lane]: W1[32 tile + (lane & 31)][2 s + (lane >> 5)]
        const int l = e & 63, s = (e >> 6) % G::KS1, mt = (e >> 6) / G::KS1, k = 2 * s + (l >> 5);
        L[G::L_W1 + e] = k < G::F ? W1[(mt * 32 + (l & 31)) * G::F + k] : 0.0f;
    }
    for (int e = tid; e < 2 * 32 * 64; e += nthreads) {           // [tile][k-step][lane]: W2[32 tile + (lane & 31)][kcol(ks, lane >> 5)]
        const int l = e & 63, ks = (e >> 6) & 31, mt = e >> 11;
        L[G::L_W2 + e] = W2[(mt * 32 + (l & 31)) * MLP_H + mlp_kcol(ks, l >> 5)];
    }
    for (int e = tid; e < 2 * 8 * 32; e += nthreads) {            // the head: [half][output][tile * 16 + register] = Wh[output][unit held there]
        const int q = e & 31, a = (e >> 5) & 7, hh = e >> 8;
        L[G::L_WH + e] = a < nout ? Wh[a * MLP_H + 32 * (q >> 4) + mlp_row(q & 15, hh)] : 0.0f;
    }
    for (int e = tid; e < MLP_H; e += nthreads) { L[G::L_B1 + e] = b1[e]; L[G::L_B2 + e] = b2[e]; }
    for (int e = tid; e < 8; e += nthreads) L[G::L_BH + e] = e < nout ? bh[e] : 0.0f;
}

// accumulator initialised with the bias: register r <- b[mlp_row(r, h)]; rows 8g + 4h .. + 3 are one 16-byte read
EWN_DEV f32x16 mlp_bias_acc(const float *b, int h)
{
    f32x16 a;
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 v = *(const float4 *)(b + 8 * g + 4 * h);
        a[4 * g] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
    }
    return a;
}

// tanh(x) = 1 - 2 / (exp(2x) + 1): one v_exp_f32, one v_rcp_f32; absolute error ~1e-7 (saturates cleanly to +-1)
EWN_DEV float mlp_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

EWN_DEV f32x16 mlp_tanh16(f32x16 a)
{
    #pragma unroll
    for (int i = 0; i < 16; i++) a[i] = mlp_tanh(a[i]);
    return a;
}

#define MLP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MLP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// value of x in the lane of the other half with the same sample (lane ^ 32)
EWN_DEV float mlp_other_half(float x, int lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane ^ 32) & 63) << 2, __float_as_int(x)));
}

// One net's forward pass on a tile of 32 samples.  L: the net's LDS image; xb(s) = this lane's B operand of layer-1 k-step s,
// i.e. feature 2 s + (lane >> 5) of sample lane & 31.  h1 / h2: the activations in MFMA layout [tile][register] (tanh applied).
// out[a] (a < NOUT): the head's outputs for sample lane & 31, the same numbers in both lane halves.
//
// The two bodies' 64-wide layers are MFMA chains (two independent accumulator tiles, alternating).  The head -- 5 or 1 rows -- is
// NOT: as a 32-row MFMA tile it cost 32 instructions x 64 cycles for 5/32 (1/32) useful rows, a quarter of the forward pass; on the
// VALU it is NOUT x 32 FMAs per lane on the units the lane already holds plus one exchange between the lane halves.  Measured
// (tools/mfma_probe.hip): the f32-input MFMA does not overlap with VALU work on its SIMD -- not of its own wave, hardly of the other
// wave -- so matrix cycles and VALU issue slots simply add up, and every MFMA that computes padding is pure loss.
template <int S, int NOUT, class XB>
EWN_DEV void mlp_forward(const float *L, int lane, XB xb, f32x16 (&h1)[2], f32x16 (&h2)[2], float (&out)[NOUT])
{
    using G = MlpGeo<S>;
    const int h = lane >> 5;
    f32x16 a0 = mlp_bias_acc(L + G::L_B1, h), a1 = mlp_bias_acc(L + G::L_B1 + 32, h);
    #pragma unroll
    for (int s = 0; s < G::KS1; s++) {
        const float b = xb(s);
        a0 = MLP_MFMA(L[G::L_W1 + s * 64 + lane], b, a0);
        a1 = MLP_MFMA(L[G::L_W1 + (G::KS1 + s) * 64 + lane], b, a1);
        if ((s & 7) == 7) MLP_SCHED_FENCE();   // at most eight k-steps' operand reads in flight: the fully unrolled loop otherwise hoists them all
    }
    h1[0] = mlp_tanh16(a0); h1[1] = mlp_tanh16(a1);
    f32x16 c0 = mlp_bias_acc(L + G::L_B2, h), c1 = mlp_bias_acc(L + G::L_B2 + 32, h);
    #pragma unroll
    for (int ks = 0; ks < 32; ks++) {
        const float b = h1[ks >> 4][ks & 15];
        c0 = MLP_MFMA(L[G::L_W2 + ks * 64 + lane], b, c0);
        c1 = MLP_MFMA(L[G::L_W2 + (32 + ks) * 64 + lane], b, c1);
        if ((ks & 7) == 7) MLP_SCHED_FENCE();
    }
    h2[0] = mlp_tanh16(c0); h2[1] = mlp_tanh16(c1);
    // the head on the VALU: my half's 32 units, then the other half's partial sum
    const float *wh = L + G::L_WH + h * 8 * 32;
    #pragma unroll
    for (int a = 0; a < NOUT; a++) {
        float acc = 0.0f;
        #pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 w = *(const float4 *)(wh + a * 32 + 4 * q4);
            const int mt = q4 >> 2, r = (4 * q4) & 15;
            acc = fmaf(w.x, h2[mt][r], acc); acc = fmaf(w.y, h2[mt][r + 1], acc); acc = fmaf(w.z, h2[mt][r + 2], acc); acc = fmaf(w.w, h2[mt][r + 3], acc);
        }
        out[a] = acc;
    }
    #pragma unroll
    for (int a = 0; a < NOUT; a++) {
        const float o = mlp_other_half(out[a], lane);
        out[a] = (h ? o + out[a] : out[a] + o) + L[G::L_BH + a];   // half 0's sum first, in both halves: bit-identical results
    }
}

