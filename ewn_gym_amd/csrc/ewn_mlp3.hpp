// ewn_mlp3.hpp -- the actor-critic's matrix products on the bf16 matrix pipe at fp32 accuracy ("bf16 x 3").
//
// Why not the f32-input MFMA (ewn_mlp.hpp): measured (tools/mfma_probe.hip) v_mfma_f32_32x32x2_f32 takes 64 cycles for K = 2 and
// overlaps with NOTHING on its SIMD (it runs on the fp32 vector ALUs), while v_mfma_f32_32x32x16_bf16 takes 32 cycles for K = 16
// and runs on the matrix pipe beside the VALU: sixteen times the multiply-adds per cycle, and free issue slots on top.  An fp32
// number is EXACTLY the sum of three bf16 numbers (8 significant bits each: hi = the top 16 bits of x, mid = the top 16 bits of
// x - hi, lo = x - hi - mid, all three subtractions exact), every bf16 x bf16 product is exact in fp32 (8 x 8 = 16 bits), and the
// matrix pipe accumulates in fp32.  So  x w = (x0 + x1 + x2)(w0 + w1 + w2)  evaluated as the six products
//   x0 w0 + x0 w1 + x1 w0 + x0 w2 + x2 w0 + x1 w1
// leaves out only terms below 2^-24 |x w| -- the size of ONE fp32 rounding of that product, which the fp32 chain it replaces
// commits as well.  Six MFMAs of 32 cycles for K = 16 against eight of 64 cycles: 2.7 x fewer matrix cycles, none of them taken
// from the VALU.  Inputs that are small integers (board cells, one-hot dice) are exact in ONE bf16: three products.
//
// Operand geometry of v_mfma_f32_32x32x16_bf16 (D = A B + C; A 32 x 16, B 16 x 32): lane l holds row (A) / column (B) l & 31 and the
// eight k-slots (h, jj), h = l >> 5, jj = 0 .. 7, as 8 bf16 in four registers; slot (h, jj) of A meets slot (h, jj) of B.  WHICH k a
// slot stands for is ours to choose, as long as both operands agree: a "k-block" kb is sixteen k values, U(kb, h, jj) below for the
// 64 hidden units, 16 kb + 8 h + jj for the features.  The 32 x 32 result has its column on the lane and row mlp_row(r, h) in
// register r (as the f32 MFMA: ewn_mlp.hpp) -- so eight consecutive registers of a result ARE the eight slots of a k-block of the
// next product, after the split: layers chain through registers, no LDS round trip, no lane movement.
#pragma once
#include "ewn_mlp.hpp"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MLP3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// hidden unit in slot (h, jj) of k-block kb (0 .. 3): register 8 (kb & 1) + jj of result tile kb >> 1, as held by lane half h
EWN_DEV constexpr int mlp3_unit(int kb, int h, int jj) { return 32 * (kb >> 1) + mlp_row(8 * (kb & 1) + jj, h); }

template <int S> struct Mlp3Geo {
    using G = MlpGeo<S>;
    static constexpr int F = G::F, KB1 = (F + 15) / 16;            // k-blocks of layer 1 (feature 16 kb + 8 h + jj; past F: zero weights)
    // LDS image of one net, bytes.  Weight operands: [part][row tile][k-block][lane] x 16 bytes
    static constexpr int O_W1 = 0, N_W1 = 2 * KB1 * 64;           // entries per part
    static constexpr int O_W2 = O_W1 + 3 * N_W1 * 16, N_W2 = 2 * 4 * 64;
    static constexpr int O_F = O_W2 + 3 * N_W2 * 16;               // floats: b1 [64], b2 [64], head W for the VALU [2][8][32], head b [8]
    static constexpr int F_B1 = 0, F_B2 = 64, F_WH = 128, F_BH = F_WH + 2 * 8 * 32, F_END = F_BH + 8;
    static constexpr int FWD_BYTES = O_F + F_END * 4;              // what the forward pass needs (k_rollout_mlp)
};

// x = p0 + p1 + p2, each the bf16 in the TOP half of the returned word (the low halves are ignored by mlp3_pack)
EWN_DEV void mlp3_split(float x, u32 &p0, u32 &p1, u32 &p2)
{
    p0 = __float_as_uint(x);
    const float r1 = x - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = __float_as_uint(r2);
}
// top halves of two words -> one register of two bf16 (a in the low half = the lower slot)
EWN_DEV u32 mlp3_pack(u32 a, u32 b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Mlp3Op { u32x4 p[3]; };             // one k-block of a three-part operand

// eight fp32 values -> the three operand parts: per value two ANDs and two subtractions, per pair and part one v_perm_b32 that picks
// the top halves straight into an operand register (5.5 instructions per value).  NOT the packed-fp32 subtraction (v_pk_add_f32, two
// values per issue slot): measured (tools/mfma_probe.hip) packed fp32 arithmetic does not run beside the matrix pipe -- eight of them
// behind a bf16 MFMA take 86 cycles against 46 for eight v_and_b32 -- and running beside it is the whole point (the translation
// unit is built with -fno-slp-vectorize for the same reason)
EWN_DEV Mlp3Op mlp3_operand(const float (&v)[8])
{
    u32 q[3][8];
    #pragma unroll
    for (int i = 0; i < 8; i++) mlp3_split(v[i], q[0][i], q[1][i], q[2][i]);
    Mlp3Op o;
    #pragma unroll
    for (int p = 0; p < 3; p++) {
        o.p[p][0] = mlp3_pack(q[p][0], q[p][1]); o.p[p][1] = mlp3_pack(q[p][2], q[p][3]);
        o.p[p][2] = mlp3_pack(q[p][4], q[p][5]); o.p[p][3] = mlp3_pack(q[p][6], q[p][7]);
    }
    return o;
}
// registers 8 c .. 8 c + 7 of a result tile
EWN_DEV Mlp3Op mlp3_operand(const f32x16 &t, int c)
{
    const float v[8] = { t[8 * c], t[8 * c + 1], t[8 * c + 2], t[8 * c + 3], t[8 * c + 4], t[8 * c + 5], t[8 * c + 6], t[8 * c + 7] };
    return mlp3_operand(v);
}

// acc += A B at fp32 accuracy: the six products, smallest first
EWN_DEV f32x16 mlp3_mac(f32x16 acc, const Mlp3Op &a, const Mlp3Op &b)
{
    acc = MLP3_MFMA(a.p[1], b.p[1], acc);
    acc = MLP3_MFMA(a.p[2], b.p[0], acc);
    acc = MLP3_MFMA(a.p[0], b.p[2], acc);
    acc = MLP3_MFMA(a.p[1], b.p[0], acc);
    acc = MLP3_MFMA(a.p[0], b.p[1], acc);
    acc = MLP3_MFMA(a.p[0], b.p[0], acc);
    return acc;
}
// ... when one operand is exact in a single bf16 (features)
EWN_DEV f32x16 mlp3_mac_ax(f32x16 acc, const Mlp3Op &a, u32x4 x)
{
    acc = MLP3_MFMA(a.p[2], x, acc); acc = MLP3_MFMA(a.p[1], x, acc); acc = MLP3_MFMA(a.p[0], x, acc);
    return acc;
}
EWN_DEV f32x16 mlp3_mac_xb(f32x16 acc, u32x4 x, const Mlp3Op &b)
{
    acc = MLP3_MFMA(x, b.p[2], acc); acc = MLP3_MFMA(x, b.p[1], acc); acc = MLP3_MFMA(x, b.p[0], acc);
    return acc;
}

// a weight operand out of an image: [part][n entries]
EWN_DEV Mlp3Op mlp3_load(const u32x4 *img, int n, int idx)
{
    Mlp3Op o;
    o.p[0] = img[idx]; o.p[1] = img[n + idx]; o.p[2] = img[2 * n + idx];
    return o;
}
EWN_DEV void mlp3_store(u32x4 *img, int n, int idx, const float (&v)[8])
{
    const Mlp3Op o = mlp3_operand(v);
    img[idx] = o.p[0]; img[n + idx] = o.p[1]; img[2 * n + idx] = o.p[2];
}

// eight board / record bytes (signed cells) -> one operand of eight bf16 (small integers: exact); the dice one-hot is OR-ed in by
// the caller (mlp3_onehot): bytes past the board are zero in a slot / must be masked by the caller in a record
EWN_DEV u32x4 mlp3_bytes_operand(u32 lo, u32 hi)
{
    const u32 w[2] = { lo, hi };
    u32x4 o;
    #pragma unroll
    for (int d = 0; d < 4; d++) {
        const float f0 = (float)(int)(int8_t)((w[d >> 1] >> (16 * (d & 1))) & 0xFFu), f1 = (float)(int)(int8_t)((w[d >> 1] >> (16 * (d & 1) + 8)) & 0xFFu);
        o[d] = mlp3_pack(__float_as_uint(f0), __float_as_uint(f1));
    }
    return o;
}
// feature `first + q` (q = 0 .. 7 are this operand's slots) set to 1.0 when q == hot
EWN_DEV u32x4 mlp3_onehot(u32x4 o, int hot)
{
    #pragma unroll
    for (int d = 0; d < 4; d++) o[d] |= (u32)((hot >> 1) == d) * (0x3F80u << (16 * (hot & 1)));
    return o;
}

// one net's forward parameters, PyTorch layout in global memory -> the LDS image.  net: 0 policy (5 logits), 1 value (1 output).
template <int S>
EWN_DEV void mlp3_pack_fwd(int8_t *img, const float *P, int net, int tid, int nthreads)
{
    using G = MlpGeo<S>;
    using Q = Mlp3Geo<S>;
    const float *W1 = P + (net ? G::O_VF : G::O_PI), *b1 = W1 + MLP_H * G::F, *W2 = b1 + MLP_H, *b2 = W2 + MLP_H * MLP_H;
    const float *Wh = P + (net ? G::O_VW : G::O_AW), *bh = P + (net ? G::O_VB : G::O_AB);
    const int nout = net ? 1 : MLP_NA;
    u32x4 *I1 = (u32x4 *)(img + Q::O_W1), *I2 = (u32x4 *)(img + Q::O_W2);
    float *Lf = (float *)(img + Q::O_F);
    for (int e = tid; e < Q::N_W1; e += nthreads) {               // [tile][k-block][lane]: W1[32 tile + (lane & 31)][16 kb + 8 h + jj]
        const int l = e & 63, kb = (e >> 6) % Q::KB1, mt = (e >> 6) / Q::KB1;
        float v[8];
        #pragma unroll
        for (int jj = 0; jj < 8; jj++) { const int f = 16 * kb + 8 * (l >> 5) + jj; v[jj] = f < G::F ? W1[(mt * 32 + (l & 31)) * G::F + f] : 0.0f; }
        mlp3_store(I1, Q::N_W1, e, v);
    }
    for (int e = tid; e < Q::N_W2; e += nthreads) {               // [tile][k-block][lane]: W2[32 tile + (lane & 31)][unit(kb, h, jj)]
        const int l = e & 63, kb = (e >> 6) & 3, mt = e >> 8;
        float v[8];
        #pragma unroll
        for (int jj = 0; jj < 8; jj++) v[jj] = W2[(mt * 32 + (l & 31)) * MLP_H + mlp3_unit(kb, l >> 5, jj)];
        mlp3_store(I2, Q::N_W2, e, v);
    }
    for (int e = tid; e < 2 * 8 * 32; e += nthreads) {            // the VALU head: [half][output][tile * 16 + register] = Wh[output][unit held there]
        const int q = e & 31, a = (e >> 5) & 7, hh = e >> 8;
        Lf[Q::F_WH + e] = a < nout ? Wh[a * MLP_H + 32 * (q >> 4) + mlp_row(q & 15, hh)] : 0.0f;
    }
    for (int e = tid; e < MLP_H; e += nthreads) { Lf[Q::F_B1 + e] = b1[e]; Lf[Q::F_B2 + e] = b2[e]; }
    for (int e = tid; e < 8; e += nthreads) Lf[Q::F_BH + e] = e < nout ? bh[e] : 0.0f;
}

// the two 64-wide layers: xb(kb) = this lane's feature operand of k-block kb (features 16 kb + 8 h + jj of sample lane & 31)
template <int S, class XB>
EWN_DEV void mlp3_body(const int8_t *img, int lane, XB xb, f32x16 (&h1)[2], f32x16 (&h2)[2])
{
    using Q = Mlp3Geo<S>;
    const u32x4 *I1 = (const u32x4 *)(img + Q::O_W1), *I2 = (const u32x4 *)(img + Q::O_W2);
    const float *Lf = (const float *)(img + Q::O_F);
    const int h = lane >> 5;
    f32x16 a0 = mlp_bias_acc(Lf + Q::F_B1, h), a1 = mlp_bias_acc(Lf + Q::F_B1 + 32, h);
    #pragma unroll
    for (int kb = 0; kb < Q::KB1; kb++) {
        const u32x4 x = xb(kb);
        a0 = mlp3_mac_ax(a0, mlp3_load(I1, Q::N_W1, kb * 64 + lane), x);
        a1 = mlp3_mac_ax(a1, mlp3_load(I1, Q::N_W1, (Q::KB1 + kb) * 64 + lane), x);
    }
    h1[0] = mlp_tanh16(a0); h1[1] = mlp_tanh16(a1);
    f32x16 c0 = mlp_bias_acc(Lf + Q::F_B2, h), c1 = mlp_bias_acc(Lf + Q::F_B2 + 32, h);
    #pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        const Mlp3Op x = mlp3_operand(h1[kb >> 1], kb & 1);
        c0 = mlp3_mac(c0, mlp3_load(I2, Q::N_W2, kb * 64 + lane), x);
        c1 = mlp3_mac(c1, mlp3_load(I2, Q::N_W2, (4 + kb) * 64 + lane), x);
    }
    h2[0] = mlp_tanh16(c0); h2[1] = mlp_tanh16(c1);
}

// the head on the VALU (5 or 1 rows: a 32-row MFMA tile would be padding): out[a] for sample lane & 31, the same bits in both halves
template <int S, int NOUT>
EWN_DEV void mlp3_head(const int8_t *img, int lane, const f32x16 (&h2)[2], float (&out)[NOUT])
{
    using Q = Mlp3Geo<S>;
    const float *Lf = (const float *)(img + Q::O_F);
    const int h = lane >> 5;
    const float *wh = Lf + Q::F_WH + h * 8 * 32;
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
        out[a] = (h ? o + out[a] : out[a] + o) + Lf[Q::F_BH + a];   // half 0's sum first, in both halves
    }
}

template <int S, int NOUT, class XB>
EWN_DEV void mlp3_forward(const int8_t *img, int lane, XB xb, f32x16 (&h1)[2], f32x16 (&h2)[2], float (&out)[NOUT])
{
    mlp3_body<S>(img, lane, xb, h1, h2);
    mlp3_head<S, NOUT>(img, lane, h2, out);
}
