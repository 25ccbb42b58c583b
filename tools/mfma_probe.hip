// mfma_probe.hip -- what a wave can issue under a running v_mfma_f32_32x32x2_f32 on gfx950 (measurement tool, not product code;
// tools/mfma_probe.py drives it).  The fused A2C gradient kernels run one wave per SIMD and interleave MFMA chains with VALU / LDS
// work: this measures cycles per MFMA for a dependent chain, for two interleaved chains, and with N independent VALU or LDS-read
// "filler" instructions behind every MFMA, at one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define MF(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MB(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a8), "v"(b8))
#define MBA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a8), "v"(b8))
#define VP(x) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(a))
#define VK(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define VF(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(a))
#define LR(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(ldsaddr))
#define VI(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a))
#define VE(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define VC(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc")

template <int MODE, int NF>
__global__ void probe(float *out, u64 *cyc, int reps)
{
    __shared__ float sm[4096];
    f32x16 c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = 0.0f; c1[i] = 1.0f; }
    float a = 1.0f + threadIdx.x * 1e-9f, b = 0.5f;
    typedef short bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 a8, b8;
    for (int i = 0; i < 8; i++) { a8[i] = (short)(0x3f80 + threadIdx.x % 3); b8[i] = (short)0x3f00; }
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = (float)i;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 d2[8];
    for (int i = 0; i < 8; i++) { d2[i][0] = (float)i; d2[i][1] = 1.0f; }
    sm[threadIdx.x] = 1.0f;
    __syncthreads();
    const unsigned ldsaddr = (threadIdx.x & 63) * 4;
    u64 t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < reps; r++) {
        #pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) MF(c0);                           // one dependent chain
            if (MODE == 1) { if (k & 1) MF(c1); else MF(c0); } // two chains, alternating
            if (MODE == 2) { MF(c0); }                       // chain + VALU fillers
            if (MODE == 3) { if (k & 1) MF(c1); else MF(c0); }
            if (MODE == 4) { }                               // fillers only
            if (MODE == 5) { MF(c0); }                       // chain + LDS-read fillers
            if (MODE == 2 || MODE == 3 || MODE == 4) {
                #pragma unroll
                for (int i = 0; i < NF; i++) VF(f[i & 15]);
            }
            if (MODE == 5) {
                #pragma unroll
                for (int i = 0; i < NF; i++) LR(f[i & 15]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (MODE == 6 || MODE == 7) { if (MODE == 6) MF(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VI(f[i & 15]); }   // integer VALU fillers
            if (MODE == 8 || MODE == 9) { if (MODE == 8) MF(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VE(f[i & 15]); }   // transcendental fillers
            if (MODE == 11) MB(c0);                                                      // bf16 32x32x16: dependent chain
            if (MODE == 12) { if (k & 1) MB(c1); else MB(c0); }                          // ... two chains
            if (MODE == 13) { MB(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VF(f[i & 15]); }   // ... chain + fp32 VALU fillers
            if (MODE == 14) { MB(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VI(f[i & 15]); }   // ... chain + integer VALU fillers
            if (MODE == 15) { if (k & 1) MB(c1); else MB(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VI(f[i & 15]); }
            if (MODE == 16) { MBA(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VI(f[i & 15]); }   // accumulator in AGPRs + integer VALU
            if (MODE == 17) { MB(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VP(f[i & 15]); }    // v_perm_b32 fillers
            if (MODE == 18) { MB(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VK(d2[i & 7], d2[(i + 1) & 7]); }   // v_pk_add_f32 fillers
            if (MODE == 19) { _Pragma("unroll") for (int i = 0; i < NF; i++) VK(d2[i & 7], d2[(i + 1) & 7]); }
            if (MODE == 20) { _Pragma("unroll") for (int i = 0; i < NF; i++) VP(f[i & 15]); }
            if (MODE == 10) { MF(c0); _Pragma("unroll") for (int i = 0; i < NF; i++) VC(f[i & 15]); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.0f;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + f[i] + d2[i & 7][i >> 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NF>
static void run(const char *name, int waves_per_simd, int reps, float *out, u64 *cyc)
{
    const int threads = 256 * waves_per_simd, blocks = 256;   // one block per CU
    probe<MODE, NF><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    probe<MODE, NF><<<blocks, threads>>>(out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<u64> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(u64), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // s_memtime ticks at the shader clock (MI355X_MICROARCH.md constants table): cycles per slot = one MFMA (if any) + its NF fillers,
    // as seen by one wave; with two waves per SIMD both waves' slots share the SIMD, so the SIMD spends half of that per slot
    printf("%-34s waves/SIMD %d  NF %2d : %7.1f cycles per slot and wave  (%6.1f per slot and SIMD)\n", name, waves_per_simd, NF,
           med / reps / 8.0, med / reps / 8.0 / waves_per_simd);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    float *out; u64 *cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * 8 * sizeof(u64));
    for (int w = 1; w <= 2; w++) {
        run<0, 0>("dependent chain", w, reps, out, cyc);
        run<1, 0>("two alternating chains", w, reps, out, cyc);
        run<2, 4>("chain + 4 VALU behind each", w, reps, out, cyc);
        run<2, 8>("chain + 8 VALU", w, reps, out, cyc);
        run<2, 12>("chain + 12 VALU", w, reps, out, cyc);
        run<2, 16>("chain + 16 VALU", w, reps, out, cyc);
        run<3, 8>("two chains + 8 VALU", w, reps, out, cyc);
        run<3, 12>("two chains + 12 VALU", w, reps, out, cyc);
        run<4, 8>("8 VALU only (no MFMA)", w, reps, out, cyc);
        run<4, 16>("16 VALU only", w, reps, out, cyc);
        run<6, 8>("chain + 8 v_and_b32", w, reps, out, cyc);
        run<6, 12>("chain + 12 v_and_b32", w, reps, out, cyc);
        run<7, 8>("8 v_and_b32 only", w, reps, out, cyc);
        run<8, 4>("chain + 4 v_exp_f32", w, reps, out, cyc);
        run<9, 4>("4 v_exp_f32 only", w, reps, out, cyc);
        run<10, 8>("chain + 8 v_cndmask", w, reps, out, cyc);
        run<11, 0>("bf16 32x32x16 dependent chain", w, reps, out, cyc);
        run<12, 0>("bf16 two alternating chains", w, reps, out, cyc);
        run<13, 4>("bf16 chain + 4 v_fma_f32", w, reps, out, cyc);
        run<13, 8>("bf16 chain + 8 v_fma_f32", w, reps, out, cyc);
        run<14, 4>("bf16 chain + 4 v_and_b32", w, reps, out, cyc);
        run<14, 8>("bf16 chain + 8 v_and_b32", w, reps, out, cyc);
        run<14, 16>("bf16 chain + 16 v_and_b32", w, reps, out, cyc);
        run<15, 8>("bf16 two chains + 8 v_and_b32", w, reps, out, cyc);
        run<16, 8>("bf16 chain (AGPR acc) + 8 v_and_b32", w, reps, out, cyc);
        run<16, 16>("bf16 chain (AGPR acc) + 16 v_and_b32", w, reps, out, cyc);
        run<17, 8>("bf16 chain + 8 v_perm_b32", w, reps, out, cyc);
        run<20, 8>("8 v_perm_b32 only", w, reps, out, cyc);
        run<18, 8>("bf16 chain + 8 v_pk_add_f32", w, reps, out, cyc);
        run<19, 8>("8 v_pk_add_f32 only", w, reps, out, cyc);
        run<5, 2>("chain + 2 ds_read_b32 + wait", w, reps, out, cyc);
        run<5, 4>("chain + 4 ds_read_b32 + wait", w, reps, out, cyc);
    }
    return 0;
}
