// valu_probe.hip -- measures the issue cost of the instruction classes the EWN step kernels are made of, on gfx950, at 1, 2, 4
// and 8 waves per SIMD.  Not product code: a measurement tool (tools/valu_probe.py builds and drives it and writes
// profiles/r02/valu_probe.json, from which bench.py's roofline.valu_issue takes its measured peak).
//
// Method: every wave runs `reps` x (32 instructions of ONE class on 8 independent registers) between two s_memtime reads
// (s_memtime ticks at the shader clock, MI355X_MICROARCH.md constants table); a block is 256 threads = one wave per SIMD of
// its CU, and the requested LDS size makes exactly W blocks fit a CU, so W = waves per SIMD.  While the SIMD's issue port is
// the bottleneck, issue cycles per wave-instruction = wave cycles / instructions / W.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32;

#define F_v_xor_b32(r) "v_xor_b32 " r ", " r ", %8\n"
#define F_v_and_or_b32(r) "v_and_or_b32 " r ", " r ", %8, %8\n"
#define F_v_lshlrev_b32(r) "v_lshlrev_b32 " r ", 1, " r "\n"
#define F_v_add_u32(r) "v_add_u32 " r ", " r ", %8\n"
#define F_v_lshl_or_b32(r) "v_lshl_or_b32 " r ", " r ", 1, %8\n"
#define F_v_xad_u32(r) "v_xad_u32 " r ", " r ", %8, %8\n"
#define F_v_min_u32(r) "v_min_u32 " r ", " r ", %8\n"
#define F_v_perm_b32(r) "v_perm_b32 " r ", " r ", %8, %8\n"
#define F_v_ffbh_u32(r) "v_ffbh_u32 " r ", " r "\n"
#define F_v_bcnt_u32_b32(r) "v_bcnt_u32_b32 " r ", " r ", %8\n"
#define F_v_mad_u32_u24(r) "v_mad_u32_u24 " r ", " r ", %8, %8\n"
#define F_v_mul_lo_u32(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define F_v_mul_hi_u32(r) "v_mul_hi_u32 " r ", " r ", %8\n"
#define F_v_cndmask_b32(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define F_v_cmp_lt_u32(r) "v_cmp_lt_u32 vcc, " r ", %8\n"
#define F_v_mov_b32(r) "v_mov_b32 " r ", %8\n"
#define F_v_mov_b32_dpp(r) "v_mov_b32_dpp " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_v_lshlrev_b64(r) "v_lshlrev_b64 " r ", 1, " r "\n"
#define F_v_lshrrev_b64(r) "v_lshrrev_b64 " r ", 1, " r "\n"
#define F_v_add_f64(r) "v_add_f64 " r ", " r ", %8\n"
#define F_v_mul_f64(r) "v_mul_f64 " r ", " r ", %8\n"
#define F_v_max_f64(r) "v_max_f64 " r ", " r ", %8\n"
#define F_v_cmp_le_f64(r) "v_cmp_le_f64 vcc, " r ", %8\n"
#define F_mix_cmp_cndmask(r) "v_cmp_lt_u32 vcc, " r ", %8\n v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define CLASSES(X) \
    X(v_xor_b32, 32) \
    X(v_and_or_b32, 32) \
    X(v_lshlrev_b32, 32) \
    X(v_add_u32, 32) \
    X(v_lshl_or_b32, 32) \
    X(v_xad_u32, 32) \
    X(v_min_u32, 32) \
    X(v_perm_b32, 32) \
    X(v_ffbh_u32, 32) \
    X(v_bcnt_u32_b32, 32) \
    X(v_mad_u32_u24, 32) \
    X(v_mul_lo_u32, 32) \
    X(v_mul_hi_u32, 32) \
    X(v_cndmask_b32, 32) \
    X(v_cmp_lt_u32, 32) \
    X(v_mov_b32, 32) \
    X(v_mov_b32_dpp, 32) \
    X(v_lshlrev_b64, 64) \
    X(v_lshrrev_b64, 64) \
    X(v_add_f64, -64) \
    X(v_mul_f64, -64) \
    X(v_max_f64, -64) \
    X(v_cmp_le_f64, -64) \
    X(mix_cmp_cndmask, 32)

enum {
#define X(name, w) OP_##name,
    CLASSES(X)
#undef X
    OP_N
};

struct ClassInfo { const char *name; int width; };
static const ClassInfo INFO[OP_N] = {
#define X(name, w) { #name, w },
    CLASSES(X)
#undef X
};

// ONE asm statement per 32 instructions: the compiler puts an s_nop between separate asm statements (4 issue cycles each)
#define B8(F) F("%0") F("%1") F("%2") F("%3") F("%4") F("%5") F("%6") F("%7")
#define B32(F) B8(F) B8(F) B8(F) B8(F)
#define RUN(F, r0, r1, r2, r3, r4, r5, r6, r7, kk) asm volatile(B32(F) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(kk) : "vcc")

template <int OP>
__global__ __launch_bounds__(256) void k_probe(u64 *out, int reps, u32 seed)
{
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u, a4 = a0 ^ 0x55u, a5 = a0 + 77u, a6 = a0 * 11u, a7 = a0 | 0x100u;
    u32 k = seed | 1u;
    u64 b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, dk = 1.0 + seed;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll 1
    for (int r = 0; r < reps; r++) {
#define X(name, w) \
        if constexpr (OP == OP_##name) { \
            if constexpr (w == 32) { RUN(F_##name, a0, a1, a2, a3, a4, a5, a6, a7, k); } \
            else if constexpr (w == 64) { RUN(F_##name, b0, b1, b2, b3, b4, b5, b6, b7, k); } \
            else { RUN(F_##name, d0, d1, d2, d3, d4, d5, d6, d7, dk); } \
        }
        CLASSES(X)
#undef X
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    const u32 sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (u32)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) ^ (u32)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if ((threadIdx.x & 63) == 0) {
        // HW_ID (register 4): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; XCC_ID (register 20)
        const u32 hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        u64 *o = out + 2 + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
        o[0] = t0; o[1] = t1; o[2] = ((u64)xcc << 32) | hw;
    }
    if (sink == 0x12345678u) out[1] = sink; // keeps the chains alive
}

typedef void (*probe_fn)(u64 *, int, u32);
static probe_fn FN[OP_N] = {
#define X(name, w) k_probe<OP_##name>,
    CLASSES(X)
#undef X
};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    const size_t lds_cu = 160 * 1024;
    u64 *dev = nullptr;
    const int max_waves = cus * 8 * 4;
    if (hipMalloc(&dev, (size_t)(2 + 3 * max_waves) * sizeof(u64)) != hipSuccess) return 1;
    std::vector<u64> host(2 + 3 * max_waves);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"reps\": %d, \"insts_per_wave\": %d, \"classes\": {\n", prop.gcnArchName, cus,
           prop.clockRate / 1000, reps, reps * 32);
    for (int op = 0; op < OP_N; op++) {
        printf("  \"%s\": {\"width\": %d", INFO[op].name, INFO[op].width < 0 ? 64 : INFO[op].width);
        for (int W = 1; W <= 8; W *= 2) {
            // W blocks per CU on average: the grid size alone sets the occupancy (every block fits at once: tiny LDS, few
            // registers); the hardware ids each wave records tell how evenly the dispatcher spread them over the SIMDs
            const size_t lds = 1024;
            const int blocks = cus * W;
            hipMemset(dev, 0, (size_t)(2 + 3 * max_waves) * sizeof(u64));
            hipLaunchKernelGGL(FN[op], dim3(blocks), dim3(256), lds, 0, dev, 64, 1u); // warm-up (code cache)
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(FN[op], dim3(blocks), dim3(256), lds, 0, dev, reps, 1u);
            hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const int nw = blocks * 4;
            hipMemcpy(host.data(), dev, (size_t)(2 + 3 * nw) * sizeof(u64), hipMemcpyDeviceToHost);
            std::vector<u64> c(nw);
            u64 tmin = ~0ull, tmax = 0;
            std::vector<int> per_simd(8 * 8 * 2 * 16 * 4 * 2, 0); // xcc, se, sh, cu, simd (generous)
            for (int i = 0; i < nw; i++) {
                const u64 t0 = host[2 + 3 * i], t1 = host[3 + 3 * i], id = host[4 + 3 * i];
                c[i] = t1 - t0; tmin = std::min(tmin, t0); tmax = std::max(tmax, t1);
                const u32 hw = (u32)id, xcc = (u32)(id >> 32) & 7u;
                const u32 simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
                per_simd[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd]++;
            }
            std::sort(c.begin(), c.end());
            int used = 0, wmax = 0;
            for (int v : per_simd) { if (v) used++; wmax = std::max(wmax, v); }
            const double med = (double)c[c.size() / 2], insts = (double)reps * (op == OP_mix_cmp_cndmask ? 64.0 : 32.0);
            // chip-wide rate from the wall clock: the figure the roofline uses (no assumption about the s_memtime tick)
            const double rate = (double)nw * insts / (ms * 1e-3);
            printf(", \"w%d\": {\"wave_ticks_per_inst\": %.3f, \"kernel_ms\": %.4f, \"span_ticks\": %llu, \"simds_used\": %d, \"max_waves_on_a_simd\": %d, "
                   "\"wave_insts_per_s\": %.4g, \"ns_per_inst_per_simd\": %.4f}", W, med / insts, ms, (unsigned long long)(tmax - tmin), used, wmax,
                   rate, 1e9 * 1024.0 / rate);
        }
        printf("}%s\n", op + 1 < OP_N ? "," : "");
    }
    printf("}}\n");
    hipFree(dev);
    return 0;
}
