// ewn_lds.hpp -- block-wide global <-> LDS staging helpers shared by the step kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BS 256

// ---------------------------------------------------------------- LDS staging

// Copy nbytes between global and LDS with the whole block: 16-byte vectors when both
// sides allow it (consecutive threads -> consecutive 16-byte pieces), then dwords, then
// the byte tail.  All loads of a pass are issued before the first LDS store.
__device__ __forceinline__ void block_copy_in(int8_t *lds, const int8_t *g, int nbytes)
{
    const int nq = (((uintptr_t)g & 15) == 0) ? nbytes >> 4 : 0;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) ((uint4 *)lds)[i] = ((const uint4 *)g)[i];
    const int w0 = nq << 2, nw = (((uintptr_t)g & 3) == 0) ? nbytes >> 2 : w0;
    for (int i = w0 + threadIdx.x; i < nw; i += blockDim.x) ((u32 *)lds)[i] = ((const u32 *)g)[i];
    for (int i = (nw << 2) + threadIdx.x; i < nbytes; i += blockDim.x) lds[i] = g[i];
}

__device__ __forceinline__ void block_copy_out(int8_t *g, const int8_t *lds, int nbytes)
{
    const int nq = (((uintptr_t)g & 15) == 0) ? nbytes >> 4 : 0;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) ((uint4 *)g)[i] = ((const uint4 *)lds)[i];
    const int w0 = nq << 2, nw = (((uintptr_t)g & 3) == 0) ? nbytes >> 2 : w0;
    for (int i = w0 + threadIdx.x; i < nw; i += blockDim.x) ((u32 *)g)[i] = ((const u32 *)lds)[i];
    for (int i = (nw << 2) + threadIdx.x; i < nbytes; i += blockDim.x) g[i] = lds[i];
}

// Search tables, global -> LDS with the LDS-DMA form of the load (global_load_lds_dwordx4):
// no VGPR staging and nothing waits on it until the barrier in front of the search.
// BYTES is a multiple of 4096 (256 threads x 16 B); src 16-byte aligned; dst = wave-uniform
// base + lane*16, which is exactly the linear image we want.
template <int BYTES>
__device__ __forceinline__ void tables_to_lds(int8_t *lds, const int8_t *g)
{
    static_assert(BYTES % (BS * 16) == 0, "table size must be padded to 4 KiB");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    #pragma unroll
    for (int cnk = 0; cnk < BYTES / (BS * 16); cnk++) {
        const int off = (cnk * (BS / 64) + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
    }
}

// The LDS-DMA loads above are outstanding vector-memory operations of the wave that issued them; a barrier does not wait for
// them.  Every wave must drain its own before the block barrier behind which OTHER waves read the chunks it fetched -- a
// wave with no other load to wait for (all its lanes past the end of the batch, or simply faster) otherwise lets the rest
// of the block read whatever the previous kernel left in that part of LDS (seen as wrong moves on one board size, only
// after kernels for other board sizes had run on the same CUs).
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

