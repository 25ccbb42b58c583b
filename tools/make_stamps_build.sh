#!/bin/bash
# Diagnostic build (never shipped): s_memtime stamps at the phase boundaries of k_step_d3, written over the reward buffer.
set -e
rm -rf /tmp/abl && mkdir -p /tmp/abl && cp -r /root/repo/ewn_gym_amd/csrc /tmp/abl/ && cd /tmp/abl/csrc
python3 - <<'PY'
p='ewn_kernels.hip'; s=open(p).read()
s=s.replace('#include "../../include/ewn_hip.h"','#include "/root/repo/include/ewn_hip.h"')
open(p,'w').write(s)
p='ewn_step_d3.hpp'; s=open(p).read()
stamp='''#define STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[i] = t_; } while (0)
'''
n0=s.count("STAMP(")
s=s.replace("template <int S, int T, int OPP, int RNGK>\n__global__ __launch_bounds__(D3_BS) void k_step_d3(D3Cfg c, D3Buf B)\n{", stamp+"template <int S, int T, int OPP, int RNGK>\n__global__ __launch_bounds__(D3_BS) void k_step_d3(D3Cfg c, D3Buf B)\n{\n    unsigned long long stamps[8]; STAMP(0);")
s=s.replace("    const int8_t *gsrc = B.board + (size_t)g0 * CELLS;","    STAMP(1);\n    const int8_t *gsrc = B.board + (size_t)g0 * CELLS;")
s=s.replace("    } else block_copy_in(lds, gsrc, nbytes);\n    __syncthreads();\n","    } else block_copy_in(lds, gsrc, nbytes);\n    __syncthreads();\n    STAMP(2);\n")
s=s.replace("    const bool active = live && !frozen;","    STAMP(7);\n    const bool active = live && !frozen;")
s=s.replace("    int oflag = 0, odir = 0;\n    if constexpr (OPP == 0) d3_search<S, T>(Tb, s, dice, sub, oflag, odir);","    STAMP(3);\n    int oflag = 0, odir = 0;\n    if constexpr (OPP == 0) d3_search<S, T>(Tb, s, dice, sub, oflag, odir);\n    STAMP(4);")
s=s.replace("    __syncthreads();\n    block_copy_out(B.board + (size_t)g0 * CELLS, lds, ng * CELLS);\n    if (B.tboard) block_copy_out(B.tboard + (size_t)g0 * CELLS, lds_t, ng * CELLS);\n","    STAMP(5);\n    __syncthreads();\n    block_copy_out(B.board + (size_t)g0 * CELLS, lds, ng * CELLS);\n    STAMP(6);\n    if (threadIdx.x == 0 && B.tboard) for (int i = 0; i < 8; i++) ((unsigned long long *)B.tboard)[blockIdx.x * 8 + i] = stamps[i];\n")
print("stamps inserted:", s.count("STAMP(")-n0)
open(p,'w').write(s)
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -o /root/repo/ewn_gym_amd/lib/libewn_hip_stamps.so ewn_kernels.hip
