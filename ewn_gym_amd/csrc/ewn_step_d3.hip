// ewn_step_d3.hip -- translation unit of the lean table-driven step kernel's instantiations (k_step_d3, ewn_step_d3.hpp):
// one launch = one env step.  Split from ewn_kernels.hip so the units compile in parallel.
#include "ewn_host.hpp"
#include "ewn_lds.hpp"
#include "ewn_step_d3.hpp"

// one k_step_d3 instance; a dynamic-LDS request above the 64 KB default (the larger boards at one lane per game, the MT19937
// kind's refill area on top) raises the kernel's limit first, once (gfx950: 160 KB per workgroup)
template <int SS, int TT, int OO, int RR>
static int d3_launch_one(dim3 grid, size_t lds, hipStream_t s, const D3Cfg &dc, const D3Buf &db)
{
    auto kern = k_step_d3<SS, TT, OO, RR>;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
            raised = true;
        }
    }
    kern<<<grid, D3_BS, lds, s>>>(dc, db);
    return EWN_OK;
}

int ewn_launch_step_d3(const ewn_config *cfg, const Geom &g, const KCfg &k, const ewn_state *st, const void *tables, const int8_t *actions,
                       const ewn_step_out *out, void *scratch, bool lean_random, bool fused_refill, hipStream_t s)
{
    // leaves are shared among lanes at depth 3-4; depth 5-6 splits its inner dice over two lanes while the chip is not full
    const int T = (lean_random || cfg->max_depth < 3) ? 1 : (cfg->max_depth > 4 ? (k.N <= 131072 && d3_threads_per_game(k.N) != 1 ? 2 : 1) : d3_threads_per_game(k.N));
    const int gpb = D3_BS / T, step_blocks = (k.N + gpb - 1) / gpb;
    // MT kind with auto-reset: window refills are extra blocks of the same launch (needs the caller's scratch).
    // One refill block (one wave) per step block: a lane of it rebuilds at most one window, ~29 k cycles, well inside the
    // step role's ~45 k.  Measured at 65 536 lanes: 256 refill blocks (two regions each, two chains back to back) 38.4 us
    // per launch, 512 blocks 26.5 us.
    const int refill_blocks = fused_refill ? step_blocks : 0;
    D3Cfg dc = { k.N, k.rng_kind, k.autoreset, k.lane_offset, k.depth, refill_blocks, k.seed_stride, k.W, k.reward, k.key,
                 cfg->shaped ? 1 : 0, k.refresh, k.illegal_reward };
    D3Buf db = { st->board, st->dice, st->done, st->rng, tables, actions, out->reward, out->terminated,
                 out->truncated, out->info, out->terminal_board, out->terminal_dice, out->random_action,
                 fused_refill ? scratch : nullptr, cfg->shaped ? st->prev_score : nullptr, cfg->shaped ? st->tolerance : nullptr };
    const dim3 grid((unsigned)(step_blocks + refill_blocks));
    // LDS: boards + terminal boards | tables | decode scatter area | (MT) this block's parked refill requests; a refill block needs (W+1) x 65 words
    size_t l3 = (((size_t)2 * gpb * g.cells + 15) & ~(size_t)15) + (size_t)gpb * 16; // + d3_decode's 16 bytes per game
    if (fused_refill) l3 += 16 + (size_t)2 * gpb * 16;
    const size_t need_refill = fused_refill ? (size_t)(k.W + 1) * 65 * 4 : 0;
#define D3_LDS(SS) (l3 + FAST_TAB_BYTES(SS) > need_refill ? l3 + FAST_TAB_BYTES(SS) : need_refill)
#define D3_LAUNCH(SS, TT, OO) do { if (k.rng_kind == 0) lrc = d3_launch_one<SS, TT, OO, 0>(grid, D3_LDS(SS), s, dc, db); \
                                   else lrc = d3_launch_one<SS, TT, OO, 1>(grid, d3_lds_static<SS, TT, 1>() ? 0 : D3_LDS(SS), s, dc, db); } while (0)
#define D3_BY_T(SS) do { if (lean_random) D3_LAUNCH(SS, 1, 1); else if (cfg->max_depth > 4) { if (T == 2) D3_LAUNCH(SS, 2, 2); else D3_LAUNCH(SS, 1, 2); } else if (T == 1) D3_LAUNCH(SS, 1, 0); else if (T == 2) D3_LAUNCH(SS, 2, 0); else D3_LAUNCH(SS, 4, 0); } while (0)
    int lrc = EWN_OK;
    switch (g.S) {
    case 5: D3_BY_T(5); break;
    case 6: D3_BY_T(6); break;
    case 7: D3_BY_T(7); break;
    default: D3_BY_T(8); break;
    }
    int rc = lrc != EWN_OK ? lrc : launch_status();
    if (rc == EWN_OK && fused_refill) {
        k_mtq_flip<<<1, 64, 0, s>>>((u32 *)scratch);
        rc = launch_status();
    }
    return rc;
}
