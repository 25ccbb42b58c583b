// ewn_host.hpp -- host-side pieces shared by the translation units of libewn_hip.so: kernel argument structs, geometry and
// configuration checks, and the launchers each unit exports to the C ABI in ewn_kernels.hip.
#pragma once
#include "ewn_core.hpp"
#include "ewn_fast.hpp"
#include "../../include/ewn_hip.h"
#include <cstdlib>

#define D3_BS 256 // threads per block of the lean step / rollout kernels

struct KCfg {
    int N, opp, depth, heur, rng_kind, shaped, autoreset, refresh, lane_offset, nsim_total;
    u32 seed_stride, W, rng_words;
    double reward, illegal_reward;
    u64 key;
};

struct KState {
    int8_t *board; int8_t *dice; uint8_t *done; u32 *rng; double *prev_score; int32_t *tolerance; const void *tables;
};

struct KOut {
    double *reward; uint8_t *terminated; uint8_t *truncated; uint8_t *info; int8_t *tboard; int8_t *tdice; int8_t *ract;
};

struct KScratch { // split-phase step (MCTS opponent)
    uint8_t *phase; int8_t *cboard; int8_t *cdice; int8_t *act; int32_t *wins; u32 *obs_id;
};

static inline bool make_geom(int S, int L, Geom &g)
{
    if (S < 3 || S > EWN_MAX_BOARD || L < 1 || L >= S - 1) return false; // assert cube_layer < board_size - 1, envs/ewn.py:47
    const int CN = L * (L + 1) / 2;
    if (CN > EWN_MAX_CUBES) return false;
    g.S = S; g.L = L; g.CN = CN; g.cells = S * S;
    g.div_magic = 65536u / (u32)S + 1u;
    g.not_lastcol = g.not_lastrow = g.not_firstcol = g.not_firstrow = 0;
    for (int t = 0; t < 8; t++) g.sq[t] = 0;
    const bool wide = S > 8; // 9x9 .. 11x11: no 64-bit masks, 7-bit positions nine per word (ewn_core.hpp GState<3>)
    if (!wide)
        for (int i = 0; i < S; i++)
            for (int j = 0; j < S; j++) {
                const u64 b = 1ull << (i * S + j);
                if (j < S - 1) g.not_lastcol |= b;
                if (i < S - 1) g.not_lastrow |= b;
                if (j > 0) g.not_firstcol |= b;
                if (i > 0) g.not_firstrow |= b;
                for (int t = 0; t < S; t++) if (i >= t && j >= t) g.sq[t] |= b;
            }
    g.corner_br = wide ? 0 : 1ull << (S * S - 1);
    for (int c = 0; c < 128; c++) g.init[c] = 0;
    int cnt = 1;
    for (int i = 1; i <= L; i++)
        for (int j = 0; j < i; j++) {
            g.init[j * S + (i - j - 1)] = (int8_t)cnt;
            g.init[(S - 1 - j) * S + (S - i + j)] = (int8_t)(-cnt);
            cnt++;
        }
    g.init_occP = g.init_occN = 0; g.init_alive = 0;
    for (int w = 0; w < 3; w++) g.init_posP[w] = g.init_posN[w] = 0;
    const int per = wide ? 9 : 10, bits = wide ? 7 : 6;
    for (int c = 0; c < S * S; c++) {
        const int v = g.init[c];
        if (v > 0) { if (!wide) g.init_occP |= 1ull << c; g.init_alive |= 1u << (v - 1); g.init_posP[(v - 1) / per] |= (u64)c << (bits * ((v - 1) % per)); }
        if (v < 0) { if (!wide) g.init_occN |= 1ull << c; g.init_posN[(-v - 1) / per] |= (u64)c << (bits * ((-v - 1) % per)); }
    }
    return true;
}

static inline int check_cfg(const ewn_config *cfg, Geom &g, KCfg &k)
{
    if (!cfg) return EWN_ENULL;
    if (cfg->board_size > EWN_MAX_BOARD && cfg->cube_layer >= 1 && cfg->cube_layer < cfg->board_size - 1) return EWN_EUNSUPPORTED;
    if (!make_geom(cfg->board_size, cfg->cube_layer, g)) return EWN_EINVAL;
    if (cfg->n_lanes < 1) return EWN_EINVAL;
    if (cfg->opponent_kind < 0 || cfg->opponent_kind > EWN_OPP_MCTS) return EWN_EINVAL;
    if (cfg->rng_kind != EWN_RNG_MT19937 && cfg->rng_kind != EWN_RNG_PHILOX) return EWN_EINVAL;
    if (cfg->opponent_kind == EWN_OPP_MINIMAX) {
        if (cfg->max_depth < 1 || cfg->max_depth > EWN_MAX_DEPTH) return EWN_EUNSUPPORTED;
        if (cfg->heuristic < 0 || cfg->heuristic > EWN_H_SIM_WINRATE) return EWN_EUNSUPPORTED;
        if (cfg->heuristic == EWN_H_SIM_WINRATE && cfg->max_depth > EWN_SIM_WINRATE_MAX_DEPTH) return EWN_EUNSUPPORTED;
    }
    // the searches and rollouts roll dice 1..6 (minimax.py:68, mcts.py:29): cube_num < 6 raises IndexError upstream
    if (cfg->opponent_kind != EWN_OPP_RANDOM && g.CN < 6) return EWN_EUNSUPPORTED;
    if (cfg->opponent_kind == EWN_OPP_MCTS && (cfg->num_simulations < 1 || cfg->num_env_copies < 1)) return EWN_EINVAL;
    u32 W = cfg->mt_window ? cfg->mt_window : 128u;
    if (W < 16 || W > EWN_MT_WINDOW_MAX) return EWN_EINVAL;
    k.N = cfg->n_lanes; k.opp = cfg->opponent_kind; k.depth = cfg->max_depth; k.heur = cfg->heuristic;
    k.rng_kind = cfg->rng_kind; k.shaped = cfg->shaped; k.autoreset = cfg->autoreset; k.refresh = cfg->shaped_refresh_on_reset;
    k.lane_offset = cfg->lane_offset; k.nsim_total = cfg->num_simulations * cfg->num_env_copies;
    k.seed_stride = cfg->seed_stride; k.W = W;
    k.rng_words = EWN_RNG_HDR + (cfg->rng_kind == EWN_RNG_MT19937 ? 3u * W + 1u : 0u);
    k.reward = cfg->reward; k.illegal_reward = cfg->illegal_move_reward; k.key = cfg->philox_key;
    return EWN_OK;
}

static inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? EWN_OK : EWN_ELAUNCH;
}

// specialised depth-3 tables exist for cube_layer 3 and board sizes whose distinct leaf values fit 10-bit ranks
static inline int64_t fast_tables_bytes(int S, int L)
{
    if (L != 3) return 0;
    switch (S) {
    case 5: return (int64_t)FAST_TAB_BYTES(5);
    case 6: return (int64_t)FAST_TAB_BYTES(6);
    case 7: return (int64_t)FAST_TAB_BYTES(7);
    case 8: return (int64_t)FAST_TAB_BYTES(8);
    default: return 0;
    }
}

// table image of a search with the given heuristic and max_depth: max_depth 4 and 6 read the image's second variant (leaves
// averaged over six dice); NULL for a heuristic without images
static inline const void *fast_image(const void *tables, int S, int L, int max_depth, int heur = EWN_H_HYBRID)
{
    const int hi = fast_heur_image(heur);
    if (!tables || hi < 0) return nullptr;
    return (const int8_t *)tables + (size_t)(hi * 2 + ((max_depth == 4 || max_depth == 6) ? 1 : 0)) * fast_tables_bytes(S, L);
}

// Lanes of one wavefront that share a game in k_step_d3: enough to put several waves on every SIMD
// (1024 SIMDs x 64 lanes) at the given number of games.  EWN_D3_T=0 disables the kernel, 1 / 2 forces T.
static inline int d3_threads_per_game(int n_games)
{
    static const int forced = [] { const char *e = getenv("EWN_D3_T"); return e ? atoi(e) : -1; }();
    if (forced == 0 || forced == 1 || forced == 2) return forced;
    // measured on MI355X (tools/sweep_T.sh, us per step, T = 1 / 2 / 4): 16 384 games 14.9 / 10.9 / 10.0; 32 768: 15.1 / 11.3 / 12.6;
    // 65 536: 15.8 / 14.8 / 18.7; 131 072: 21.6 / 22.4 / 32.3; 262 144: 36.3 / 39.4 / 57.5; 1 048 576: 118 / 132 / 205.
    // The kernel is bound by integer VALU issue once the chip is full, so lanes added beyond what hides the LDS/global
    // latency only add redundant instructions.
    // Re-measured at the end of round 2 (table-driven setup, one-min pair selection): two lanes per game are now as fast as or faster
    // than four at every lane count below 131 072 (2 048 .. 24 576 games: one-step launch 9.9-10.4 us against 10.1-12.1), so the
    // four-lanes-per-game instances were dropped from the library (d3_search keeps its T = 4 code: `own_of` / `publish`).
    if (n_games >= 131072) return 1;
    return 2;
}

// the lean kernel's MT refill hand-off (ewn_core.hpp MtQueue): ctrl[4] | cnt[2][nb4] | list[2][nblk][2 * games per block] x 16 B.
// The layout depends on the lanes-per-game T the launch picks (games per block = 256 / T, nblk = ceil(N / games per block)), and
// the rounding-up of nblk makes T = 1 the LARGEST for a small N (64 lanes: one block of 512 slots), so take the maximum over T.
// (Sizing it for T = 4 only let a T = 1 launch on 64 lanes write 8 KB past the buffer -- into whatever tensor came next.)
static inline int64_t mtq_bytes(int64_t N)
{
    int64_t best = 0;
    for (int T = 1; T <= 4; T *= 2) {
        const int64_t gpb = D3_BS / T, nblk = (N + gpb - 1) / gpb, nb4 = (nblk + 3) / 4 * 4;
        const int64_t b = 16 + 2 * nb4 * 4 + 2 * nblk * (2 * gpb) * 16;
        if (b > best) best = b;
    }
    return best;
}


// the same choice for the K-step rollout kernel (ewn_rollout.hpp), whose per-step cost outside the search is much smaller
static inline int rollout_threads_per_game(int n_games)
{
    static const int forced = [] { const char *e = getenv("EWN_ROLLOUT_T"); return e ? atoi(e) : -1; }();
    if (forced == 1 || forced == 2) return forced;
    // measured (us per step with the trajectory, T = 2 / 4): 2 048 games 6.46 / 6.93, 8 192: 6.56 / 7.11, 24 576: 6.79 / 8.45
    if (n_games >= 131072) return 1;
    return 2;
}

// ewn_step_d3.hip: the lean table-driven step kernel (one launch = one env step)
int ewn_launch_step_d3(const ewn_config *cfg, const Geom &g, const KCfg &k, const ewn_state *st, const void *tables, const int8_t *actions,
                       const ewn_step_out *out, void *scratch, bool lean_random, bool fused_refill, hipStream_t s);
// ewn_step_d3_h2.hip: the same for the 'two_min_dist' table image
int ewn_launch_step_d3_h2(const ewn_config *cfg, const Geom &g, const KCfg &k, const ewn_state *st, const void *tables, const int8_t *actions,
                          const ewn_step_out *out, void *scratch, bool lean_random, bool fused_refill, hipStream_t s);

// ewn_rollout_s<S>.hip: K env steps per launch (ewn_rollout.hpp), one unit per board size
struct RollCfg;
struct RollBuf;
int ewn_launch_rollout_s5(const RollCfg &rc, const RollBuf &rb, int T, int opp, int rngk, int agent, bool h2, hipStream_t s);
int ewn_launch_rollout_s6(const RollCfg &rc, const RollBuf &rb, int T, int opp, int rngk, int agent, bool h2, hipStream_t s);
int ewn_launch_rollout_s7(const RollCfg &rc, const RollBuf &rb, int T, int opp, int rngk, int agent, bool h2, hipStream_t s);
int ewn_launch_rollout_s8(const RollCfg &rc, const RollBuf &rb, int T, int opp, int rngk, int agent, bool h2, hipStream_t s);
// ewn_policy.hip: the policy-driven rollout (ewn_step_k_policy); EWN_OK when it serves the configuration
int ewn_policy_supported(const ewn_config *cfg, const Geom &g);
