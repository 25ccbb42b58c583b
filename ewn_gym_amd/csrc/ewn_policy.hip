// ewn_policy.hip -- the policy-driven rollout (ewn_step_k_policy) and the fused A2C update (ewn_a2c_*): kernels in
// ewn_policy.hpp / ewn_a2c.hpp, C ABI here.
#include "ewn_host.hpp"
#include "ewn_lds.hpp"
#include "ewn_policy.hpp"
#include "ewn_a2c.hpp"
#include "ewn_a2c2.hpp"
#include "ewn_a2c3.hpp"

// which instantiation serves the configuration: opp 0 minimax (table image, max_depth 1-4), 1 RandomAgent
static int policy_plan(const ewn_config *cfg, const Geom &g, int &opp)
{
    if (fast_tables_bytes(g.S, g.L) <= 0 || (g.S != 5 && g.S != 7)) return EWN_EUNSUPPORTED;
    if (cfg->rng_kind != EWN_RNG_PHILOX) return EWN_EUNSUPPORTED;
    if (cfg->opponent_kind == EWN_OPP_RANDOM) opp = 1;
    else if (cfg->opponent_kind == EWN_OPP_MINIMAX && fast_heur_lean(cfg->heuristic) && cfg->max_depth <= 4) opp = 0;
    else return EWN_EUNSUPPORTED;
    return EWN_OK;
}

int ewn_policy_supported(const ewn_config *cfg, const Geom &g)
{
    int opp;
    return policy_plan(cfg, g, opp);
}

int64_t ewn_policy_param_count(int board_size, int cube_layer)
{
    if (cube_layer != 3) return EWN_EUNSUPPORTED;
    switch (board_size) {
    case 5: return MlpGeo<5>::P;
    case 7: return MlpGeo<7>::P;
    default: return EWN_EUNSUPPORTED;
    }
}

template <int S, int OPP, int NT, int TRJ = 0>
static int pol_launch(const PolCfg &pc, const PolBuf &pb, hipStream_t s)
{
    if constexpr (TRJ == 0 && OPP == 0) {
        // the trainer's call (FusedA2CTrainer): records from the initial observation on + the reward column, nothing else per step,
        // sampled actions, no value output -- the instance that knows it at compile time
        const bool trainer = pb.t_rec && pb.t_reward && pc.rec0 && !pc.want_value && !pc.deterministic && !pb.t_board && !pb.t_dice && !pb.t_action
                             && !pb.t_term && !pb.t_trunc && !pb.t_info && !pb.t_logits && !pb.t_value && !pb.t_noise;
        if (trainer) return pol_launch<S, OPP, NT, 1>(pc, pb, s);
    }
    auto kern = k_rollout_mlp<S, OPP, NT, TRJ>;
    const size_t lds = pol_lds_bytes<S, NT>(pc.want_value != 0);
    if (lds > 160 * 1024) return EWN_EUNSUPPORTED;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    const int gpb = NT / 2;
    kern<<<dim3((unsigned)((pc.N + gpb - 1) / gpb)), NT, lds, s>>>(pc, pb);
    return launch_status();
}

template <int S>
static int pol_dispatch(const PolCfg &pc, const PolBuf &pb, int opp, hipStream_t s)
{
    // 512 threads (256 games) per block where the block's LDS fits the CU, else 256
    const bool big = pol_lds_bytes<S, 512>(pc.want_value != 0) <= 160 * 1024;
    if (opp == 0) return big ? pol_launch<S, 0, 512>(pc, pb, s) : pol_launch<S, 0, 256>(pc, pb, s);
    return big ? pol_launch<S, 1, 512>(pc, pb, s) : pol_launch<S, 1, 256>(pc, pb, s);
}

int ewn_step_k_policy(const ewn_config *cfg, const ewn_state *st, int K, const ewn_policy *pol, const ewn_rollout_out *out, void *stream)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (K < 1) return EWN_EINVAL;
    if (!st || !st->board || !st->dice || !st->done || !st->rng || !st->tables || !pol || !pol->params) return EWN_ENULL;
    if (cfg->shaped && (!st->prev_score || !st->tolerance)) return EWN_ENULL;
    int opp;
    rc = policy_plan(cfg, g, opp);
    if (rc) return rc;
    PolCfg pc;
    pc.N = k.N; pc.autoreset = k.autoreset; pc.lane_offset = k.lane_offset; pc.depth = k.depth; pc.K = K;
    pc.shaped = k.shaped; pc.refresh = k.refresh; pc.deterministic = pol->deterministic ? 1 : 0; pc.want_value = pol->value ? 1 : 0;
    pc.rec0 = pol->record_initial_obs ? 1 : 0;
    static const int stagger = [] { const char *e = getenv("EWN_POLICY_STAGGER"); return e ? atoi(e) : 0; }();   // tuning knob; measured: no effect (the f32 MFMA does not overlap with the other wave's VALU work)
    pc.stagger = stagger;
    pc.seed_stride = k.seed_stride; pc.W = k.W; pc.reward = k.reward; pc.illegal_reward = k.illegal_reward;
    pc.key = k.key; pc.noise_key = pol->noise_key;
    PolBuf pb;
    memset(&pb, 0, sizeof(pb));
    pb.board = st->board; pb.dice = st->dice; pb.done = st->done; pb.rng = st->rng; pb.prev_score = st->prev_score; pb.tolerance = st->tolerance;
    pb.tables = opp == 1 ? st->tables : fast_image(st->tables, g.S, g.L, cfg->max_depth, cfg->heuristic);
    pb.params = pol->params;
    pb.t_logits = pol->logits; pb.t_value = pol->value; pb.t_noise = pol->noise;
    if (out) {
        pb.t_board = out->board; pb.t_dice = out->dice; pb.t_action = out->action; pb.t_reward = out->reward;
        pb.t_term = out->terminated; pb.t_trunc = out->truncated; pb.t_info = out->info; pb.t_rec = out->record;
        pb.ret_sum = out->return_sum; pb.n_steps = out->n_steps; pb.n_episodes = out->n_episodes; pb.n_wins = out->n_wins;
    }
    hipStream_t s = (hipStream_t)stream;
    return g.S == 5 ? pol_dispatch<5>(pc, pb, opp, s) : pol_dispatch<7>(pc, pb, opp, s);
}

// ---------------------------------------------------------------- the A2C update

#define A2C_MAX_BLOCKS 256   // one block per CU

template <int S> struct A2cWaves { static constexpr int N = 4; };
template <> struct A2cWaves<7> { static constexpr int N = 3; };   // the 7x7 images leave room for three waves' transpose tiles

static int a2c_blocks(int N, int nwv)
{
    const int tiles = (N + 31) / 32, need = (tiles + nwv - 1) / nwv;
    return need < A2C_MAX_BLOCKS ? need : A2C_MAX_BLOCKS;
}

static int a2c_geom(const ewn_config *cfg, Geom &g, KCfg &k)
{
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (g.L != 3 || (g.S != 5 && g.S != 7)) return EWN_EUNSUPPORTED;
    return EWN_OK;
}

int64_t ewn_a2c_scratch_bytes(const ewn_config *cfg, int K)
{
    Geom g; KCfg k;
    int rc = a2c_geom(cfg, g, k);
    if (rc) return rc;
    if (K < 1) return EWN_EINVAL;
    const int64_t P = ewn_policy_param_count(g.S, g.L);
    return ((int64_t)K * k.N + (int64_t)A2C_MAX_BLOCKS * (P + 8) + 64) * 4;
}

// partials -> flat gradient: the 8-byte-load kernel where the layout allows it
static void a2c_reduce_launch(const A2cRedBuf &rb, hipStream_t s)
{
    if ((rb.P & 1) == 0 && ((uintptr_t)rb.partial & 7) == 0) k_a2c_reduce2<<<(rb.P + 8 + 2 * A2C_RED2_E - 1) / (2 * A2C_RED2_E), 256, 0, s>>>(rb);
    else k_a2c_reduce<<<(rb.P + 8 + A2C_RED_E - 1) / A2C_RED_E, 256, 0, s>>>(rb);
}

// 5x5, f32 MFMA: two waves per tile (k_a2c_grad2)
static int a2c_grad_launch_team(const A2cCfg &ac, A2cBuf ab, float *grad, hipStream_t s)
{
    constexpr size_t lds = A2c2Geo<5>::lds_bytes();
    static_assert(lds <= 160 * 1024, "weight images + four teams' tiles must fit the CU's LDS");
    auto kv = k_a2c_grad2<5, 1>;
    auto kp = k_a2c_grad2<5, 0>;
    if (hipFuncSetAttribute((const void *)kv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    if (hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    const int blocks = a2c_blocks(ac.N, A2c2Geo<5>::TEAMS);
    kv<<<blocks, 512, lds, s>>>(ac, ab);
    kp<<<blocks, 512, lds, s>>>(ac, ab);
    A2cRedBuf rb = { ab.partial, ab.stats, grad, blocks, MlpGeo<5>::P };
    a2c_reduce_launch(rb, s);
    return launch_status();
}

// the bf16 x 3 kernel (ewn_a2c3.hpp): one wave per tile and SIMD, everything in registers
template <int S>
static int a2c_grad_launch_b3(const A2cCfg &ac, A2cBuf ab, float *grad, hipStream_t s)
{
    constexpr size_t lds = A2c3Geo<S>::lds_bytes();
    static_assert(lds <= 160 * 1024, "weight images + the gradient image must fit the CU's LDS");
    auto kv = k_a2c_grad3<S, 1>;
    auto kp = k_a2c_grad3<S, 0>;
    if (hipFuncSetAttribute((const void *)kv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    if (hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    const int blocks = a2c_blocks(ac.N, 4);
    kv<<<blocks, 256, lds, s>>>(ac, ab);         // value pass first: it leaves the advantages for the policy pass
    kp<<<blocks, 256, lds, s>>>(ac, ab);
    A2cRedBuf rb = { ab.partial, ab.stats, grad, blocks, MlpGeo<S>::P };
    a2c_reduce_launch(rb, s);
    return launch_status();
}

// EWN_A2C_KERNEL: 3 (default) the bf16 x 3 register kernel; 2 the f32-MFMA team kernel (5x5); 1 the f32-MFMA one-wave kernel -- the
// older two stay for A/B measurements and as independent implementations the tests compare
template <int S>
static int a2c_grad_launch(const A2cCfg &ac, A2cBuf ab, float *grad, hipStream_t s)
{
    static const int kind = [] { const char *e = getenv("EWN_A2C_KERNEL"); return e ? atoi(e) : 3; }();
    if (kind >= 3 || kind <= 0) return a2c_grad_launch_b3<S>(ac, ab, grad, s);
    if (S == 5 && kind == 2) return a2c_grad_launch_team(ac, ab, grad, s);
    constexpr int NWV = A2cWaves<S>::N;
    constexpr size_t lds = a2c_lds_bytes<S, NWV>();
    static_assert(lds <= 160 * 1024, "the gradient kernel's images and transpose tiles must fit the CU's LDS");
    auto kv = k_a2c_grad<S, 1, NWV>;
    auto kp = k_a2c_grad<S, 0, NWV>;
    if (hipFuncSetAttribute((const void *)kv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    if (hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EWN_ELAUNCH;
    const int blocks = a2c_blocks(ac.N, NWV);
    kv<<<blocks, NWV * 64, lds, s>>>(ac, ab);    // value pass first: it leaves the advantages for the policy pass
    kp<<<blocks, NWV * 64, lds, s>>>(ac, ab);
    A2cRedBuf rb = { ab.partial, ab.stats, grad, blocks, MlpGeo<S>::P };
    a2c_reduce_launch(rb, s);
    return launch_status();
}

int ewn_a2c_grad(const ewn_config *cfg, int K, const uint8_t *record, const double *reward, const float *params, const ewn_a2c_hyper *hp,
                 float *grad, void *scratch, void *stream)
{
    Geom g; KCfg k;
    int rc = a2c_geom(cfg, g, k);
    if (rc) return rc;
    if (K < 1) return EWN_EINVAL;
    if (!record || !reward || !params || !hp || !grad || !scratch) return EWN_ENULL;
    const int64_t P = ewn_policy_param_count(g.S, g.L);
    A2cCfg ac = { k.N, K, hp->gamma, hp->vf_coef, hp->ent_coef, 1.0f / ((float)K * (float)k.N) };
    A2cBuf ab;
    ab.rec = record; ab.reward = reward; ab.params = params;
    ab.adv = (float *)scratch;
    ab.partial = ab.adv + (size_t)K * k.N;
    ab.stats = ab.partial + (size_t)A2C_MAX_BLOCKS * P;
    hipStream_t s = (hipStream_t)stream;
    return g.S == 5 ? a2c_grad_launch<5>(ac, ab, grad, s) : a2c_grad_launch<7>(ac, ab, grad, s);
}

int ewn_a2c_apply(const ewn_config *cfg, float *params, float *sq_avg, const float *grad, const ewn_a2c_hyper *hp, float *grad_norm_out,
                  void *stream)
{
    Geom g; KCfg k;
    int rc = a2c_geom(cfg, g, k);
    if (rc) return rc;
    if (!params || !sq_avg || !grad || !hp) return EWN_ENULL;
    if (hp->world_size < 1) return EWN_EINVAL;
    A2cApplyCfg ac = { (int)ewn_policy_param_count(g.S, g.L), hp->learning_rate, hp->rms_alpha, hp->rms_eps, hp->max_grad_norm, 1.0f / (float)hp->world_size };
    const bool vec = ac.P <= 1024 * 4 * A2C_APPLY_V && ((uintptr_t)params | (uintptr_t)sq_avg | (uintptr_t)grad) % 16 == 0;
    if (vec) k_a2c_apply_v4<<<1, 1024, 0, (hipStream_t)stream>>>(ac, params, sq_avg, grad, grad_norm_out);
    else k_a2c_apply<<<1, 1024, 0, (hipStream_t)stream>>>(ac, params, sq_avg, grad, grad_norm_out);
    return launch_status();
}
