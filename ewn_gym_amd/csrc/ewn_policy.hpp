// ewn_policy.hpp -- K env steps per launch with the TRAINED policy as the agent (C ABI: ewn_step_k_policy, EWN_AGENT_MLP):
// the rollout collector of the reference's trainer (train.py:35-63, 134, 148: SB3 A2C.learn -> collect_rollouts over
// SubprocVecEnv workers of `MiniMaxHeuristicEnv`, envs/training_ewn.py:40-99) as ONE kernel.  Per env step and game:
// observation -> features -> the policy network on the matrix cores (ewn_mlp.hpp) -> Gumbel-max sample of MultiDiscrete([2, 3])
// -> the env step (plain or reward-shaped, with its tolerance counter) -> the opponent's search -> reply -> auto-reset ->
// one trajectory record.  The game never leaves its registers, the network's weights never leave LDS.
//
// Lock-step over env steps (not the slot-task loop of k_rollout_slots): the network is evaluated for a whole wave of games at
// once, so the games of a wave have to be at the same step.  Two lanes per game (as k_rollout_d3), 512 threads = 256 games per
// block so that ONE table image and ONE weight image serve eight waves: 65 536 games are 256 blocks, one per CU, two waves per
// SIMD -- while one wave's MFMAs run, the other's search has the VALU.
#pragma once
#include "ewn_rollout.hpp"
#include "ewn_mlp3.hpp"

struct PolCfg {
    int N, autoreset, lane_offset, depth, K;
    int stagger;                                            // start delay of every other wave, in units of 64 cycles (0: none)
    int shaped, refresh, deterministic, want_value, rec0;   // rec0: record row 0 = the observation before step 0 (then K + 1 rows)
    u32 seed_stride, W;
    double reward, illegal_reward;
    u64 key, noise_key;
};

struct PolBuf {
    int8_t *board; int8_t *dice; uint8_t *done; u32 *rng; double *prev_score; int32_t *tolerance;
    const void *tables;
    const float *params;
    // trajectory (all optional): the columns of ewn_rollout_out, the record, and the policy's own outputs
    int8_t *t_board; int8_t *t_dice; int8_t *t_action; double *t_reward; uint8_t *t_term; uint8_t *t_trunc; uint8_t *t_info; uint8_t *t_rec;
    float *t_logits; float *t_value; float *t_noise;
    double *ret_sum; int32_t *n_steps; int32_t *n_episodes; int32_t *n_wins;
};

// LDS-DMA copy of the table image for a block of NT threads (tables_to_lds assumes 256)
template <int BYTES, int NT>
EWN_DEV void tables_to_lds_nt(int8_t *lds, const int8_t *g)
{
    static_assert(BYTES % 4096 == 0, "table size must be padded to 4 KiB");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int off = wave * 1024; off < BYTES; off += (NT / 64) * 1024)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
}

// LDS bytes of k_rollout_mlp<S, ., NT>: table image | weight image(s) | 8 floats of head outputs per game | game slots | decode scratch
template <int S, int NT>
constexpr size_t pol_lds_bytes(bool want_value)
{
    constexpr int GPB = NT / 2;
    return (size_t)FAST_TAB_BYTES(S) + (size_t)(want_value ? 2 : 1) * Mlp3Geo<S>::FWD_BYTES + (size_t)GPB * 8 * 4
           + (size_t)GPB * RecGeo<S>::STR + (size_t)GPB * 16;
}

// the uniforms behind a step's Gumbel noise: five words hashed out of the engine's per-step agent hash (the stream
// ewn_step_out.random_action draws from), as floats in (0, 1)
EWN_DEV float pol_uniform(u32 w0, int i)
{
    const u32 w = fmix32(w0 + (u32)(i + 1) * 0x9E3779B9u);
    return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// ln x on v_log_f32 (log2, ~1 ulp) -- the Gumbel noise -ln(-ln u) ten times per game and step; the library logf is ~20 instructions each
EWN_DEV float pol_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// OPP 0: minimax max_depth 1-4 on a (level, count) table image; 1: RandomAgent.  Philox dice.
// TRJ 1: the trainer's call, known at compile time -- records (row 0 = the initial observation) and the reward column, nothing else
// written per step, actions sampled, no value output (FusedA2CTrainer; the gradient kernels recompute the forward pass).  0: everything
// by PolCfg / PolBuf at run time (a dozen loop-invariant tests whose masks the compiler keeps in spilled SGPRs: 126 of them).
template <int S, int OPP, int NT, int TRJ = 0>
__global__ __launch_bounds__(NT, NT / 256) void k_rollout_mlp(PolCfg c, PolBuf B)
{
    constexpr bool FIX = TRJ == 1;
    const bool want_value = !FIX && c.want_value, deterministic = !FIX && c.deterministic, rec0 = FIX || c.rec0;
    constexpr int T = 2, GPB = NT / T, NW = NT / 64, CELLS = S * S, STR = RecGeo<S>::STR, NCH = RecGeo<S>::NCH;
    using G = MlpGeo<S>;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    using Q3 = Mlp3Geo<S>;
    static_assert(Q3::FWD_BYTES % 16 == 0, "image alignment");
    int8_t *tb = lds;
    int8_t *Wpi = lds + FAST_TAB_BYTES(S);
    int8_t *Wvf = Wpi + Q3::FWD_BYTES;
    float *lx_all = (float *)(Wpi + (want_value ? 2 : 1) * Q3::FWD_BYTES);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *LX = lx_all + wave * 32 * 8;                    // this wave's 8 floats per game of head outputs
    int8_t *slots = (int8_t *)(lx_all + GPB * 8);
    uint8_t *garr = (uint8_t *)(slots + GPB * STR);
    tables_to_lds_nt<FAST_TAB_BYTES(S), NT>(tb, (const int8_t *)B.tables);
    const FastTab<S> *Tb = (const FastTab<S> *)tb;
    mlp3_pack_fwd<S>(Wpi, B.params, 0, threadIdx.x, NT);
    if (want_value) mlp3_pack_fwd<S>(Wvf, B.params, 1, threadIdx.x, NT);

    const int g0 = (int)blockIdx.x * GPB, ng = min(GPB, c.N - g0);
    const int gl = threadIdx.x / T, sub = threadIdx.x % T, game = g0 + gl;
    const int jw = lane >> 1;                              // my game's sample column inside the wave's tile
    const bool live = game < c.N, writer = live && sub == 0;

    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1, tol = 0;
    double prev = 0.0;
    bool frozen = true;
    if (live) {
        hdr = *rng_hdr_ptr(B.rng, game);
        dice = B.dice[game];
        frozen = B.done[game] != 0;
        if (c.shaped) { tol = B.tolerance[game]; prev = B.prev_score[game]; }
    }
    const bool frozen0 = frozen;
    block_copy_in(slots, B.board + (size_t)g0 * CELLS, ng * CELLS);      // packed boards, decoded from there
    LaneRng r; r.load(1, hdr, nullptr, c.W, c.key);
    r.begin_kernel();
    lds_dma_wait();
    __syncthreads();
    RState<S> s;
    d3_decode<S, T>(live ? slots + gl * CELLS : slots, sub, garr + gl * 16, s);
    __syncthreads();                                       // every game is in registers: the board area becomes the per-game slots
    int8_t *slot = slots + gl * STR;
    rec_slot_build<S, T>(Tb, s, sub, slot);
    if ((FIX || B.t_rec) && rec0 && live) rec_store<S, T>(slot, sub, dice, 0, 0, 0, 0, 0, B.t_rec + (size_t)game * STR);
    double ret_acc = 0.0;
    int n_steps = 0, n_eps = 0, n_wins = 0;
    // The two waves that share a SIMD run the same loop; started together they stay in phase -- both in the network (the matrix
    // pipe contended, the VALU idle), then both in the search (the reverse).  The block's second half of waves starts c.stagger x 64 cycles late, about
    // one network evaluation, so that one wave's MFMAs run under the other's search from then on (nothing in the loop re-aligns them).
    if (wave >= NW / 2 && c.stagger > 0) { for (int i = 0; i < c.stagger; i += 127) __builtin_amdgcn_s_sleep(127); }   // waves w and w + NW / 2 share a SIMD

    #pragma unroll 1
    for (int kstep = 0; kstep < c.K; kstep++) {
        const bool active = live && !frozen;
        double reward = 0.0;
        int term = 0, trunc = 0, info = EWN_INFO_NONE;
        if (live && frozen) term = 1;
        // ---- the network(s): this wave's 32 games are the 32 columns of the MFMA tiles (game j of the wave = column j, both lane halves);
        // the features come straight out of the games' slots: lane (j, h) turns bytes 16 kb + 8 h .. + 7 of game j's board into the eight
        // bf16 of its k-block operand (the slot's bytes past the board are zero) and sets the dice one-hot (features CELLS .. CELLS + 6)
        __builtin_amdgcn_wave_barrier();
        {
            const int j = lane & 31, h = lane >> 5;
            const int dj = __builtin_amdgcn_ds_bpermute((2 * j) << 2, dice);          // game j's dice (its lanes are 2 j, 2 j + 1)
            const int8_t *sj = slots + (wave * 32 + j) * STR + 8 * h;
            auto xb = [&](int kb) {
                const uint2 v = *(const uint2 *)(sj + 16 * kb);
                u32x4 o = mlp3_bytes_operand(v.x, v.y);
                if (16 * kb + 15 >= CELLS && 16 * kb < CELLS + 7) o = mlp3_onehot(o, CELLS + dj - 1 - (16 * kb + 8 * h));
                return o;
            };
            f32x16 h1[2], h2[2];
            float lo[MLP_NA];
            mlp3_forward<S, MLP_NA>(Wpi, lane, xb, h1, h2, lo);
            if (lane < 32) { *(float4 *)(LX + lane * 8) = make_float4(lo[0], lo[1], lo[2], lo[3]); LX[lane * 8 + 4] = lo[4]; }
            if (want_value) {
                float vo[1];
                mlp3_forward<S, 1>(Wvf, lane, xb, h1, h2, vo);
                if (lane < 32) LX[lane * 8 + 5] = vo[0];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float4 lg = *(const float4 *)(LX + jw * 8);
        const float lg4 = LX[jw * 8 + 4], val = LX[jw * 8 + 5];
        __builtin_amdgcn_wave_barrier();
        // ---- Gumbel-max sample (argmax of logits when deterministic): a[0] ~ softmax(l0, l1), a[1] ~ softmax(l2, l3, l4)
        // keyed by (episode, draws so far, lane) like the stand-in agents' hash, and by the tolerance left: an illegal move of the shaped
        // env (training_ewn.py:48-56) changes neither the observation nor the dice stream, and must not replay the same noise
        const u32 w0 = agent_hash(r.seed_mix() ^ ((u32)tol * 0x632BE5ABu), r.draws(), (u32)(c.lane_offset + game), c.key ^ c.noise_key);
        float u[5], gn[5];
        #pragma unroll
        for (int i = 0; i < 5; i++) { u[i] = pol_uniform(w0, i); gn[i] = deterministic ? 0.0f : -pol_log(-pol_log(u[i])); }
        const float z0 = lg.x + gn[0], z1 = lg.y + gn[1], z2 = lg.z + gn[2], z3 = lg.w + gn[3], z4 = lg4 + gn[4];
        const int aflag = z1 > z0 ? 1 : 0;
        const int adir = z3 > z2 ? (z4 > z3 ? 2 : 1) : (z4 > z2 ? 2 : 0);
        if (!FIX && writer) {
            const size_t o = (size_t)kstep * c.N + game;
            if (B.t_logits) { float *p = B.t_logits + o * 5; p[0] = lg.x; p[1] = lg.y; p[2] = lg.z; p[3] = lg.w; p[4] = lg4; }
            if (B.t_value) B.t_value[o] = val;
            if (B.t_noise) { float *p = B.t_noise + o * 5; for (int i = 0; i < 5; i++) p[i] = u[i]; }
        }
        // ---- agent half, envs/ewn.py:438-458 / envs/training_ewn.py:44-66 (the agent is the canonical BOTTOM_RIGHT side)
        bool reply = false;
        if (active) {
            r.begin_step();
            r.ps.prime();
            const int k = pk_cube(pk_sel<S>(Tb, s.posN, dice), aflag == 1);
            const int pb = pk_get(s.posN, k);
            const int q = Tb->nbn[adir][pb];
            if (q == 255) {
                if (c.shaped) { // an illegal move costs tolerance; the game goes on until it is used up (training_ewn.py:48-56)
                    tol -= 1;
                    if (tol <= 0) { reward = -c.reward; term = 1; trunc = 1; info = EWN_INFO_INVALID_PLAYER; }
                    else { reward = c.illegal_reward; info = EWN_INFO_TOLERANCE; }
                } else { reward = -c.reward; term = 1; trunc = 1; info = EWN_INFO_INVALID_PLAYER; }
            } else {
                const int cp = Tb->real_of_ring[pb & 63], cq = Tb->real_of_ring[q];
                slot[cp] = 0; slot[cq] = (int8_t)(k + 1);
                rs_move<S, false>(s, k, q);
                if (q == Tb->ri_origin || s.P == 0) { reward = c.reward; term = 1; info = EWN_INFO_WON; }
                else { dice = r.randint(1, 7); reply = true; }
            }
        }
        // ---- the opponent's search: run by every lane (lanes without a pending reply compute on a harmless state)
        int oflag = 0, odir = 0;
        if constexpr (OPP == 0) d3_search<S, T>(Tb, s, dice, sub, c.depth, oflag, odir);
        if (reply) {
            const u32 e = pk_sel<S>(Tb, s.posP, dice);
            if constexpr (OPP == 1) {
                const u32 pp = pk_pair(s.posP, e);
                const u32 okm = (u32)Tb->lgp[pp & 0xFFu] | ((u32)Tb->lgp[pp >> 8] << 3);
                const int sl = Tb->nth[okm * 8u + (u32)r.randint(0, __popc(okm))];
                oflag = sl < 3 ? (int)(e >> 15) : 0;
                odir = sl < 3 ? sl : sl - 3;
            }
            roll_opponent_half<S>(Tb, s, e, oflag, odir, dice, r, c.reward, reward, term, info, slot);
            if (c.shaped && !term) { // reward = evaluate() - prev_score (training_ewn.py:94-96)
                const double cur = d3_shaped_score<S>(Tb, s);
                reward = cur - prev;
                prev = cur;
            }
        }
        if (active) {
            ret_acc += reward; n_steps++; n_eps += term; n_wins += info == EWN_INFO_WON ? 1 : 0;
            if (term) {
                if (c.autoreset) {
                    r.next_episode(B.rng, c.N, game, c.seed_stride, c.key, nullptr);
                    d3_init_state<S>(Tb, s);
                    rec_slot_init<S>(slot);
                    dice = r.first_dice(6);
                    if (c.shaped && c.refresh) prev = d3_shaped_score<S>(Tb, s);
                } else frozen = true;
            }
        }
        // ---- this step's trajectory row
        __builtin_amdgcn_wave_barrier();
        if (live) {
            const size_t o = (size_t)kstep * c.N + game;
            if constexpr (FIX) {
                if (sub == 0) B.t_reward[o] = reward;
            } else if (sub == 0) {
                if (B.t_action) ((uint16_t *)B.t_action)[o] = (uint16_t)((uint8_t)aflag | ((uint16_t)(uint8_t)adir << 8));
                if (B.t_dice) B.t_dice[o] = (int8_t)dice;
                if (B.t_reward) B.t_reward[o] = reward;
                if (B.t_term) B.t_term[o] = (uint8_t)term;
                if (B.t_trunc) B.t_trunc[o] = (uint8_t)trunc;
                if (B.t_info) B.t_info[o] = (uint8_t)info;
            }
            if (FIX || B.t_rec) rec_store<S, T>(slot, sub, dice, aflag, adir, term, trunc, info, B.t_rec + (o + (rec0 ? (size_t)c.N : 0)) * STR);
            if (!FIX && B.t_board && sub == 0) { int8_t *dst = B.t_board + o * CELLS; for (int i = 0; i < CELLS; i++) dst[i] = slot[i]; }
        }
    }
    // ---- the state goes back to HBM once, through the packed board area
    __syncthreads();
    if (live) d3_encode<S, T>(Tb, s, sub, slots + gl * CELLS);
    if (writer) {
        if (!frozen0) { *rng_hdr_ptr(B.rng, game) = r.header(); B.dice[game] = (int8_t)dice; }
        B.done[game] = frozen ? 1 : 0;
        if (c.shaped) { B.tolerance[game] = tol; B.prev_score[game] = prev; }
        if (B.ret_sum) B.ret_sum[game] += ret_acc;
        if (B.n_steps) B.n_steps[game] += n_steps;
        if (B.n_episodes) B.n_episodes[game] += n_eps;
        if (B.n_wins) B.n_wins[game] += n_wins;
    }
    __syncthreads();
    block_copy_out(B.board + (size_t)g0 * CELLS, slots, ng * CELLS);
}
