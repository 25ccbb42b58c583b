// ewn_rollout.hpp -- K env steps in ONE launch (C ABI: ewn_step_k), for the case where the agent is an engine policy too:
// the loop  `while not done: action, _ = agent.predict(obs); obs, reward, done, trunc, info = env.step(action)`  of the
// reference's evaluation scripts (eval_minimax.py:16-50, eval_pairs.py:10-35) and of a rollout collector, with
// agent in {RandomAgent, ExpectiMinimaxAgent('hybrid')}.
//
// Why a second kernel next to k_step_d3 (ewn_step_d3.hpp), measured there with in-kernel stamps at 65 536 lanes: of the
// ~26 k cycles a wave spends per step, ~4.7 k are loading the state, staging the boards through LDS and decoding them and
// ~3.6 k encoding and storing them again, and every launch ends with a device-wide barrier (the next step cannot start before
// the slowest block of this one has finished: ~8 k cycles of ramp and tail per launch).  Here a game is loaded ONCE, stays
// in registers (two occupancy masks + twelve position bytes + the RNG header) for K steps, and is stored once; blocks run
// their K steps without ever waiting for each other.  Every step's observation / action / reward / flags can still be
// written (ewn_rollout_out, a [K][N] trajectory: what a trainer's rollout buffer holds), through the same LDS-staged
// coalesced copies; they are write-only traffic, nothing is read back.
//
// Results are identical, step for step, to calling ewn_step K times with the agent's action fed back (tests:
// tests/test_gpu_rollout.py, against ewn_step and against the CPU oracle).
#pragma once
#include "ewn_step_d3.hpp"

// Diagnostic build only (tools/rollout_stamps.py, -DEWN_ROLLOUT_STAMPS): s_memtime at the phase boundaries of a step, summed over the
// K steps of a launch per wave and written over the (then meaningless) return_sum buffer.  Never defined in the shipped library.
#ifdef EWN_ROLLOUT_STAMPS
#define RSTAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif

struct RollCfg {
    int N, autoreset, lane_offset, depth, agent_depth, K;  // depth / agent_depth: max_depth of the opponent's / the agent's search
    int agent_sample;                                      // AGENT 0 only: 1 = action_space.sample() (uniform over all six actions) instead of RandomAgent
    u32 seed_stride, W;
    double reward;
    u64 key;
};

struct RollBuf {
    int8_t *board; int8_t *dice; uint8_t *done; u32 *rng;
    const void *tables;        // table image of the opponent's search (ewn_fast.hpp)
    const void *agent_tables;  // table image of the agent's search; == tables when both use the same image (or the agent is random)
    // trajectory [K][N]..., each may be NULL
    int8_t *t_board; int8_t *t_dice; int8_t *t_action; double *t_reward; uint8_t *t_term; uint8_t *t_trunc; uint8_t *t_info;
    uint8_t *t_rec;            // [K][N][EWN_TRAJ_RECORD_STRIDE(S)] one aligned record per lane-step (ewn_rollout_out.record), or NULL
    // per-lane accumulators, each may be NULL
    double *ret_sum; int32_t *n_steps; int32_t *n_episodes; int32_t *n_wins;
};

// The position seen from the other side: sides swapped, every square rotated by 180 degrees (opponent_action's
// np.rot90(-board, 2), envs/ewn.py:289-296), so that the side to move is the table's TOP_LEFT mover again.
template <int S>
EWN_DEV RState<S> rs_flip(const FastTab<S> *Tb, const RState<S> &s)
{
    typedef typename MaskOf<S>::type M;
    RState<S> f;
    u32 rp[6], rn[6];
    #pragma unroll
    for (int k = 0; k < 6; k++) { rp[k] = Tb->rot[pk_get(s.posN, k) & 63]; rn[k] = Tb->rot[pk_get(s.posP, k) & 63]; }
    f.posP = PK_PADS; f.posN = PK_PADS; f.P = 0; f.N = 0;
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        const bool ap = !(pk_get(s.posN, k) & PK_OFF), an = !(pk_get(s.posP, k) & PK_OFF);
        f.posP |= (u64)(ap ? rp[k] : (u32)PK_OFF) << (8 * k);
        f.posN |= (u64)(an ? rn[k] : (u32)PK_OFF) << (8 * k);
        f.P |= ap ? ((M)1 << rp[k]) : (M)0;
        f.N |= an ? ((M)1 << rn[k]) : (M)0;
    }
    return f;
}

// ---- the trajectory as ONE aligned record per lane-step (ewn_rollout_out.record): the S*S board bytes, then dice, action[2],
// terminated, truncated, info, zero padding up to a multiple of 16 bytes (32 for 5x5, 64 for 7x7).  A game's row is then one or
// two whole 32-byte sectors written with 16-byte stores, wherever its neighbours are in their own step count (the slot-task
// kernel's games drift apart): measured round 2, the packed [K][N][S*S] board column of drifted games cost 47-69 written bytes
// per lane-step for 39 algorithmic.  The board bytes come from a per-game LDS slot that is kept CURRENT move by move (two byte
// stores per move) instead of being re-encoded from the position registers every step.
template <int S> struct RecGeo {
    static constexpr int CELLS = S * S, STR = (CELLS + 6 + 15) & ~15, NCH = STR / 16;
    static_assert(STR == EWN_TRAJ_RECORD_STRIDE(S), "include/ewn_hip.h states the stride");
};

// the start position (envs/ewn.py:94-107) as record bytes (meta and padding zero), little-endian words
template <int S> struct InitRec {
    u32 w[RecGeo<S>::STR / 4];
    constexpr InitRec() : w()
    {
        int b[RecGeo<S>::STR] = {};
        int cnt = 1;
        for (int i = 1; i <= 3; i++)
            for (int j = 0; j < i; j++) { b[j * S + (i - j - 1)] = cnt; b[(S - 1 - j) * S + (S - i + j)] = -cnt; cnt++; }
        for (int i = 0; i < RecGeo<S>::STR; i++) w[i / 4] |= (u32)(b[i] & 0xFF) << (8 * (i % 4));
    }
};

// (re)build a game's slot from the position registers: launch start only
template <int S, int T>
EWN_DEV void rec_slot_build(const FastTab<S> *Tb, const RState<S> &s, int sub, int8_t *slot)
{
    #pragma unroll
    for (int c0 = 0; c0 < RecGeo<S>::NCH; c0 += T) { const int c = c0 + sub; if (c < RecGeo<S>::NCH) ((uint4 *)slot)[c] = make_uint4(0u, 0u, 0u, 0u); }
    __builtin_amdgcn_wave_barrier();   // the zeroing of every lane of the game is issued before any cube byte
    d3_encode_cubes<S, T>(Tb, s, sub, slot);
    __builtin_amdgcn_wave_barrier();
}

// setup_game into the slot (auto-reset): every lane of the game stores the same constants
template <int S>
EWN_DEV void rec_slot_init(int8_t *slot)
{
    constexpr InitRec<S> I{};
    #pragma unroll
    for (int c = 0; c < RecGeo<S>::NCH; c++) ((uint4 *)slot)[c] = make_uint4(I.w[4 * c], I.w[4 * c + 1], I.w[4 * c + 2], I.w[4 * c + 3]);
}

// one record out: lane `sub` of the game's T lanes takes the 16-byte pieces sub, sub + T, ...; the six meta bytes are OR-ed into
// the piece(s) they fall in on the way (their slot bytes stay zero)
template <int S, int T>
EWN_DEV void rec_store(const int8_t *slot, int sub, int dice, int aflag, int adir, int term, int trunc, int info, uint8_t *dst)
{
    constexpr int CELLS = RecGeo<S>::CELLS, NCH = RecGeo<S>::NCH;
    const u32 meta[6] = { (u32)dice & 0xFFu, (u32)aflag & 0xFFu, (u32)adir & 0xFFu, (u32)term, (u32)trunc, (u32)info };
    #pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += T) {
        const int c = c0 + sub;
        if (c < NCH) {
            const uint4 v = ((const uint4 *)slot)[c];
            u32 w[4] = { v.x, v.y, v.z, v.w };
            #pragma unroll
            for (int cc = c0; cc < c0 + T && cc < NCH; cc++) { // the compile-time candidates for c
                u32 add[4] = { 0u, 0u, 0u, 0u };
                bool any = false;
                #pragma unroll
                for (int m = 0; m < 6; m++) {
                    const int off = CELLS + m;
                    if (off / 16 == cc) { add[(off % 16) / 4] |= meta[m] << (8 * (off % 4)); any = true; }
                }
                if (any) {
                    const bool me = T == 1 || c == cc;
                    #pragma unroll
                    for (int i = 0; i < 4; i++) w[i] |= me ? add[i] : 0u;
                }
            }
            *(uint4 *)(dst + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// ---- pieces of one env step shared by the two rollout kernels below (the game is in registers, canonical ring space)

// The stand-in agent's action for the current observation (the agent is the canonical BOTTOM_RIGHT side): RandomAgent.predict -- the
// same draw ewn_step_out.random_action makes at the end of the previous step -- or, with c.agent_sample, env.action_space.sample():
// one of the six (flag, direction) pairs, legal or not.  Hash-driven: the dice stream is not touched.
template <int S>
EWN_DEV void roll_stand_in_action(const FastTab<S> *Tb, const RState<S> &s, int dice, LaneRng &r, const RollCfg &c, int game, int &aflag, int &adir)
{
    const u32 e = pk_sel<S>(Tb, s.posN, dice), pp = pk_pair(s.posN, e);
    const u32 okm = (u32)Tb->lgn[pp & 0xFFu] | ((u32)Tb->lgn[pp >> 8] << 3);
    const int n = __popc(okm);
    const u32 w = agent_hash(r.seed_mix(), r.draws(), (u32)(c.lane_offset + game), c.key);
    if (c.agent_sample) {
        const int a6 = (int)__umulhi(w, 6u);
        aflag = a6 >= 3 ? 1 : 0;
        adir = a6 - 3 * aflag;
    } else if (n > 0) {
        const int slot = Tb->nth[okm * 8u + __umulhi(w, (u32)n)];
        aflag = slot < 3 ? (int)(e >> 15) : 0;
        adir = slot < 3 ? slot : slot - 3;
    }
}

// agent half of step() (envs/ewn.py:438-458): true if the opponent has to reply (then `dice` is the opponent's roll)
// slot: the game's board bytes in LDS, kept current (NULL: the caller encodes the board itself)
template <int S>
EWN_DEV bool roll_agent_half(const FastTab<S> *Tb, RState<S> &s, int aflag, int adir, int &dice, LaneRng &r, double R,
                             double &reward, int &term, int &trunc, int &info, int8_t *slot = nullptr)
{
    const int k = pk_cube(pk_sel<S>(Tb, s.posN, dice), aflag == 1);
    const int pb = pk_get(s.posN, k);
    const int q = Tb->nbn[adir][pb]; // no cube at all: byte 6 -> 255
    if (q == 255) { reward = -R; term = 1; trunc = 1; info = EWN_INFO_INVALID_PLAYER; return false; }
    if (slot) { // the agent's cube k is the real +(k + 1); source first: LDS stores of one wave land in program order
        const int cp = Tb->real_of_ring[pb & 63], cq = Tb->real_of_ring[q];
        slot[cp] = 0; slot[cq] = (int8_t)(k + 1);
    }
    rs_move<S, false>(s, k, q);
    if (q == Tb->ri_origin || s.P == 0) { reward = R; term = 1; info = EWN_INFO_WON; return false; }
    dice = r.randint(1, 7);
    return true;
}

// opponent half (envs/ewn.py:464-486) once its action (oflag, odir) is known
template <int S>
EWN_DEV void roll_opponent_half(const FastTab<S> *Tb, RState<S> &s, u32 e, int oflag, int odir, int &dice, LaneRng &r, double R,
                                double &reward, int &term, int &info, int8_t *slot = nullptr)
{
    const int k = pk_cube(e, oflag == 1);
    const int pb = pk_get(s.posP, k);
    const int q = Tb->nbp[odir][pb];
    if (slot) { const int cp = Tb->real_of_ring[pb & 63], cq = Tb->real_of_ring[q & 63]; slot[cp] = 0; slot[cq] = (int8_t)(-(k + 1)); }
    rs_move<S, true>(s, k, q);
    if (q == FastTab<S>::CELLS - 1 || s.N == 0) { reward = -R; term = 1; info = EWN_INFO_LOST; }
    else dice = r.randint(1, 7);
}

// AGENT 0: RandomAgent (the hash-driven uniform legal pick of ewn_step_out.random_action); 1: ExpectiMinimaxAgent of
// max_depth 1-4 ('hybrid'); 2: of max_depth 5-6.  OPP as in k_step_d3: 0 minimax max_depth 1-4, 1 RandomAgent, 2 minimax 5-6.
// H2: the opponent's search runs on the 'two_min_dist' table image (envs/minimax_ewn.py:133-178; a side's index is the sum of its two
// smallest distances, ewn_fast.hpp): RandomAgent / sample agents only
// TRJ 1: the trajectory is the record + the reward column and nothing else, known at compile time (see k_rollout_slots below)
template <int S, int T, int OPP, int RNGK, int AGENT, bool H2 = false, int TRJ = 0>
__global__ __launch_bounds__(D3_BS, ((OPP == 2 || AGENT == 2) ? 2 : 1)) void k_rollout_d3(RollCfg c, RollBuf B) // the max_depth 5 / 6 search: hold it to 256 registers (two waves per SIMD)
{
    constexpr int CELLS = S * S, GPB = D3_BS / T; // games per block
    constexpr int TS = T > 2 ? 2 : T;             // lanes per game the depth-5 search can use
    // LDS: boards | table image(s) | 16 bytes per game of decode scratch.  With the RandomAgent agent (one image) the block is a
    // STATIC array: its address is then a compile-time constant that folds into the offset field of every ds_read, where the
    // dynamic-LDS base is a link-time symbol the compiler adds with one VALU instruction in front of every table read (43 per
    // root of the search).  Two images (minimax agents of another depth class) can exceed the 64 KB static limit: dynamic.
    extern __shared__ __attribute__((aligned(16))) int8_t lds_dyn[];
    constexpr int LDS_STATIC = AGENT == 0 ? ((GPB * CELLS + 15) & ~15) + FAST_TAB_BYTES(S) + GPB * 16 + GPB * RecGeo<S>::STR : 16;
    __shared__ __attribute__((aligned(16))) int8_t lds_st[LDS_STATIC];
    int8_t *lds = AGENT == 0 ? lds_st : lds_dyn;
    int8_t *tb = lds + ((GPB * CELLS + 15) & ~15);
    tables_to_lds<FAST_TAB_BYTES(S)>(tb, (const int8_t *)B.tables); // LDS-DMA, waited for at the barrier
    const FastTab<S> *Tb = (const FastTab<S> *)tb;
    const FastTab<S> *Ta = Tb;
    int8_t *after = tb + FAST_TAB_BYTES(S);
    if (AGENT != 0 && B.agent_tables != B.tables) { // a second image for the agent's search (other depth class)
        tables_to_lds<FAST_TAB_BYTES(S)>(after, (const int8_t *)B.agent_tables);
        Ta = (const FastTab<S> *)after;
        after += FAST_TAB_BYTES(S);
    }
    uint8_t *garr = (uint8_t *)after;             // 16 bytes per game: d3_decode's scatter area
    int8_t *rec_slot = (int8_t *)garr + GPB * 16 + ((int)threadIdx.x / T) * RecGeo<S>::STR; // the game's record staging slot (B.t_rec)

    const int g0 = (int)blockIdx.x * GPB, ng = min(GPB, c.N - g0);
    const int gl = threadIdx.x / T, sub = threadIdx.x % T, game = g0 + gl;
    const bool live = game < c.N, writer = live && sub == 0;

    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1;
    bool frozen = true;
    if (live) {
        hdr = *rng_hdr_ptr(B.rng, game);
        dice = B.dice[game];
        frozen = B.done[game] != 0;
    }
    const bool frozen0 = frozen;
    block_copy_in(lds, B.board + (size_t)g0 * CELLS, ng * CELLS);
    LaneRng r; r.load(RNGK, hdr, rng_win_ptr(B.rng, c.N, c.W, live ? game : 0, RNGF_CUR(hdr.w)), c.W, c.key);
    r.begin_kernel();
    lds_dma_wait();
    __syncthreads();
    int8_t *mine = lds + gl * CELLS;
    RState<S> s;
    d3_decode<S, T>(live ? mine : lds, sub, garr + gl * 16, s);
    const bool want_board = TRJ == 0 && B.t_board != nullptr;
    double ret_acc = 0.0;
    int n_steps = 0, n_eps = 0, n_wins = 0;

#ifdef EWN_ROLLOUT_STAMPS
    unsigned long long st_acc[6] = { 0, 0, 0, 0, 0, 0 }, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory");
    const unsigned long long st_begin = st_prev;
#endif
    #pragma unroll 1
    for (int kstep = 0; kstep < c.K; kstep++) {
        const bool active = live && !frozen;
        double reward = 0.0;
        int term = 0, trunc = 0, info = EWN_INFO_NONE;
        if (live && frozen) term = 1; // a finished, un-reset game stays put (as in ewn_step)
        // ---- the agent's action for the current observation (the agent is the canonical BOTTOM_RIGHT side)
        int aflag = 0, adir = 0;
        if constexpr (AGENT == 0) {
            roll_stand_in_action<S>(Tb, s, dice, r, c, game, aflag, adir);
        } else {
            // ExpectiMinimaxAgent.predict(canonical observation): the agent's own position IS canonical for it once flipped
            const RState<S> f = rs_flip<S>(Ta, s);
            if constexpr (AGENT == 1) d3_search<S, T>(Ta, f, dice, sub, c.agent_depth, aflag, adir);
            else d5_dispatch<S, TS>(Ta, f, dice, T > 2 ? (sub & 1) : sub, aflag, adir);
        }
        if (active) { // a frozen lane's stream stays where its last step left it
            if constexpr (RNGK == 0) r.prefetch();
            r.begin_step();
            if constexpr (RNGK == 1) r.ps.prime();
        }
        RSTAMP(0); // agent's action + RNG block
        bool reply = false;
        if (active) reply = roll_agent_half<S>(Tb, s, aflag, adir, dice, r, c.reward, reward, term, trunc, info);
        RSTAMP(1); // agent half
        // the opponent's search: run by every lane (lanes without a pending reply compute on a harmless state)
        int oflag = 0, odir = 0;
        if constexpr (OPP == 0) d3_search<S, T, H2>(Tb, s, dice, sub, c.depth, oflag, odir);
        if constexpr (OPP == 2) d5_dispatch<S, TS, H2>(Tb, s, dice, T > 2 ? (sub & 1) : sub, oflag, odir);
        RSTAMP(2); // search
        if (reply) {
            // opponent half, envs/ewn.py:464-486
            const u32 e = pk_sel<S>(Tb, s.posP, dice);
            if constexpr (OPP == 1) {
                const u32 pp = pk_pair(s.posP, e);
                const u32 okm = (u32)Tb->lgp[pp & 0xFFu] | ((u32)Tb->lgp[pp >> 8] << 3);
                const int slot = Tb->nth[okm * 8u + (u32)r.randint(0, __popc(okm))]; // 0..5 = cube slot * 3 + dir
                oflag = slot < 3 ? (int)(e >> 15) : 0;
                odir = slot < 3 ? slot : slot - 3;
            }
            roll_opponent_half<S>(Tb, s, e, oflag, odir, dice, r, c.reward, reward, term, info);
        }
        if (active) {
            ret_acc += reward; n_steps++; n_eps += term; n_wins += info == EWN_INFO_WON ? 1 : 0;
            if (term) {
                if (c.autoreset) { // reset(seed = next_seed) + setup_game (envs/ewn.py:488-494, 94-108); Philox kind only (host check)
                    r.next_episode(B.rng, c.N, game, c.seed_stride, c.key, nullptr);
                    d3_init_state<S>(Tb, s);
                    dice = r.first_dice(6);
                } else frozen = true;
            }
        }
        RSTAMP(3); // opponent half + bookkeeping + auto-reset
        // ---- this step's row of the trajectory
        if constexpr (TRJ == 1) {
            if (writer) B.t_reward[(size_t)kstep * c.N + game] = reward;
        } else if (writer) {
            const size_t o = (size_t)kstep * c.N + game;
            if (B.t_action) ((uint16_t *)B.t_action)[o] = (uint16_t)((uint8_t)aflag | ((uint16_t)(uint8_t)adir << 8));
            if (B.t_dice) B.t_dice[o] = (int8_t)dice;
            if (B.t_reward) B.t_reward[o] = reward;
            if (B.t_term) B.t_term[o] = (uint8_t)term;
            if (B.t_trunc) B.t_trunc[o] = (uint8_t)trunc;
            if (B.t_info) B.t_info[o] = (uint8_t)info;
        }
        if (TRJ == 1 || B.t_rec) { // one aligned record per lane-step (ewn_rollout_out.record), the board re-encoded into the game's slot
            rec_slot_build<S, T>(Tb, s, sub, rec_slot);
            if (live) rec_store<S, T>(rec_slot, sub, dice, aflag, adir, term, trunc, info, B.t_rec + ((size_t)kstep * c.N + game) * RecGeo<S>::STR);
            __builtin_amdgcn_wave_barrier();
        }
        if (want_board) {
            // A wave's games are one contiguous, 16-byte aligned span of LDS (64 / T games x S*S bytes): the wave copies its own
            // span out and no block-wide barrier is needed, so the waves of a block drift apart freely (LDS operations of one
            // wave execute in program order).  The last, partial block of the grid and an unaligned row take the block copy.
            int8_t *row = B.t_board + ((size_t)kstep * c.N + g0) * CELLS;
            constexpr int WB = (64 / T) * CELLS;       // bytes per wave
            static_assert(WB % 16 == 0, "a wave's span of boards is a whole number of 16-byte pieces");
            if (ng == GPB && (((uintptr_t)row) & 15) == 0) {
                // zero the wave's span with 16-byte stores (one instruction for the wave instead of S*S / T byte stores per
                // lane), drop the <= 12 cube bytes of every game in, copy the span out
                const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
                uint4 *span = (uint4 *)(lds + wv * WB);
                __builtin_amdgcn_wave_barrier();
                #pragma unroll
                for (int i = 0; i < (WB / 16 + 63) / 64; i++) { const int j = i * 64 + ln; if (j < WB / 16) span[j] = make_uint4(0u, 0u, 0u, 0u); }
                __builtin_amdgcn_wave_barrier();
                d3_encode_cubes<S, T>(Tb, s, sub, mine);
                __builtin_amdgcn_wave_barrier();
                uint4 *dst = (uint4 *)(row + wv * WB);
                #pragma unroll
                for (int i = 0; i < (WB / 16 + 63) / 64; i++) { const int j = i * 64 + ln; if (j < WB / 16) dst[j] = span[j]; }
                __builtin_amdgcn_wave_barrier();
            } else {
                if (live) d3_encode<S, T>(Tb, s, sub, mine);
                __syncthreads();
                block_copy_out(row, lds, ng * CELLS);
                __syncthreads();
            }
        }
        RSTAMP(4); // trajectory row
    }
#ifdef EWN_ROLLOUT_STAMPS
    if (B.ret_sum && (threadIdx.x & 63) == 0) {
        unsigned long long *o = (unsigned long long *)B.ret_sum + (size_t)(blockIdx.x * (D3_BS / 64) + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 6; i++) o[i] = st_acc[i];
        o[6] = st_begin; o[7] = st_prev;
    }
#endif
    // ---- the state goes back to HBM once
    if (live) d3_encode<S, T>(Tb, s, sub, mine);
    if (writer) {
        if (!frozen0) { *rng_hdr_ptr(B.rng, game) = r.header(); B.dice[game] = (int8_t)dice; }
        B.done[game] = frozen ? 1 : 0;
#ifndef EWN_ROLLOUT_STAMPS
        if (B.ret_sum) B.ret_sum[game] += ret_acc;
#endif
        if (B.n_steps) B.n_steps[game] += n_steps;
        if (B.n_episodes) B.n_episodes[game] += n_eps;
        if (B.n_wins) B.n_wins[game] += n_wins;
    }
    __syncthreads();
    block_copy_out(B.board + (size_t)g0 * CELLS, lds, ng * CELLS);
}

// ---------------------------------------------------------------- slot-task rollout (RandomAgent agent, minimax opponent)
//
// In k_rollout_d3 every game of a wave walks all six root slots of the opponent's search each step, although the dice selects ONE
// cube (three roots) in ~85 % of the positions (measured on the bench's rollouts: 80 % of the decisions have exactly three legal
// moves, 11 % six): a wave always holds some game with a second cube, so the second half of the root loop is always executed and
// mostly wasted.  Here a loop iteration searches the three roots of ONE cube per game (d3_search / d5c_search in PERLANE mode);
// a game with a second cube takes one more iteration for the same env step (best / action carried over), the others go on to
// their next env step.  The games of a wave therefore drift apart by a few steps; every game still plays exactly K steps per
// launch and writes row k of the trajectory when it finishes its step k -- results are identical to k_rollout_d3's, bit for bit
// (tests/test_gpu_rollout.py).  A wave leaves the loop when its last game is done: ~K (1 + p) + 2 sqrt(K p (1 - p)) iterations
// for p = the share of two-cube decisions, against K iterations of twice the search length.
// TRJ: what the trajectory is, known at compile time or not.  0: whatever RollBuf's pointers say (every column optional: a dozen
// loop-invariant null tests, which the compiler keeps as 64-bit masks in SGPRs, spills to VGPR lanes and reads back with ~70
// v_readlane per iteration); 1: the record layout with its reward column and nothing else (what bench.py and the trainers ask for);
// 2: no trajectory at all, only the final state and the per-lane totals (tournaments).
template <int S, int T, int OPP, int RNGK, bool H2 = false, int TRJ = 0>
__global__ __launch_bounds__(D3_BS, (OPP == 2 ? 2 : 1)) void k_rollout_slots(RollCfg c, RollBuf B)
{
    static_assert(OPP == 0 || OPP == 2, "minimax opponents");
    static_assert(!(H2 && OPP == 2), "'two_min_dist' at max_depth 5 / 6 keeps the reference's loops (d5_search): lock-step kernel only");
    constexpr int CELLS = S * S, GPB = D3_BS / T;   // games per block
    constexpr int TS = T > 2 ? 2 : T;               // lanes per game the depth-5 search can use
    constexpr int STR = RecGeo<S>::STR;             // LDS bytes per game: the game's board slot, one trajectory record wide
    __shared__ __attribute__((aligned(16))) int8_t lds_st[GPB * STR + FAST_TAB_BYTES(S) + GPB * 16];
    int8_t *lds = lds_st;
    int8_t *tb = lds + GPB * STR;
    tables_to_lds<FAST_TAB_BYTES(S)>(tb, (const int8_t *)B.tables); // LDS-DMA, waited for at the barrier
    const FastTab<S> *Tb = (const FastTab<S> *)tb;
    uint8_t *garr = (uint8_t *)(tb + FAST_TAB_BYTES(S));

    const int g0 = (int)blockIdx.x * GPB, ng = min(GPB, c.N - g0);
    const int gl = threadIdx.x / T, sub = threadIdx.x % T, game = g0 + gl;
    const bool live = game < c.N, writer = live && sub == 0;

    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1;
    bool frozen = true;
    if (live) {
        hdr = *rng_hdr_ptr(B.rng, game);
        dice = B.dice[game];
        frozen = B.done[game] != 0;
    }
    const bool frozen0 = frozen;
    // boards come in at their packed stride (the state tensor's layout) and are decoded from there
    block_copy_in(lds, B.board + (size_t)g0 * CELLS, ng * CELLS);
    LaneRng r; r.load(RNGK, hdr, rng_win_ptr(B.rng, c.N, c.W, live ? game : 0, RNGF_CUR(hdr.w)), c.W, c.key);
    r.begin_kernel();
    lds_dma_wait();
    __syncthreads();
    RState<S> s;
    d3_decode<S, T>(live ? lds + gl * CELLS : lds, sub, garr + gl * 16, s);
    __syncthreads();                                // every game is in registers: the board area becomes the per-game slots
    int8_t *slot_b = lds + gl * STR;
    const bool want_slot = TRJ == 1 || (TRJ == 0 && (B.t_board != nullptr || B.t_rec != nullptr));
    int8_t *slot_m = want_slot ? slot_b : nullptr;  // kept current by the two halves of a step and by the auto-reset
    if (want_slot) rec_slot_build<S, T>(Tb, s, sub, slot_b);
    double ret_acc = 0.0;
    int n_steps = 0, n_eps = 0, n_wins = 0;

    int kdone = 0, phase = 0;                       // env steps finished; 0 = next iteration starts a step, 1 = it searches the second cube
    double reward = 0.0, best = 0.0;
    int term = 0, trunc = 0, info = EWN_INFO_NONE, aflag = 0, adir = 0, oflag = 0, odir = 0;
    bool reply = false;
#ifdef EWN_ROLLOUT_STAMPS
    unsigned long long st_acc[6] = { 0, 0, 0, 0, 0, 0 }, st_prev;   // [4] counts the iterations
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory");
    const unsigned long long st_begin = st_prev;
#endif
    while (true) {
        const bool pending = live && kdone < c.K;
        if (__builtin_amdgcn_ballot_w64(pending) == 0) break; // this wave's games have all played K steps
        const bool start = pending && phase == 0;
        const bool active = start && !frozen;
        if (start) {
            reward = 0.0; term = frozen ? 1 : 0; trunc = 0; info = EWN_INFO_NONE; reply = false; aflag = 0; adir = 0;
            roll_stand_in_action<S>(Tb, s, dice, r, c, game, aflag, adir);
            if (active) { // a frozen lane's stream stays where its last step left it
                if constexpr (RNGK == 0) r.prefetch();
                r.begin_step();
                if constexpr (RNGK == 1) r.ps.prime();
                reply = roll_agent_half<S>(Tb, s, aflag, adir, dice, r, c.reward, reward, term, trunc, info, slot_m);
            }
        }
        RSTAMP(0); // start of an env step: the agent's action, RNG block, agent half
        // one cube's three roots of the opponent's search, run by every lane (lanes without a pending reply compute on a harmless
        // state, so the DPP exchanges inside always see their partners)
        bool second = false;
        if constexpr (OPP == 0) best = d3_search<S, T, H2, true>(Tb, s, dice, sub, c.depth, oflag, odir, phase, best, &second);
        else best = d5c_search<S, TS, true>(Tb, s, dice, T > 2 ? (sub & 1) : sub, oflag, odir, phase, best, &second);
        RSTAMP(1); // search
        if (pending && phase == 0 && reply && second) phase = 1; // the same env step goes on with the second cube
        else if (pending) {
            phase = 0;
            if (reply) roll_opponent_half<S>(Tb, s, pk_sel<S>(Tb, s.posP, dice), oflag, odir, dice, r, c.reward, reward, term, info, slot_m);
            if (!frozen) {
                ret_acc += reward; n_steps++; n_eps += term; n_wins += info == EWN_INFO_WON ? 1 : 0;
                if (term) {
                    if (c.autoreset) { // reset(seed = next_seed) + setup_game (envs/ewn.py:488-494, 94-108); Philox kind only (host check)
                        r.next_episode(B.rng, c.N, game, c.seed_stride, c.key, nullptr);
                        d3_init_state<S>(Tb, s);
                        if (want_slot) rec_slot_init<S>(slot_b);
                        dice = r.first_dice(6);
                    } else frozen = true;
                }
            }
            // ---- row kdone of the trajectory
            RSTAMP(2); // opponent half + bookkeeping + auto-reset (stamps inside a divergent region: the wave's time while any lane is here)
            const size_t o = (size_t)kdone * c.N + game;
            if constexpr (TRJ == 1) {
                if (sub == 0) B.t_reward[o] = reward;
            } else if (TRJ == 0 && sub == 0) {
                if (B.t_action) ((uint16_t *)B.t_action)[o] = (uint16_t)((uint8_t)aflag | ((uint16_t)(uint8_t)adir << 8));
                if (B.t_dice) B.t_dice[o] = (int8_t)dice;
                if (B.t_reward) B.t_reward[o] = reward;
                if (B.t_term) B.t_term[o] = (uint8_t)term;
                if (B.t_trunc) B.t_trunc[o] = (uint8_t)trunc;
                if (B.t_info) B.t_info[o] = (uint8_t)info;
            }
            if (want_slot) __builtin_amdgcn_wave_barrier(); // this step's byte stores of every lane of the game are issued before its slot is read
            if (TRJ == 1 || (TRJ == 0 && B.t_rec)) rec_store<S, T>(slot_b, sub, dice, aflag, adir, term, trunc, info, B.t_rec + o * STR);
            if (TRJ == 0 && B.t_board) {
                // the game's board out of its LDS slot (LDS operations of one wave execute in program order; the T lanes of a game
                // are in one wave)
                // copy out: each of the T lanes a contiguous run of whole dwords, the last lane the odd tail.  The row sits at a
                // byte offset of o * CELLS: an UNALIGNED destination, which global memory accepts (the compiler knows: a 4-byte
                // memcpy to a char pointer becomes one dword store on this target); a quarter of the requests of a byte copy
                constexpr int CH = ((CELLS / T + 3) & ~3) < CELLS ? ((CELLS / T + 3) & ~3) : (CELLS & ~3); // bytes per lane, whole dwords
                int8_t *dst = B.t_board + o * CELLS;
                #pragma unroll
                for (int i = 0; i < CH; i += 4) {
                    const int at = sub * CH + i;
                    if (at + 4 <= CELLS) { const u32 w = *(const u32 *)(slot_b + at); __builtin_memcpy(dst + at, &w, 4); }
                }
                if (sub == T - 1) {
                    #pragma unroll
                    for (int i = (T * CH < CELLS ? T * CH : (CELLS & ~3)); i < CELLS; i++) dst[i] = slot_b[i];
                }
            }
            kdone++;
        }
        RSTAMP(3); // trajectory row
#ifdef EWN_ROLLOUT_STAMPS
        st_acc[4]++;
#endif
    }
#ifdef EWN_ROLLOUT_STAMPS
    if (B.ret_sum && (threadIdx.x & 63) == 0) {
        unsigned long long *o = (unsigned long long *)B.ret_sum + (size_t)(blockIdx.x * (D3_BS / 64) + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 6; i++) o[i] = st_acc[i];
        o[6] = st_begin; o[7] = st_prev;
    }
#endif
    // ---- the state goes back to HBM once, through the packed board area
    __syncthreads();
    if (live) d3_encode<S, T>(Tb, s, sub, lds + gl * CELLS);
    if (writer) {
        if (!frozen0) { *rng_hdr_ptr(B.rng, game) = r.header(); B.dice[game] = (int8_t)dice; }
        B.done[game] = frozen ? 1 : 0;
#ifndef EWN_ROLLOUT_STAMPS
        if (B.ret_sum) B.ret_sum[game] += ret_acc;
#endif
        if (B.n_steps) B.n_steps[game] += n_steps;
        if (B.n_episodes) B.n_episodes[game] += n_eps;
        if (B.n_wins) B.n_wins[game] += n_wins;
    }
    __syncthreads();
    block_copy_out(B.board + (size_t)g0 * CELLS, lds, ng * CELLS);
}
