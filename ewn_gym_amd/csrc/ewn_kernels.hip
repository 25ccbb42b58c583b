// ewn_kernels.hip -- the generic gfx950 kernels (every board size / cube layer / opponent / depth / heuristic /
// reward shaping), the stateless policy and rule queries, and the C ABI (include/ewn_hip.h) of libewn_hip.so.
// The headline kernel (depth-3 'hybrid' or random opponent, cube_layer 3) lives in ewn_step_d3.hpp.
//
// Layout in HBM: board int8 [N][S*S] (one contiguous row per lane, so a block of 256 lanes is one contiguous
// 6.4 KB (5x5) / 12.5 KB (7x7) span copied to and from LDS in coalesced 16-byte pieces), dice int8 [N], done u8 [N],
// rng u32 (headers [N][4], then for the MT kind windows [N][3][W] and epochs [N]).  In these generic kernels one thread
// owns one lane (game); the wavefront is the unit of scheduling, not the unit of work -- a 25-cell board would leave
// 39 of 64 lanes idle if a whole wavefront served one game.
#include "ewn_host.hpp"
#include "ewn_playout.hpp"

#include "ewn_lds.hpp"
#include "ewn_step_d3.hpp"
#include "ewn_rollout.hpp"

// ---------------------------------------------------------------- per-lane pieces

// reset(seed) + setup_game, envs/ewn.py:488-494, 94-108, for an explicitly given seed: builds the window of this
// episode and, ahead of time, of the next two (seed + stride, seed + 2 stride)
template <int NW>
EWN_DEV void lane_reset(const Geom &g, const KCfg &c, u32 *rng, int lane, u32 seed, GState<NW> &s, int &dice, LaneRng &r)
{
    u32 f = 0;
    if (c.rng_kind == 0) {
        for (u32 j = 0; j < 3; j++) mt_fill_window(seed + j * c.seed_stride, (int)c.W, rng_win_ptr(rng, c.N, c.W, lane, j));
        f = rngf_make(0u, 0u, 2u, 0u, 0u);
        *rng_epoch_ptr(rng, c.N, c.W, lane) += 1u; // refills queued for this lane before the reset are now stale
    }
    r.load(c.rng_kind, make_uint4(seed, 0u, seed + c.seed_stride, f), rng_win_ptr(rng, c.N, c.W, lane, 0u), c.W, c.key);
    init_state<NW>(g, s);
    dice = r.first_dice(g.CN); // roll_dice :90-91
}

// the auto-reset inside a step: next_seed becomes the episode seed (the freed window is rebuilt by k_mt_refill afterwards)
template <int NW>
EWN_DEV void lane_auto_reset(const Geom &g, const KCfg &c, u32 *rng, int lane, GState<NW> &s, int &dice, LaneRng &r)
{
    r.next_episode(rng, c.N, lane, c.seed_stride, c.key, nullptr);
    init_state<NW>(g, s);
    dice = r.first_dice(g.CN);
}

// Generic step path: right after the step kernel, rebuild the window slots it freed (header field X > 0) and count
// them ready.  One thread per lane, coalesced header reads, waves without a flagged lane leave at once.
#define REFILL_BS 64
__global__ __launch_bounds__(REFILL_BS) void k_mt_refill(u32 *rng, int N, u32 W, u32 stride)
{
    extern __shared__ u32 sm[]; // (W + 1) rows of 65 words
    const int t = threadIdx.x, lane = blockIdx.x * REFILL_BS + t;
    uint4 h = make_uint4(0u, 0u, 0u, 0u);
    if (lane < N) h = *rng_hdr_ptr(rng, lane);
    const u32 x = RNGF_X(h.w);
    if (!__any(x != 0u)) return;
    if (x == 0u) return;
    const u32 cur = RNGF_CUR(h.w);
    for (u32 j = 1; j <= x; j++) // the slot freed j resets ago will serve episode e + 3 - j
        mt_window_lds(h.x + (3u - j) * stride, W, sm, t, rng_win_ptr(rng, N, W, lane, (cur + 3u - j) % 3u));
    rng_hdr_ptr(rng, lane)->w = rngf_make(h.w, cur, min(2u, RNGF_READY(h.w) + x), 0u, RNGF_Y(h.w));
}

struct StepRes { double reward; int term, trunc, info; };

// Agent half of step(): envs/ewn.py:438-458 and training_ewn.py:44-66.
// Returns true when the opponent must still reply.
template <int NW>
EWN_DEV bool step_agent(const Geom &g, const KCfg &c, GState<NW> &s, int &dice, int flag, int dir, LaneRng &r,
                           int32_t *tol, StepRes &o)
{
    o.reward = 0.0; o.term = 0; o.trunc = 0; o.info = EWN_INFO_NONE;
    const CubeSel cs = select_cubes(s.aliveP, dice);
    const int k = cube_to_move(cs, flag == 1);
    const bool valid = k >= 0 && dir >= 0 && dir <= 2 && dir_ok<0>(g, pos_of<0>(s, k), dir);
    if (!valid) {
        if (c.shaped) {
            const int t = *tol - 1;
            *tol = t;
            if (t <= 0) { o.reward = -c.reward; o.term = 1; o.trunc = 1; o.info = EWN_INFO_INVALID_PLAYER; }
            else { o.reward = c.illegal_reward; o.info = EWN_INFO_TOLERANCE; }
        } else { o.reward = -c.reward; o.term = 1; o.trunc = 1; o.info = EWN_INFO_INVALID_PLAYER; }
        return false;
    }
    apply_move<0, NW>(g, s, k, dir);
    if (is_win<NW>(g, s)) { o.reward = c.reward; o.term = 1; o.info = EWN_INFO_WON; return false; }
    dice = r.randint(1, g.CN + 1); // the opponent's dice, :458
    return true;
}

// Opponent half: envs/ewn.py:464-486, training_ewn.py:75-99.
template <int NW>
EWN_DEV void step_opponent(const Geom &g, const KCfg &c, GState<NW> &s, int &dice, int oflag, int odir, LaneRng &r,
                              double *prev_score, StepRes &o)
{
    const CubeSel cs = select_cubes(s.aliveN, dice);
    const int k = cube_to_move(cs, oflag == 1);
    const bool valid = k >= 0 && odir >= 0 && odir <= 2 && dir_ok<1>(g, pos_of<1>(s, k), odir);
    if (!valid) { o.reward = 0.0; o.term = 1; o.trunc = 1; o.info = EWN_INFO_INVALID_OPP; return; }
    apply_move<1, NW>(g, s, k, odir);
    if (is_win<NW>(g, s)) { o.reward = -c.reward; o.term = 1; o.info = EWN_INFO_LOST; return; }
    dice = r.randint(1, g.CN + 1); // :483
    if (c.shaped) {
        const double cur = evaluate<NW>(g, s, EWN_H_HYBRID);
        o.reward = cur - *prev_score;
        *prev_score = cur;
    }
}

// RandomAgent.predict on the live env (classical_policies/random_policy.py:11-15):
// uniform index into BOTTOM_RIGHT's legal list, drawn from the lane's own stream.
template <int NW>
EWN_DEV void policy_random(const Geom &g, const GState<NW> &s, int dice, LaneRng &r, int &oflag, int &odir)
{
    const int n = for_each_legal<1, NW>(g, s, dice, [](int, int, int) { return true; });
    const int pick = r.randint(0, n);
    int i = 0;
    oflag = 0; odir = 0;
    for_each_legal<1, NW>(g, s, dice, [&](int flag, int, int dir) { if (i == pick) { oflag = flag; odir = dir; } i++; return i <= pick; });
}

template <int NW, int DEPTH>
__device__ void policy_minimax(const Geom &g, const GState<NW> &s, int dice, int heur, int &oflag, int &odir, double *value)
{
    const GState<NW> cst = canonicalize<NW>(g, s); // the policy always plays TOP_LEFT (envs/ewn.py:291-295)
    oflag = 0; odir = 0;
    EvalLeaf<NW> leaf = { heur };
    const double v = search<NW, DEPTH, 0, true>(g, cst, dice, -__builtin_inf(), __builtin_inf(), leaf, oflag, odir);
    if (value) *value = v;
}

template <int NW>
__device__ void policy_minimax_rt(const Geom &g, const GState<NW> &s, int dice, int depth, int heur, int &oflag, int &odir)
{
    switch (depth) {
    case 1: policy_minimax<NW, 1>(g, s, dice, heur, oflag, odir, nullptr); break;
    case 2: policy_minimax<NW, 2>(g, s, dice, heur, oflag, odir, nullptr); break;
    case 3: policy_minimax<NW, 3>(g, s, dice, heur, oflag, odir, nullptr); break;
    case 4: policy_minimax<NW, 4>(g, s, dice, heur, oflag, odir, nullptr); break;
    case 5: policy_minimax<NW, 5>(g, s, dice, heur, oflag, odir, nullptr); break;
    default: policy_minimax<NW, 6>(g, s, dice, heur, oflag, odir, nullptr); break;
    }
}

// ---------------------------------------------------------------- reset / init kernels

template <int NW>
__global__ __launch_bounds__(BS) void k_reset(Geom g, KCfg c, KState st, const u32 *seeds, const uint8_t *mask)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int lane0 = blockIdx.x * BS, nl = min(BS, c.N - lane0), lane = lane0 + threadIdx.x;
    block_copy_in(lds, st.board + (size_t)lane0 * g.cells, nl * g.cells);
    __syncthreads();
    if (lane < c.N && (!mask || mask[lane])) {
        uint4 *hp = rng_hdr_ptr(st.rng, lane);
        const u32 seed = seeds ? seeds[lane] : hp->z;
        GState<NW> s; int dice; LaneRng r;
        lane_reset<NW>(g, c, st.rng, lane, seed, s, dice, r);
        *hp = r.header();
        st.dice[lane] = (int8_t)dice;
        st.done[lane] = 0;
        if (c.shaped && c.refresh && st.prev_score) st.prev_score[lane] = evaluate<NW>(g, s, EWN_H_HYBRID);
        encode_board<NW>(g, s, lds + threadIdx.x * g.cells);
    }
    __syncthreads();
    block_copy_out(st.board + (size_t)lane0 * g.cells, lds, nl * g.cells);
}

template <int NW>
__global__ __launch_bounds__(BS) void k_init_aux(Geom g, KCfg c, KState st, int tol0)
{
    const int lane = blockIdx.x * BS + threadIdx.x;
    if (lane >= c.N) return;
    st.done[lane] = 0;
    *rng_hdr_ptr(st.rng, lane) = make_uint4(0u, 0u, 0u, 0u);
    if (st.prev_score) {
        GState<NW> s;
        init_state<NW>(g, s);
        st.prev_score[lane] = evaluate<NW>(g, s, EWN_H_HYBRID); // training_ewn.py:35
    }
    if (st.tolerance) st.tolerance[lane] = tol0;
}

// roll_dice (envs/ewn.py:90-92): dice_roll = np.random.randint(1, cube_num + 1) on the lane's own stream -- the one draw, nothing else
// of the game changes.  A lane whose game is finished and not yet reset keeps its (stale) dice, as the terminal observation does upstream.
__global__ __launch_bounds__(BS) void k_roll_dice(Geom g, KCfg c, KState st, const uint8_t *mask)
{
    const int lane = blockIdx.x * BS + threadIdx.x;
    if (lane >= c.N || (mask && !mask[lane]) || st.done[lane]) return;
    const uint4 hdr = *rng_hdr_ptr(st.rng, lane);
    LaneRng r; r.load(c.rng_kind, hdr, rng_win_ptr(st.rng, c.N, c.W, lane, RNGF_CUR(hdr.w)), c.W, c.key);
    st.dice[lane] = (int8_t)r.randint(1, g.CN + 1);
    *rng_hdr_ptr(st.rng, lane) = r.header();
}

// ---------------------------------------------------------------- step kernels

// PHASE 0: fused (agent + in-thread opponent policy + finish)
// PHASE 1: agent half only; a lane that still needs the opponent's reply leaves its
//          canonical observation in scratch for a policy kernel (sc.phase[lane] = 1)
// PHASE 2: opponent half for the lanes with sc.phase[lane] == 1, action taken from scratch
// FAST = board size S of the specialised depth-3 policy (ewn_fast.hpp), 0 = generic policies
template <int NW, int PHASE, int FAST>
__global__ __launch_bounds__(BS) void k_step(Geom g, KCfg c, KState st, const int8_t *actions, KOut out, KScratch sc)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *lds_t = lds + BS * g.cells; // terminal-observation staging
    const int lane0 = blockIdx.x * BS, nl = min(BS, c.N - lane0), lane = lane0 + threadIdx.x;
    const bool live = lane < c.N;
    [[maybe_unused]] const FastTab<(FAST ? FAST : 5)> *ftab = nullptr;
    if constexpr (FAST != 0) { // LDS-DMA, waited for only at the barrier below
        int8_t *tb = lds + ((2 * BS * g.cells + 15) & ~15);
        tables_to_lds<FAST_TAB_BYTES(FAST ? FAST : 5)>(tb, (const int8_t *)st.tables);
        ftab = (const FastTab<(FAST ? FAST : 5)> *)tb;
    }
    // every per-lane input is requested up front so the round trips overlap
    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1, aflag = 0, adir = 0;
    bool frozen = false, pending = false;
    if (live) {
        hdr = *rng_hdr_ptr(st.rng, lane);
        dice = st.dice[lane];
        frozen = st.done[lane] != 0;
        if (PHASE != 2) { const uint16_t a2 = ((const uint16_t *)actions)[lane]; aflag = (int8_t)(a2 & 0xff); adir = (int8_t)(a2 >> 8); }
        else { pending = sc.phase[lane] != 0; const uint16_t a2 = ((const uint16_t *)sc.act)[lane]; aflag = (int8_t)(a2 & 0xff); adir = (int8_t)(a2 >> 8); }
    }
    block_copy_in(lds, st.board + (size_t)lane0 * g.cells, nl * g.cells);
    if (PHASE == 2 && out.tboard) block_copy_in(lds_t, out.tboard + (size_t)lane0 * g.cells, nl * g.cells);
    if constexpr (FAST != 0) lds_dma_wait();
    __syncthreads();
    if (live) {
        int8_t *mine = lds + threadIdx.x * g.cells, *mine_t = lds_t + threadIdx.x * g.cells;
        const bool skip = PHASE == 2 && (frozen || !pending); // settled by the pre phase
        if (!skip) {
            StepRes o; o.reward = 0.0; o.term = 0; o.trunc = 0; o.info = EWN_INFO_NONE;
            bool settled = true; // this launch produces the lane's step result
            if (frozen) {
                // no reference counterpart: stepping a finished game is undefined upstream; the lane stays put
                o.term = 1;
                if (PHASE == 1) {
                    sc.phase[lane] = 0;
                    if (c.opp == EWN_OPP_MCTS) for (int i = 0; i < 6; i++) sc.wins[(size_t)lane * 6 + i] = -1;
                }
                if (out.tboard) for (int i = 0; i < g.cells; i++) mine_t[i] = mine[i];
                if (out.tdice) out.tdice[lane] = (int8_t)dice;
                if (out.ract) ((uint16_t *)out.ract)[lane] = 0;
            } else {
                LaneRng r; r.load(c.rng_kind, hdr, rng_win_ptr(st.rng, c.N, c.W, lane, RNGF_CUR(hdr.w)), c.W, c.key);
                if (PHASE != 2) r.begin_kernel();
                r.prefetch();
                if (PHASE != 2) r.begin_step();
                GState<NW> s;
                decode_board<NW>(g, mine, s);
                bool reply;
                int oflag = aflag, odir = adir;
                if (PHASE != 2) reply = step_agent<NW>(g, c, s, dice, aflag, adir, r, st.tolerance ? st.tolerance + lane : nullptr, o);
                else reply = true;
                if (PHASE == 1) {
                    sc.phase[lane] = reply ? 1 : 0;
                    int n_root = 0;
                    if (reply) {
                        const GState<NW> cst = canonicalize<NW>(g, s);
                        encode_board<NW>(g, cst, sc.cboard + (size_t)lane * g.cells);
                        sc.cdice[lane] = (int8_t)dice;
                        sc.obs_id[lane] = r.seed_mix() * 0x9E3779B1u + r.draws();
                        settled = false;
                        if (c.opp == EWN_OPP_MCTS) n_root = for_each_legal<0, NW>(g, cst, dice, [](int, int, int) { return true; });
                    }
                    // flat Monte-Carlo opponent: the win counters of the root moves start here (0 = a move to play out, -1 = no
                    // such move / no reply pending) -- what k_mcts_init does for the stateless policy, without its launch
                    if (c.opp == EWN_OPP_MCTS) for (int i = 0; i < 6; i++) sc.wins[(size_t)lane * 6 + i] = i < n_root ? 0 : -1;
                } else if (reply) {
                    if (PHASE == 2 && c.opp == EWN_OPP_MCTS) {
                        // MctsAgent's choice, classical_policies/mcts.py:68: np.argmax over the root moves' wins (first maximum),
                        // then that entry of the legal list of the canonical observation -- k_mcts_pick without its launch
                        int best = 0, bw = -1;
                        for (int i = 0; i < 6; i++) { const int w = sc.wins[(size_t)lane * 6 + i]; if (w > bw) { bw = w; best = i; } }
                        const GState<NW> cst = canonicalize<NW>(g, s);
                        int j = 0;
                        oflag = 0; odir = 0;
                        for_each_legal<0, NW>(g, cst, dice, [&](int flag, int, int dir) { if (j == best) { oflag = flag; odir = dir; } j++; return j <= best; });
                    }
                    if constexpr (FAST != 0) {
                        const GState<1> cst = canonicalize<1>(g, s);
                        if (c.heur == EWN_H_TWO_MIN_DIST) fast_d3<(FAST ? FAST : 5), true>(ftab, cst, dice, c.depth, oflag, odir);
                        else fast_d3<(FAST ? FAST : 5), false>(ftab, cst, dice, c.depth, oflag, odir);
                    } else if (PHASE == 0) {
                        if (c.opp == EWN_OPP_RANDOM) policy_random<NW>(g, s, dice, r, oflag, odir);
                        else policy_minimax_rt<NW>(g, s, dice, c.depth, c.heur, oflag, odir);
                    }
                    step_opponent<NW>(g, c, s, dice, oflag, odir, r, st.prev_score ? st.prev_score + lane : nullptr, o);
                }
                if (settled) {
                    if (out.tboard) encode_board<NW>(g, s, mine_t);
                    if (out.tdice) out.tdice[lane] = (int8_t)dice;
                    if (o.term) {
                        if (c.autoreset) {
                            lane_auto_reset<NW>(g, c, st.rng, lane, s, dice, r);
                            if (c.shaped && c.refresh && st.prev_score) st.prev_score[lane] = evaluate<NW>(g, s, EWN_H_HYBRID);
                        } else st.done[lane] = 1;
                    }
                }
                *rng_hdr_ptr(st.rng, lane) = r.header();
                encode_board<NW>(g, s, mine);
                st.dice[lane] = (int8_t)dice;
                if (settled && out.ract) { // RandomAgent.predict on the post-step observation
                    int f = 0, d = 0;
                    if (!(o.term && !c.autoreset)) {
                        const int n = for_each_legal<0, NW>(g, s, dice, [](int, int, int) { return true; });
                        if (n > 0) {
                            const int pick = (int)__umulhi(agent_hash(r.seed_mix(), r.draws(), (u32)(c.lane_offset + lane), c.key), (u32)n);
                            int i = 0;
                            for_each_legal<0, NW>(g, s, dice, [&](int flag, int, int dir) { if (i == pick) { f = flag; d = dir; } i++; return i <= pick; });
                        }
                    }
                    ((uint16_t *)out.ract)[lane] = (uint16_t)((uint8_t)f | ((uint16_t)(uint8_t)d << 8));
                }
            }
            if (settled) {
                out.reward[lane] = o.reward; out.terminated[lane] = (uint8_t)o.term;
                out.truncated[lane] = (uint8_t)o.trunc; out.info[lane] = (uint8_t)o.info;
            }
        }
    }
    __syncthreads();
    block_copy_out(st.board + (size_t)lane0 * g.cells, lds, nl * g.cells);
    if (out.tboard) block_copy_out(out.tboard + (size_t)lane0 * g.cells, lds_t, nl * g.cells);
}

// ---------------------------------------------------------------- stateless queries

template <int NW>
__global__ __launch_bounds__(BS) void k_legal(Geom g, int M, const int8_t *boards, const int8_t *dice, int player, int8_t *acts,
                                              int8_t *n_acts, int8_t *cube_small, int8_t *cube_large, uint8_t *win)
{
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    const int d = dice[m];
    if (win) win[m] = is_win<NW>(g, s) ? 1 : 0;
    const u32 alive = player == 1 ? s.aliveP : s.aliveN;
    int n = 0;
    int8_t la[12];
    for (int i = 0; i < 12; i++) la[i] = -1;
    if (alive != 0 && d >= 1 && d <= g.CN) {
        auto rec = [&](int flag, int, int dir) { for (int i = 0; i < 6; i++) if (i == n) { la[2 * i] = (int8_t)flag; la[2 * i + 1] = (int8_t)dir; } n++; return true; };
        if (player == 1) for_each_legal<0, NW>(g, s, d, rec); else for_each_legal<1, NW>(g, s, d, rec);
        const CubeSel cs = select_cubes(alive, d);
        if (cube_small) cube_small[m] = (int8_t)(cube_to_move(cs, false) + 1);
        if (cube_large) cube_large[m] = (int8_t)(cube_to_move(cs, true) + 1);
    } else {
        if (cube_small) cube_small[m] = 0;
        if (cube_large) cube_large[m] = 0;
    }
    if (acts) for (int i = 0; i < 12; i++) acts[(size_t)m * 12 + i] = la[i];
    if (n_acts) n_acts[m] = (int8_t)n;
}

template <int SIDE, int NW> EWN_DEV void rollout_ply(const Geom &g, GState<NW> &s, PlayoutRng &ps);

template <int NW>
__global__ __launch_bounds__(BS) void k_apply_action(Geom g, int M, const int8_t *boards, const int8_t *dice, int player,
                                                     const int8_t *actions, int8_t *new_boards, uint8_t *valid)
{
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    const int d = dice[m], flag = actions[2 * m], dir = actions[2 * m + 1];
    const u32 alive = player == 1 ? s.aliveP : s.aliveN;
    bool ok = false;
    if (alive != 0 && d >= 1 && d <= g.CN && dir >= 0 && dir <= 2) {
        const int k = cube_to_move(select_cubes(alive, d), flag == 1);
        if (player == 1) { ok = k >= 0 && dir_ok<0>(g, pos_of<0>(s, k), dir); if (ok) apply_move<0, NW>(g, s, k, dir); }
        else { ok = k >= 0 && dir_ok<1>(g, pos_of<1>(s, k), dir); if (ok) apply_move<1, NW>(g, s, k, dir); }
    }
    encode_board<NW>(g, s, new_boards + (size_t)m * g.cells);
    if (valid) valid[m] = ok ? 1 : 0;
}

// MinimaxEnv.simulate, envs/minimax_ewn.py:215-238: one thread per (position, playout); any cube_layer
template <int NW>
__global__ __launch_bounds__(BS) void k_playout_wins(Geom g, int M, int n_sims, const int8_t *boards, int first_player, u64 key,
                                                     int32_t *wins)
{
    const long long idx = (long long)blockIdx.x * BS + threadIdx.x;
    if (idx >= (long long)M * n_sims) return;
    const int m = (int)(idx / n_sims), r = (int)(idx % n_sims);
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    PlayoutRng ps;
    ps.seed(PlayoutRng::obs_word((u32)m, 0x53494D55u, key), (u32)r);
    int cur = first_player == 1 ? 0 : 1;
    for (int ply = 0; ply < 1024 && !is_win<NW>(g, s); ply++) {
        if (cur == 0) rollout_ply<0, NW>(g, s, ps); else rollout_ply<1, NW>(g, s, ps);
        cur ^= 1;
    }
    if (top_left_won<NW>(g, s)) atomicAdd(&wins[m], 1);
}

// the same for cube_layer <= 3 (ewn_playout.hpp): an aligned group of 2^gl lanes per position, lane t plays playouts
// t, t + 2^gl, ... back to back
__global__ __launch_bounds__(BS) void k_playout_wins_lean(Geom g, int M, int n_sims, int gl, const int8_t *boards, int first_player,
                                                          u64 key, int32_t *wins)
{
    __shared__ PlayTab T;
    __shared__ int next[BS / 8];
    __shared__ u32 gw[BS / 8][4];
    playtab_build(&T, g.S);
    __syncthreads();
    const long long idx = (long long)blockIdx.x * BS + threadIdx.x;
    const int tc = 1 << gl, lane = (int)(idx & (tc - 1));
    const long long m = idx >> gl;
    int *nx = &next[threadIdx.x >> gl];
    if (lane == 0) *nx = tc; // same wave as the lanes that read it: LDS operations of a wave execute in order
    int w = 0;
    if (m < M) {
        const PState b0 = pstate_load(g, boards + (size_t)m * g.cells, lane, tc, gw[threadIdx.x >> gl]);
        bool tl;
        if (pstate_is_win(b0, g.S, tl)) w = tl && lane < n_sims ? (n_sims - lane + tc - 1) >> gl : 0;
        else {
            const u32 word = PlayoutRng::obs_word((u32)m, 0x53494D55u, key);
            w = first_player == 1 ? run_playouts<0>(&T, b0, g.S, word, 0u, lane, n_sims, nx)
                                  : run_playouts<1>(&T, b0, g.S, word, 0u, lane, n_sims, nx);
        }
    }
    for (int off = tc >> 1; off > 0; off >>= 1) w += __shfl_down(w, off, tc); // groups never straddle a wave
    if (lane == 0 && m < M && w) atomicAdd(&wins[m], w);
}

template <int NW>
__global__ __launch_bounds__(BS) void k_evaluate(Geom g, int M, const int8_t *boards, int heur, double *out)
{
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    out[m] = evaluate<NW>(g, s, heur);
}

template <int NW, int DEPTH>
__global__ __launch_bounds__(BS) void k_predict_minimax(Geom g, int M, const int8_t *boards, const int8_t *dice, int heur,
                                                        int8_t *actions, double *values)
{
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    int f = -1, d = -1;
    double v = 0.0;
    const int dc = dice[m];
    if (s.aliveP != 0 && dc >= 1 && dc <= g.CN) {
        EvalLeaf<NW> leaf = { heur };
        v = search<NW, DEPTH, 0, true>(g, s, dc, -__builtin_inf(), __builtin_inf(), leaf, f, d);
    }
    actions[2 * m] = (int8_t)f; actions[2 * m + 1] = (int8_t)d;
    if (values) values[m] = v;
}

template <int S, bool H2>
__global__ __launch_bounds__(BS) void k_predict_minimax_fast(Geom g, int M, const int8_t *boards, const int8_t *dice, int depth,
                                                             int8_t *actions, double *values, const void *tables)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    tables_to_lds<FAST_TAB_BYTES(S)>(lds, (const int8_t *)tables);
    lds_dma_wait();
    __syncthreads();
    const FastTab<S> *T = (const FastTab<S> *)lds;
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    GState<1> s;
    decode_board<1>(g, boards + (size_t)m * g.cells, s);
    int f = -1, d = -1;
    double v = 0.0;
    const int dc = dice[m];
    if (s.aliveP != 0 && dc >= 1 && dc <= g.CN) {
        if (is_win<1>(g, s)) v = evaluate<1>(g, s, EWN_H_HYBRID);
        else v = fast_d3<S, H2>(T, s, dc, depth, f, d);
    }
    actions[2 * m] = (int8_t)f; actions[2 * m + 1] = (int8_t)d;
    if (values) values[m] = v;
}

template <int NW>
__global__ __launch_bounds__(BS) void k_predict_random(Geom g, int M, const int8_t *boards, const int8_t *dice, u64 key, u32 step,
                                                       const u32 *step_dev, int lane_offset, int8_t *actions)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    if (step_dev) step += *step_dev;
    const int m0 = blockIdx.x * BS, nm = min(BS, M - m0), m = m0 + threadIdx.x;
    block_copy_in(lds, boards + (size_t)m0 * g.cells, nm * g.cells); // coalesced 16-byte pieces instead of 25 byte loads per lane
    const int d = m < M ? dice[m] : 1;
    __syncthreads();
    if (m >= M) return;
    GState<NW> s;
    decode_board<NW>(g, lds + threadIdx.x * g.cells, s);
    int f = 0, dr = 0;
    if (s.aliveP != 0 && d >= 1 && d <= g.CN) {
        const int n = for_each_legal<0, NW>(g, s, d, [](int, int, int) { return true; });
        u32 o[4];
        philox4x32_10(step, (u32)(lane_offset + m), 0x41474E54u, 0u, (u32)key, (u32)(key >> 32), o);
        const int pick = (int)__umulhi(o[0], (u32)n);
        int i = 0;
        for_each_legal<0, NW>(g, s, d, [&](int flag, int, int dir) { if (i == pick) { f = flag; dr = dir; } i++; return i <= pick; });
    }
    ((uint16_t *)actions)[m] = (uint16_t)((uint8_t)f | ((uint16_t)(uint8_t)dr << 8));
}

// ---------------------------------------------------------------- flat Monte-Carlo ("MCTS")

template <int NW>
__global__ __launch_bounds__(BS) void k_mcts_init(Geom g, int M, const int8_t *boards, const int8_t *dice, const uint8_t *active,
                                                  int32_t *wins)
{
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    int n = 0;
    if (!active || active[m]) {
        GState<NW> s;
        decode_board<NW>(g, boards + (size_t)m * g.cells, s);
        const int d = dice[m];
        if (s.aliveP != 0 && d >= 1 && d <= g.CN && !is_win<NW>(g, s))
            n = for_each_legal<0, NW>(g, s, d, [](int, int, int) { return true; });
    }
    for (int i = 0; i < 6; i++) wins[(size_t)m * 6 + i] = i < n ? 0 : -1;
}

// One uniformly random legal move of SIDE (classical_policies/mcts.py:29-35): dice, legal list in the reference's
// order as a 6-bit mask (larger-neighbour cube's dirs, then smaller-neighbour cube's), uniform pick, apply.
template <int SIDE, int NW>
EWN_DEV void rollout_ply(const Geom &g, GState<NW> &s, PlayoutRng &ps)
{
    u32 d0, frac;
    ps.draw(d0, frac);
    const int d = 1 + (int)d0; // random.randint(1, 6), mcts.py:29
    const CubeSel cs = select_cubes(alive_of<SIDE>(s), d);
    const bool have0 = cs.exact || cs.has_up, have1 = !cs.exact && cs.has_down;
    const int k0 = cs.exact ? cs.k_exact : cs.k_up, k1 = cs.k_down;
    const int p0 = pos_of<SIDE>(s, have0 ? k0 : 0), p1 = pos_of<SIDE>(s, have1 ? k1 : 0);
    u32 okm = 0;
    #pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        okm |= ((have0 && dir_ok<SIDE>(g, p0, dir)) ? 1u : 0u) << dir;
        okm |= ((have1 && dir_ok<SIDE>(g, p1, dir)) ? 1u : 0u) << (3 + dir);
    }
    const int n = __popc(okm);
    const int pick = (int)PlayoutRng::pick(frac, (u32)n);
    u32 m = okm;
    for (int i = 0; i < pick; i++) m &= m - 1;
    const int slot = __ffs((int)m) - 1;
    if (n > 0) apply_move<SIDE, NW>(g, s, slot < 3 ? k0 : k1, slot < 3 ? slot : slot - 3);
}

// The 'sim_winrate' heuristic as a search leaf: MinimaxEnv.simulate (envs/minimax_ewn.py:215-238) -- num_simulations = 100
// uniformly random playouts from the leaf position, value = playouts TOP_LEFT won / 100.  As upstream, no terminal +-10 here
// (evaluate() returns simulate() before that test, :37-38: a finished position scores 1.0 or 0.0), and the side to move in a
// playout is `current_player` of the policy's private env, which simulate() flips before EVERY move and never restores: the
// first mover of a playout is whoever did not make the last move of the previous one, across leaves too.  The search never sets
// current_player, so that chain is all there is.  DELIBERATE DEVIATION: here the chain starts at TOP_LEFT (the env right after
// construction) at EVERY predict(); upstream the agent's private env persists, so from its second predict() on the first mover of
// the first playout is inherited from the previous call.  A stateless batched policy has no "previous call" (observations of many
// games arrive in one batch, in any order); the effect is on which side moves first in a random playout, statistical only, and the
// reference pins nothing here (unseeded `random`).  One generator per search, drawn from in the order of the depth-first search.  Statistical parity with the reference (unseeded Python
// `random`); bit-exact with oracle/ewn_oracle.c, which mirrors it.
template <int NW>
struct SimLeaf {
    static constexpr bool outline_deep = true;   // ewn_core.hpp search_outlined
    PlayoutRng ps;
    int cur;      // 0 TOP_LEFT, 1 BOTTOM_RIGHT: MinimaxEnv.current_player
    int nsims;
    EWN_DEV double operator()(const Geom &g, const GState<NW> &s0)
    {
        int wins = 0;
        for (int sim = 0; sim < nsims; sim++) {
            GState<NW> s = s0;
            for (int ply = 0; ply < 4096 && !is_win<NW>(g, s); ply++) {
                cur ^= 1;                                               // self.switch_player()
                if (cur == 0) rollout_ply<0, NW>(g, s, ps); else rollout_ply<1, NW>(g, s, ps);
            }
            wins += top_left_won<NW>(g, s) ? 1 : 0;    // :233-235
        }
        return (double)wins / (double)nsims;
    }
};

// ExpectiMinimaxAgent(heuristic='sim_winrate').predict: one thread per observation (a leaf costs 100 playouts; nobody runs this
// at scale, the reference included).  active (may be NULL): lanes of a split-phase step that need no reply are skipped.
template <int NW, int DEPTH>
__global__ __launch_bounds__(64) void k_predict_minimax_sim(Geom g, int M, const int8_t *boards, const int8_t *dice, const uint8_t *active,
                                                            const u32 *obs_id, u64 key, int nsims, int8_t *actions, double *values)
{
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M || (active && !active[m])) return;
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    int f = -1, d = -1;
    double v = 0.0;
    const int dc = dice[m];
    if (s.aliveP != 0 && dc >= 1 && dc <= g.CN) {
        SimLeaf<NW> leaf;
        leaf.ps.seed(PlayoutRng::obs_word(obs_id ? obs_id[m] : (u32)m, 0x53494D57u, key), 0u);
        leaf.cur = 0; leaf.nsims = nsims;
        v = search<NW, DEPTH, 0, true>(g, s, dc, -__builtin_inf(), __builtin_inf(), leaf, f, d);
    }
    actions[2 * m] = (int8_t)f; actions[2 * m + 1] = (int8_t)d;
    if (values) values[m] = v;
}

// classical_policies/mcts.py:21-45, any cube_layer: thread (observation m, root move i, playout r)
template <int NW>
__global__ __launch_bounds__(BS) void k_mcts_rollout(Geom g, int M, int total, const int8_t *boards, const int8_t *dice,
                                                     const u32 *obs_id, u64 key, int32_t *wins)
{
    const long long idx = (long long)blockIdx.x * BS + threadIdx.x;
    const long long per = 6ll * total;
    if (idx >= (long long)M * per) return;
    const int m = (int)(idx / per), rem = (int)(idx % per), i = rem / total, r = rem % total;
    if (wins[(size_t)m * 6 + i] < 0) return; // no such root move (or inactive lane); set by k_mcts_init, never by this kernel
    GState<NW> s;
    decode_board<NW>(g, boards + (size_t)m * g.cells, s);
    {
        int j = 0, mk = 0, md = 0;
        for_each_legal<0, NW>(g, s, dice[m], [&](int, int k, int dir) { if (j == i) { mk = k; md = dir; } j++; return j <= i; });
        apply_move<0, NW>(g, s, mk, md);
    }
    PlayoutRng ps;
    ps.seed(PlayoutRng::obs_word(obs_id ? obs_id[m] : (u32)m, 0x4D435453u, key), (u32)(i * total + r));
    int cur = 1; // BOTTOM_RIGHT replies first, mcts.py:26
    for (int ply = 0; ply < 1024 && !is_win<NW>(g, s); ply++) {
        if (cur == 0) rollout_ply<0, NW>(g, s, ps); else rollout_ply<1, NW>(g, s, ps);
        cur ^= 1;
    }
    if (top_left_won<NW>(g, s)) atomicAdd(&wins[(size_t)m * 6 + i], 1); // mcts.py:39-41
}

// the same for cube_layer <= 3 (ewn_playout.hpp).  A block owns opb <= 10 observations = up to 60 root-move slots, lists
// the slots that hold a move (k_mcts_init left wins >= 0 there), and its 256 >> gl aligned groups of 2^gl lanes walk that
// list; the lanes of a group share the playouts of the group's root move (run_playouts).
// Measured (MI355X, ms per step): 7x7/400 playouts (4 groups): opb 10 -> 1.76, 5 -> 1.63, 3 -> 1.58, 2 -> 1.60;
// 5x5/50 playouts (32 groups): opb 10 -> 0.417, 5 -> 0.446, 3 -> 0.535: about 0.8 live slots per group and pass.
static inline int mcts_obs_per_block(int gl)
{
    const int groups = BS >> gl, opb = (3 * groups + 2) / 4;
    return opb < 2 ? 2 : (opb > 10 ? 10 : opb);
}

__global__ __launch_bounds__(BS) void k_mcts_rollout_lean(Geom g, int M, int total, int gl, int opb, const int8_t *boards, const int8_t *dice,
                                                          const u32 *obs_id, u64 key, int32_t *wins)
{
    __shared__ PlayTab T;
    __shared__ uint8_t live[64];
    __shared__ int nlive_s;
    __shared__ int next[BS / 8];
    __shared__ u32 gw[BS / 8][4];
    playtab_build(&T, g.S);
    const long long cell0 = (long long)blockIdx.x * (opb * 6);
    if (threadIdx.x < 64) {
        const long long cell = cell0 + threadIdx.x;
        const bool lv = (int)threadIdx.x < opb * 6 && cell < (long long)M * 6 && wins[cell] >= 0;
        const unsigned long long mask = __ballot(lv);
        if (lv) live[__popcll(mask & ((1ull << threadIdx.x) - 1ull))] = (uint8_t)threadIdx.x;
        if (threadIdx.x == 0) nlive_s = __popcll(mask);
    }
    __syncthreads();
    const int nlive = nlive_s, tc = 1 << gl, lane = threadIdx.x & (tc - 1);
    int *nx = &next[threadIdx.x >> gl];
    for (int slot = threadIdx.x >> gl; slot < nlive; slot += BS >> gl) {
        const long long cell = cell0 + live[slot];
        const int m = (int)(cell / 6), i = (int)(cell % 6);
        if (lane == 0) *nx = tc; // same wave as the lanes that read it: LDS operations of a wave execute in order
        int w;
        PState b0 = pstate_load(g, boards + (size_t)m * g.cells, lane, tc, gw[threadIdx.x >> gl]);
        if (playout_root_move(&T, b0, g.S, dice[m], i)) w = lane < total ? (total - lane + tc - 1) >> gl : 0; // TOP_LEFT has won
        else w = run_playouts<1>(&T, b0, g.S, PlayoutRng::obs_word(obs_id ? obs_id[m] : (u32)m, 0x4D435453u, key),
                                 (u32)(i * total), lane, total, nx); // BOTTOM_RIGHT replies first, mcts.py:26
        for (int off = tc >> 1; off > 0; off >>= 1) w += __shfl_down(w, off, tc); // groups never straddle a wave
        if (lane == 0 && w) atomicAdd(&wins[cell], w);
    }
}

// ---------------------------------------------------------------- K env steps per launch with the flat Monte-Carlo opponent
//
// ewn_step with the MCTS opponent is three launches per step (agent half, playouts, opponent half) with the canonical
// observation, the win counters and the chosen action travelling through scratch memory in between.  Here the three phases are
// the phases of ONE loop body, K steps per launch: a thread owns a game for the rules (state in registers for the whole launch),
// and between the two halves of a step the block's 256 lanes regroup into groups of 8..64 lanes that play the playouts of the
// block's (game, root move) cells out of LDS -- the same generator, the same playout numbering, so the results are those of
// k_mcts_rollout_lean bit for bit.  Block barriers separate the phases (the games of a block are in lock step).
EWN_DEV PState pstate_from_gstate(const Geom &g, const GState<1> &c)
{
    u32 w[4] = { 0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u };   // every cube off the board (pstate_load's encoding)
    #pragma unroll
    for (int k = 0; k < 6; k++) {
        const u32 cp = (u32)pos_get<1>(c.posP, k), rp = (cp * g.div_magic) >> 16, cn = (u32)pos_get<1>(c.posN, k), rn = (cn * g.div_magic) >> 16;
        if ((c.aliveP >> k) & 1u) w[k & 1] ^= (0x40u ^ (rp * 8u + (cp - rp * (u32)g.S))) << (8 * (k >> 1));
        if ((c.aliveN >> k) & 1u) w[2 + (k & 1)] ^= (0x40u ^ (rn * 8u + (cn - rn * (u32)g.S))) << (8 * (k >> 1));
    }
    const PState st = { w[0], w[1], w[2], w[3] };
    return st;
}

// ---------------------------------------------------------------- ewn_step_k for the geometries without a table image
//
// cube_layer 4 / 5 (ten / fifteen cubes a side) and boards of 9x9 .. 11x11: the generic rules and the compile-time-unrolled recursion
// of k_step<NW, 0, 0>, one thread per game, K env steps per launch with the state in registers -- what k_rollout_d3 is for the
// table-driven geometries (eval_minimax.py:16-50's predict / step loop with RandomAgent or env.action_space.sample() as the agent,
// RandomAgent or minimax opponents of the four evaluate() heuristics).  Same trajectory, same totals, same results as K ewn_step calls.
struct GenRoll { int K, agent_sample, strd; };   // strd: bytes per game of the LDS staging area (a record, or S*S)

template <int NW>
__global__ __launch_bounds__(BS) void k_rollout_generic(Geom g, KCfg c, KState st, GenRoll gr, RollBuf B)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];   // [BS][strd]: packed boards in and out, trajectory rows in between
    const int tid = (int)threadIdx.x, lane0 = (int)blockIdx.x * BS, nl = min(BS, c.N - lane0), lane = lane0 + tid;
    const bool live = lane < c.N;
    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1;
    bool frozen = true;
    if (live) { hdr = *rng_hdr_ptr(st.rng, lane); dice = st.dice[lane]; frozen = st.done[lane] != 0; }
    const bool frozen0 = frozen;
    block_copy_in(lds, st.board + (size_t)lane0 * g.cells, nl * g.cells);
    __syncthreads();
    GState<NW> s;
    decode_board<NW>(g, lds + (live ? tid : 0) * g.cells, s);
    __syncthreads();
    LaneRng r; r.load(c.rng_kind, hdr, rng_win_ptr(st.rng, c.N, c.W, live ? lane : 0, RNGF_CUR(hdr.w)), c.W, c.key);
    r.begin_kernel();
    double ret_acc = 0.0;
    int n_steps = 0, n_eps = 0, n_wins = 0;

    for (int kstep = 0; kstep < gr.K; kstep++) {
        const bool active = live && !frozen;
        StepRes o; o.reward = 0.0; o.term = (live && frozen) ? 1 : 0; o.trunc = 0; o.info = EWN_INFO_NONE;
        int aflag = 0, adir = 0;
        if (active) {
            // the stand-in agent: RandomAgent.predict (the hash pick of ewn_step_out.random_action) or action_space.sample()
            const u32 w = agent_hash(r.seed_mix(), r.draws(), (u32)(c.lane_offset + lane), c.key);
            if (gr.agent_sample) { const int a6 = (int)__umulhi(w, 6u); aflag = a6 >= 3 ? 1 : 0; adir = a6 - 3 * aflag; }
            else {
                const int n = for_each_legal<0, NW>(g, s, dice, [](int, int, int) { return true; });
                if (n > 0) {
                    const int pick = (int)__umulhi(w, (u32)n);
                    int i = 0;
                    for_each_legal<0, NW>(g, s, dice, [&](int flag, int, int dir) { if (i == pick) { aflag = flag; adir = dir; } i++; return i <= pick; });
                }
            }
            r.prefetch();
            r.begin_step();
            if (step_agent<NW>(g, c, s, dice, aflag, adir, r, nullptr, o)) {            // envs/ewn.py:438-458
                int oflag = 0, odir = 0;
                if (c.opp == EWN_OPP_RANDOM) policy_random<NW>(g, s, dice, r, oflag, odir);
                else policy_minimax_rt<NW>(g, s, dice, c.depth, c.heur, oflag, odir);
                step_opponent<NW>(g, c, s, dice, oflag, odir, r, nullptr, o);              // envs/ewn.py:464-486
            }
            ret_acc += o.reward; n_steps++; n_eps += o.term; n_wins += o.info == EWN_INFO_WON ? 1 : 0;
            if (o.term) { if (c.autoreset) lane_auto_reset<NW>(g, c, st.rng, lane, s, dice, r); else frozen = true; }
        }
        // ---- this step's trajectory row
        if (live) {
            const size_t oo = (size_t)kstep * c.N + lane;
            if (B.t_action) ((uint16_t *)B.t_action)[oo] = (uint16_t)((uint8_t)aflag | ((uint16_t)(uint8_t)adir << 8));
            if (B.t_dice) B.t_dice[oo] = (int8_t)dice;
            if (B.t_reward) B.t_reward[oo] = o.reward;
            if (B.t_term) B.t_term[oo] = (uint8_t)o.term;
            if (B.t_trunc) B.t_trunc[oo] = (uint8_t)o.trunc;
            if (B.t_info) B.t_info[oo] = (uint8_t)o.info;
        }
        if (B.t_board) {
            if (live) encode_board<NW>(g, s, lds + tid * g.cells);
            __syncthreads();
            block_copy_out(B.t_board + ((size_t)kstep * c.N + lane0) * g.cells, lds, nl * g.cells);
            __syncthreads();
        }
        if (B.t_rec) { // one aligned record per lane-step: board | dice | action | flags | padding (ewn_rollout_out.record)
            if (live) {
                int8_t *rec = lds + tid * gr.strd;
                for (int i = g.cells; i < gr.strd; i++) rec[i] = 0;
                encode_board<NW>(g, s, rec);
                rec[g.cells] = (int8_t)dice; rec[g.cells + 1] = (int8_t)aflag; rec[g.cells + 2] = (int8_t)adir;
                rec[g.cells + 3] = (int8_t)o.term; rec[g.cells + 4] = (int8_t)o.trunc; rec[g.cells + 5] = (int8_t)o.info;
            }
            __syncthreads();
            block_copy_out((int8_t *)B.t_rec + ((size_t)kstep * c.N + lane0) * gr.strd, lds, nl * gr.strd);
            __syncthreads();
        }
    }
    if (live) {
        encode_board<NW>(g, s, lds + tid * g.cells);
        if (!frozen0) { *rng_hdr_ptr(st.rng, lane) = r.header(); st.dice[lane] = (int8_t)dice; }
        st.done[lane] = frozen ? 1 : 0;
        if (B.ret_sum) B.ret_sum[lane] += ret_acc;
        if (B.n_steps) B.n_steps[lane] += n_steps;
        if (B.n_episodes) B.n_episodes[lane] += n_eps;
        if (B.n_wins) B.n_wins[lane] += n_wins;
    }
    __syncthreads();
    block_copy_out(st.board + (size_t)lane0 * g.cells, lds, nl * g.cells);
}

struct MctsRoll { int K, total, gl, agent_sample, strd, gpb; };   // strd: bytes per game of the dynamic LDS area (a record, or S*S)

// mr.gpb games per block of BS threads: the rules run one thread per game (the block's first lanes), the playouts on all BS lanes.
// Not 256 games per block: the playouts are where the time goes, and 65 536 games at 256 per block are 1 024 waves -- ONE per SIMD,
// half the VALU issue rate and no latency hiding (measured: 767 us per step at MCTS(10 x 5) against 409 for the three-launch step;
// 426 at 32 games per block).  The block barriers between the phases are what is left of the difference: the three-launch step
// balances its playouts over the whole chip, this kernel over one block's cells.
#define MR_GPB 128        // the most games a block takes (LDS arrays); the launcher picks mr.gpb <= MR_GPB (default 64)

__global__ __launch_bounds__(BS) void k_rollout_mcts(Geom g, KCfg c, KState st, MctsRoll mr, RollBuf B)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];   // [MR_GPB][strd]: packed boards in and out, trajectory rows in between
    __shared__ PlayTab T;
    __shared__ PState pb0[MR_GPB];
    __shared__ u32 pword[MR_GPB];
    __shared__ int8_t pdice[MR_GPB];
    __shared__ int wins[MR_GPB][6];
    __shared__ uint16_t livec[MR_GPB * 6];
    __shared__ int nlive_s, next_slot;
    __shared__ int nextc[BS / 8], myslot[BS / 8];
    playtab_build(&T, g.S);
    const int tid = (int)threadIdx.x, lane0 = (int)blockIdx.x * mr.gpb, nl = min(mr.gpb, c.N - lane0), lane = lane0 + tid;
    const bool owner = tid < mr.gpb, live = owner && lane < c.N;
    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    int dice = 1;
    bool frozen = true;
    if (live) { hdr = *rng_hdr_ptr(st.rng, lane); dice = st.dice[lane]; frozen = st.done[lane] != 0; }
    const bool frozen0 = frozen;
    block_copy_in(lds, st.board + (size_t)lane0 * g.cells, nl * g.cells);
    __syncthreads();
    GState<1> s;
    decode_board<1>(g, lds + (live ? tid : 0) * g.cells, s);
    LaneRng r; r.load(c.rng_kind, hdr, rng_win_ptr(st.rng, c.N, c.W, live ? lane : 0, RNGF_CUR(hdr.w)), c.W, c.key);
    r.begin_kernel();
    double ret_acc = 0.0;
    int n_steps = 0, n_eps = 0, n_wins = 0;
    const int tc = 1 << mr.gl, glane = tid & (tc - 1), grp = tid >> mr.gl;

    for (int kstep = 0; kstep < mr.K; kstep++) {
        const bool active = live && !frozen;
        StepRes o; o.reward = 0.0; o.term = (live && frozen) ? 1 : 0; o.trunc = 0; o.info = EWN_INFO_NONE;
        int aflag = 0, adir = 0, n_root = 0;
        bool reply = false;
        GState<1> cst = s;
        if (active) {
            // the stand-in agent: RandomAgent.predict (the hash pick of ewn_step_out.random_action) or action_space.sample()
            const u32 w = agent_hash(r.seed_mix(), r.draws(), (u32)(c.lane_offset + lane), c.key);
            if (mr.agent_sample) { const int a6 = (int)__umulhi(w, 6u); aflag = a6 >= 3 ? 1 : 0; adir = a6 - 3 * aflag; }
            else {
                const int n = for_each_legal<0, 1>(g, s, dice, [](int, int, int) { return true; });
                if (n > 0) {
                    const int pick = (int)__umulhi(w, (u32)n);
                    int i = 0;
                    for_each_legal<0, 1>(g, s, dice, [&](int flag, int, int dir) { if (i == pick) { aflag = flag; adir = dir; } i++; return i <= pick; });
                }
            }
            r.prefetch();
            r.begin_step();
            reply = step_agent<1>(g, c, s, dice, aflag, adir, r, nullptr, o);      // envs/ewn.py:438-458
            if (reply) { // MctsAgent.predict's input: the canonical observation (envs/ewn.py:289-296), its root moves, its playout stream
                cst = canonicalize<1>(g, s);
                pb0[tid] = pstate_from_gstate(g, cst);
                pdice[tid] = (int8_t)dice;
                pword[tid] = PlayoutRng::obs_word(r.seed_mix() * 0x9E3779B1u + r.draws(), 0x4D435453u, c.key);
                n_root = for_each_legal<0, 1>(g, cst, dice, [](int, int, int) { return true; });
            }
        }
        if (owner) {
            #pragma unroll
            for (int i = 0; i < 6; i++) wins[tid][i] = i < n_root ? 0 : -1;
        }
        if (tid == 0) { nlive_s = 0; next_slot = BS >> mr.gl; }
        __syncthreads();
        if (n_root > 0) { const int base = atomicAdd(&nlive_s, n_root); for (int i = 0; i < n_root; i++) livec[base + i] = (uint16_t)(tid * 8 + i); }
        __syncthreads();
        // ---- the playouts of the block's (game, root move) cells, mcts.py:47-69: a group of 2^gl lanes per cell; a group that has
        // finished its cell takes the next unplayed one (results do not depend on who plays what)
        const int nlive = nlive_s;
        int slot = grp;
        while (slot < nlive) {
            const int cell = livec[slot], gi = cell >> 3, i = cell & 7;
            if (glane == 0) nextc[grp] = tc;   // same wave as the lanes that read it: LDS operations of a wave execute in order
            PState b0 = pb0[gi];
            int w;
            if (playout_root_move(&T, b0, g.S, pdice[gi], i)) w = glane < mr.total ? (mr.total - glane + tc - 1) >> mr.gl : 0;   // TOP_LEFT has won
            else w = run_playouts<1>(&T, b0, g.S, pword[gi], (u32)(i * mr.total), glane, mr.total, &nextc[grp]);         // BOTTOM_RIGHT replies first
            for (int off = tc >> 1; off > 0; off >>= 1) w += __shfl_down(w, off, tc);
            if (glane == 0) { wins[gi][i] = w; myslot[grp] = atomicAdd(&next_slot, 1); }
            __builtin_amdgcn_wave_barrier();
            slot = myslot[grp];
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        if (reply) { // np.argmax over the root moves' wins (first maximum), that entry of the legal list (mcts.py:68), then envs/ewn.py:464-486
            int best = 0, bw = -1;
            #pragma unroll
            for (int i = 0; i < 6; i++) { const int w = wins[tid][i]; if (w > bw) { bw = w; best = i; } }
            int oflag = 0, odir = 0, j = 0;
            for_each_legal<0, 1>(g, cst, dice, [&](int flag, int, int dir) { if (j == best) { oflag = flag; odir = dir; } j++; return j <= best; });
            step_opponent<1>(g, c, s, dice, oflag, odir, r, nullptr, o);
        }
        if (active) {
            ret_acc += o.reward; n_steps++; n_eps += o.term; n_wins += o.info == EWN_INFO_WON ? 1 : 0;
            if (o.term) { if (c.autoreset) lane_auto_reset<1>(g, c, st.rng, lane, s, dice, r); else frozen = true; }
        }
        // ---- this step's trajectory row
        if (live) {
            const size_t oo = (size_t)kstep * c.N + lane;
            if (B.t_action) ((uint16_t *)B.t_action)[oo] = (uint16_t)((uint8_t)aflag | ((uint16_t)(uint8_t)adir << 8));
            if (B.t_dice) B.t_dice[oo] = (int8_t)dice;
            if (B.t_reward) B.t_reward[oo] = o.reward;
            if (B.t_term) B.t_term[oo] = (uint8_t)o.term;
            if (B.t_trunc) B.t_trunc[oo] = (uint8_t)o.trunc;
            if (B.t_info) B.t_info[oo] = (uint8_t)o.info;
        }
        if (B.t_board) {
            if (live) encode_board<1>(g, s, lds + tid * g.cells);
            __syncthreads();
            block_copy_out(B.t_board + ((size_t)kstep * c.N + lane0) * g.cells, lds, nl * g.cells);
            __syncthreads();
        }
        if (B.t_rec) { // one aligned record per lane-step: board | dice | action | flags | padding (ewn_rollout_out.record)
            if (live) {
                int8_t *rec = lds + tid * mr.strd;
                for (int i = g.cells; i < mr.strd; i++) rec[i] = 0;
                encode_board<1>(g, s, rec);
                rec[g.cells] = (int8_t)dice; rec[g.cells + 1] = (int8_t)aflag; rec[g.cells + 2] = (int8_t)adir;
                rec[g.cells + 3] = (int8_t)o.term; rec[g.cells + 4] = (int8_t)o.trunc; rec[g.cells + 5] = (int8_t)o.info;
            }
            __syncthreads();
            block_copy_out((int8_t *)B.t_rec + ((size_t)kstep * c.N + lane0) * mr.strd, lds, nl * mr.strd);
            __syncthreads();
        }
    }
    if (live) encode_board<1>(g, s, lds + tid * g.cells);
    if (live) {
        if (!frozen0) { *rng_hdr_ptr(st.rng, lane) = r.header(); st.dice[lane] = (int8_t)dice; }
        st.done[lane] = frozen ? 1 : 0;
        if (B.ret_sum) B.ret_sum[lane] += ret_acc;
        if (B.n_steps) B.n_steps[lane] += n_steps;
        if (B.n_episodes) B.n_episodes[lane] += n_eps;
        if (B.n_wins) B.n_wins[lane] += n_wins;
    }
    __syncthreads();
    block_copy_out(st.board + (size_t)lane0 * g.cells, lds, nl * g.cells);
}

template <int NW>
__global__ __launch_bounds__(BS) void k_mcts_pick(Geom g, int M, const int8_t *boards, const int8_t *dice, const int32_t *wins,
                                                  int8_t *actions)
{
    const int m = blockIdx.x * BS + threadIdx.x;
    if (m >= M) return;
    int best = -1, bw = -1;
    for (int i = 0; i < 6; i++) { const int w = wins[(size_t)m * 6 + i]; if (w > bw) { bw = w; best = i; } } // np.argmax: first max
    int f = 0, dr = 0;
    if (best >= 0) {
        GState<NW> s;
        decode_board<NW>(g, boards + (size_t)m * g.cells, s);
        int j = 0;
        for_each_legal<0, NW>(g, s, dice[m], [&](int flag, int, int dir) { if (j == best) { f = flag; dr = dir; } j++; return j <= best; });
    }
    actions[2 * m] = (int8_t)f; actions[2 * m + 1] = (int8_t)dr;
}

// ---------------------------------------------------------------- host side: C ABI

#define GRID(n) dim3((unsigned)(((long long)(n) + BS - 1) / BS))
// NWV = words of packed cube positions per side: 1 (<= 10 cubes) or 2 on boards up to 8x8 (64-bit occupancy masks, 6-bit positions),
// 3 = the mask-free state of boards from 9x9 to 11x11 (7-bit positions, ewn_core.hpp GState<3>)
#define BY_NW(g, X) do { if ((g).S > 8) { constexpr int NWV = 3; X; } else if ((g).CN <= 10) { constexpr int NWV = 1; X; } else { constexpr int NWV = 2; X; } } while (0)

template <int NW>
static void launch_minimax(const Geom &g, int M, const int8_t *boards, const int8_t *dice, int depth, int heur, int8_t *actions,
                           double *values, hipStream_t s)
{
    switch (depth) {
    case 1: k_predict_minimax<NW, 1><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, heur, actions, values); break;
    case 2: k_predict_minimax<NW, 2><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, heur, actions, values); break;
    case 3: k_predict_minimax<NW, 3><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, heur, actions, values); break;
    case 4: k_predict_minimax<NW, 4><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, heur, actions, values); break;
    case 5: k_predict_minimax<NW, 5><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, heur, actions, values); break;
    default: k_predict_minimax<NW, 6><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, heur, actions, values); break;
    }
}

template <int NW>
static int launch_minimax_sim(const Geom &g, int M, const int8_t *boards, const int8_t *dice, const uint8_t *active, const u32 *obs_id,
                              u64 key, int depth, int8_t *actions, double *values, hipStream_t s)
{
    const dim3 grid((unsigned)((M + 63) / 64));
    switch (depth) {
    case 1: k_predict_minimax_sim<NW, 1><<<grid, 64, 0, s>>>(g, M, boards, dice, active, obs_id, key, EWN_SIM_WINRATE_PLAYOUTS, actions, values); break;
    case 2: k_predict_minimax_sim<NW, 2><<<grid, 64, 0, s>>>(g, M, boards, dice, active, obs_id, key, EWN_SIM_WINRATE_PLAYOUTS, actions, values); break;
    case 3: k_predict_minimax_sim<NW, 3><<<grid, 64, 0, s>>>(g, M, boards, dice, active, obs_id, key, EWN_SIM_WINRATE_PLAYOUTS, actions, values); break;
    case 4: k_predict_minimax_sim<NW, 4><<<grid, 64, 0, s>>>(g, M, boards, dice, active, obs_id, key, EWN_SIM_WINRATE_PLAYOUTS, actions, values); break;
    // max_depth 5 / 6: thousands of leaves x 100 playouts per observation, one thread each -- seconds per launch (the reference: minutes
    // per move); served because the reference has no such limit (classical_policies/minimax.py:19-73 with envs/minimax_ewn.py:36-37)
    case 5: k_predict_minimax_sim<NW, 5><<<grid, 64, 0, s>>>(g, M, boards, dice, active, obs_id, key, EWN_SIM_WINRATE_PLAYOUTS, actions, values); break;
    case 6: k_predict_minimax_sim<NW, 6><<<grid, 64, 0, s>>>(g, M, boards, dice, active, obs_id, key, EWN_SIM_WINRATE_PLAYOUTS, actions, values); break;
    default: return EWN_EUNSUPPORTED;
    }
    return launch_status();
}

extern "C" {

int ewn_abi_version(void) { return EWN_ABI_VERSION; }

const char *ewn_strerror(int code)
{
    switch (code) {
    case EWN_OK: return "ok";
    case EWN_EINVAL: return "invalid argument or configuration";
    case EWN_ENULL: return "required pointer is NULL";
    case EWN_ELAUNCH: return "kernel launch failed";
    case EWN_EUNSUPPORTED: return "configuration valid upstream but not supported by this build";
    default: return "unknown error";
    }
}

int64_t ewn_tables_bytes(int board_size, int cube_layer)
{
    Geom g;
    if (!make_geom(board_size, cube_layer, g)) return 0;
    // per heuristic image ('hybrid', 'min_dist', 'attk') two variants: max_depth 1-3 and 5, then max_depth 4 and 6 (ewn_fast.hpp)
    return 2 * FAST_HEUR_IMAGES * fast_tables_bytes(board_size, cube_layer);
}

int ewn_build_tables(int board_size, int cube_layer, void *host_out)
{
    if (!host_out) return EWN_ENULL;
    if (ewn_tables_bytes(board_size, cube_layer) <= 0) return EWN_EUNSUPPORTED;
    int rc = 0;
    static const int heur_of_image[FAST_HEUR_IMAGES] = { EWN_H_HYBRID, EWN_H_MIN_DIST, EWN_H_ATTK, EWN_H_TWO_MIN_DIST };
    for (int hi = 0; hi < FAST_HEUR_IMAGES; hi++)
        for (int variant = 0; variant < 2; variant++) {
            void *dst = (int8_t *)host_out + (size_t)(hi * 2 + variant) * fast_tables_bytes(board_size, cube_layer);
            switch (board_size) {
            case 5: rc |= build_fast_tables<5>((FastTab<5> *)dst, variant, heur_of_image[hi]); break;
            case 6: rc |= build_fast_tables<6>((FastTab<6> *)dst, variant, heur_of_image[hi]); break;
            case 7: rc |= build_fast_tables<7>((FastTab<7> *)dst, variant, heur_of_image[hi]); break;
            case 8: rc |= build_fast_tables<8>((FastTab<8> *)dst, variant, heur_of_image[hi]); break;
            default: rc = -1;
            }
        }
    return rc == 0 ? EWN_OK : EWN_EUNSUPPORTED;
}

int ewn_rng_words(const ewn_config *cfg)
{
    Geom g; KCfg k;
    const int rc = check_cfg(cfg, g, k);
    return rc ? rc : (int)k.rng_words;
}

int64_t ewn_step_scratch_bytes(const ewn_config *cfg)
{
    Geom g; KCfg k;
    const int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    const int64_t N = k.N;
    const bool split = cfg->opponent_kind == EWN_OPP_MCTS || (cfg->opponent_kind == EWN_OPP_MINIMAX && cfg->heuristic == EWN_H_SIM_WINRATE);
    if (!split)   // MT kind with auto-reset: the refill queue of the lean step kernel (ewn_step_d3.hpp)
        return (cfg->rng_kind == EWN_RNG_MT19937 && cfg->autoreset) ? mtq_bytes(k.N) : 0;
    // phase u8 | cdice i8 | act i8x2 | (pad to 8) | obs_id u32 | wins i32x6 | cboard i8[cells]
    return ((N * 4 + 7) / 8) * 8 + N * 4 + N * 24 + N * g.cells;
}

static void carve_scratch(const Geom &g, const KCfg &k, void *scratch, KScratch &sc)
{
    int8_t *p = (int8_t *)scratch;
    const int64_t N = k.N;
    sc.phase = (uint8_t *)p;
    sc.cdice = p + N;
    sc.act = p + 2 * N;
    p += ((N * 4 + 7) / 8) * 8;
    sc.obs_id = (u32 *)p; p += N * 4;
    sc.wins = (int32_t *)p; p += N * 24;
    sc.cboard = p;
}

static KState kstate(const ewn_state *st)
{
    KState s = { st->board, st->dice, st->done, st->rng, st->prev_score, st->tolerance, st->tables };
    return s;
}

int ewn_init_aux(const ewn_config *cfg, const ewn_state *st, void *stream)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (!st || !st->done || !st->rng) return EWN_ENULL;
    if (cfg->shaped && (!st->prev_score || !st->tolerance)) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    BY_NW(g, (k_init_aux<NWV><<<GRID(k.N), BS, 0, s>>>(g, k, kstate(st), cfg->illegal_move_tolerance)));
    return launch_status();
}

int ewn_reset(const ewn_config *cfg, const ewn_state *st, const uint32_t *seeds, const uint8_t *lane_mask, void *stream)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (!st || !st->board || !st->dice || !st->done || !st->rng) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)BS * g.cells;
    BY_NW(g, (k_reset<NWV><<<GRID(k.N), BS, lds, s>>>(g, k, kstate(st), seeds, lane_mask)));
    return launch_status();
}

int ewn_roll_dice(const ewn_config *cfg, const ewn_state *st, const uint8_t *lane_mask, void *stream)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (!st || !st->dice || !st->done || !st->rng) return EWN_ENULL;
    k_roll_dice<<<GRID(k.N), BS, 0, (hipStream_t)stream>>>(g, k, kstate(st), lane_mask);
    return launch_status();
}

int ewn_predict_mcts(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int num_simulations,
                     int num_env_copies, uint64_t key, const uint32_t *obs_id, int8_t *actions, int32_t *wins, void *stream);

// init_and_pick: the stateless policy (ewn_predict_mcts) needs k_mcts_init / k_mcts_pick; inside ewn_step the two half-step
// kernels do both jobs (k_step PHASE 1 / 2), so the MCTS opponent's step is three launches: half step, playouts, half step
static int mcts_launch(const Geom &g, int M, const int8_t *boards, const int8_t *dice, const uint8_t *active, int total, u64 key,
                       const u32 *obs_id, int8_t *actions, int32_t *wins, hipStream_t s, bool init_and_pick = true)
{
    const bool lean = g.CN <= 6 && g.S <= 8; // the byte-per-cube playout numbers cells row * 8 + col
    const int gl = playout_group_log2(total), opb = mcts_obs_per_block(gl);
    const long long threads = lean ? (((long long)M + opb - 1) / opb) * BS : (long long)M * 6 * total;
    if (threads > 0x7fffffffll * BS) return EWN_EINVAL;
    if (init_and_pick) BY_NW(g, (k_mcts_init<NWV><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, active, wins)));
    if (lean) k_mcts_rollout_lean<<<GRID(threads), BS, 0, s>>>(g, M, total, gl, opb, boards, dice, obs_id, key, wins);
    else BY_NW(g, (k_mcts_rollout<NWV><<<GRID(threads), BS, 0, s>>>(g, M, total, boards, dice, obs_id, key, wins)));
    if (init_and_pick) BY_NW(g, (k_mcts_pick<NWV><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, wins, actions)));
    return launch_status();
}


int ewn_step(const ewn_config *cfg, const ewn_state *st, const int8_t *actions, const ewn_step_out *out, void *scratch, void *stream)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (!st || !st->board || !st->dice || !st->done || !st->rng || !actions || !out) return EWN_ENULL;
    if (!out->reward || !out->terminated || !out->truncated || !out->info) return EWN_ENULL;
    if (cfg->shaped && (!st->prev_score || !st->tolerance)) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    KOut ko = { out->reward, out->terminated, out->truncated, out->info, out->terminal_board, out->terminal_dice, out->random_action };
    KState ks = kstate(st);
    if (!cfg->shaped) { ks.prev_score = nullptr; ks.tolerance = nullptr; }
    KScratch sc = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    // MT kind with auto-reset: the step kernel flags the lanes whose spare window it consumed; k_mt_refill rebuilds them right after
    const bool refill = cfg->rng_kind == EWN_RNG_MT19937 && cfg->autoreset;
    const bool sim_opp = cfg->opponent_kind == EWN_OPP_MINIMAX && cfg->heuristic == EWN_H_SIM_WINRATE;
    const bool split = cfg->opponent_kind == EWN_OPP_MCTS || sim_opp; // the opponent's policy runs as kernels of its own between two half steps
    if (split) {
        if (!scratch) return EWN_ENULL;
        carve_scratch(g, k, scratch, sc);
    }
    const size_t lds = (size_t)2 * BS * g.cells;
    if (!split) {
        const bool fast = st->tables && fast_tables_bytes(g.S, g.L) > 0 && cfg->opponent_kind == EWN_OPP_MINIMAX &&
                          cfg->max_depth <= 6 && fast_heur_image(cfg->heuristic) >= 0;
        const bool lean_random = st->tables && fast_tables_bytes(g.S, g.L) > 0 && cfg->opponent_kind == EWN_OPP_RANDOM;
        if (fast) ks.tables = fast_image(st->tables, g.S, g.L, cfg->max_depth, cfg->heuristic); // the image of this heuristic and depth class
        if (((fast && fast_heur_step(cfg->heuristic)) || lean_random) && d3_threads_per_game(k.N) > 0) {
            // the lean fused kernel: canonical ring space end to end, T lanes per game (ewn_step_d3.hpp / ewn_step_d3.hip).
            // MT kind with auto-reset: window refills are extra blocks of the same launch (needs the caller's scratch).
            const bool fused_refill = refill && scratch != nullptr;
            if (fast && cfg->heuristic == EWN_H_TWO_MIN_DIST) rc = ewn_launch_step_d3_h2(cfg, g, k, st, ks.tables, actions, out, scratch, false, fused_refill, s);
            else rc = ewn_launch_step_d3(cfg, g, k, st, ks.tables, actions, out, scratch, lean_random, fused_refill, s);
            if (rc || fused_refill) return rc;
        } else if (fast) {
            const size_t base = (lds + 15) & ~(size_t)15;
            switch (g.S) {
            case 5: k_step<1, 0, 5><<<GRID(k.N), BS, base + FAST_TAB_BYTES(5), s>>>(g, k, ks, actions, ko, sc); break;
            case 6: k_step<1, 0, 6><<<GRID(k.N), BS, base + FAST_TAB_BYTES(6), s>>>(g, k, ks, actions, ko, sc); break;
            case 7: k_step<1, 0, 7><<<GRID(k.N), BS, base + FAST_TAB_BYTES(7), s>>>(g, k, ks, actions, ko, sc); break;
            default: k_step<1, 0, 8><<<GRID(k.N), BS, base + FAST_TAB_BYTES(8), s>>>(g, k, ks, actions, ko, sc); break;
            }
        } else {
            BY_NW(g, (k_step<NWV, 0, 0><<<GRID(k.N), BS, lds, s>>>(g, k, ks, actions, ko, sc)));
        }
    } else {
        BY_NW(g, (k_step<NWV, 1, 0><<<GRID(k.N), BS, lds, s>>>(g, k, ks, actions, ko, sc)));
        rc = launch_status();
        if (rc) return rc;
        if (sim_opp) {
            BY_NW(g, rc = launch_minimax_sim<NWV>(g, k.N, sc.cboard, sc.cdice, sc.phase, sc.obs_id, k.key, k.depth, sc.act, nullptr, s));
        } else rc = mcts_launch(g, k.N, sc.cboard, sc.cdice, sc.phase, k.nsim_total, k.key, sc.obs_id, sc.act, sc.wins, s, false);
        if (rc) return rc;
        BY_NW(g, (k_step<NWV, 2, 0><<<GRID(k.N), BS, lds, s>>>(g, k, ks, actions, ko, sc)));
    }
    rc = launch_status();
    if (rc) return rc;
    if (refill) {
        k_mt_refill<<<dim3((unsigned)((k.N + REFILL_BS - 1) / REFILL_BS)), REFILL_BS, (size_t)(k.W + 1) * 65 * 4, s>>>(st->rng, k.N, k.W, k.seed_stride);
        rc = launch_status();
    }
    return rc;
}


// which k_rollout_d3 instantiation serves (cfg, agent): EWN_OK and the template selectors, or why not
static int rollout_plan(const ewn_config *cfg, const Geom &g, const KCfg &k, int agent_kind, int agent_max_depth, int &T, int &opp, int &agent)
{
    if (fast_tables_bytes(g.S, g.L) <= 0 || cfg->shaped) return EWN_EUNSUPPORTED;
    if (cfg->opponent_kind == EWN_OPP_RANDOM) opp = 1;
    else if (cfg->opponent_kind == EWN_OPP_MINIMAX && fast_heur_step(cfg->heuristic)) opp = cfg->max_depth > 4 ? 2 : 0;
    else return EWN_EUNSUPPORTED;
    const bool h2 = cfg->opponent_kind == EWN_OPP_MINIMAX && cfg->heuristic == EWN_H_TWO_MIN_DIST;   // its own kernel instances, RandomAgent / sample agents
    if (h2 && agent_kind == EWN_AGENT_MINIMAX) return EWN_EUNSUPPORTED;
    if (agent_kind == EWN_AGENT_RANDOM || agent_kind == EWN_AGENT_SAMPLE) agent = 0;
    else if (agent_kind == EWN_AGENT_MINIMAX) {
        if (agent_max_depth < 1) return EWN_EINVAL;
        if (agent_max_depth > EWN_MAX_DEPTH) return EWN_EUNSUPPORTED;
        agent = agent_max_depth > 4 ? 2 : 1;
    } else return EWN_EINVAL;
    // the MT19937-compat windows of an auto-resetting lane are rebuilt BETWEEN launches (ewn_core.hpp): a launch that plays
    // several episodes of a lane cannot use them.  Without auto-reset (one episode per lane: evaluation) the kind is fine.
    if (cfg->rng_kind == EWN_RNG_MT19937 && cfg->autoreset) return EWN_EUNSUPPORTED;
    const bool mt = cfg->rng_kind == EWN_RNG_MT19937;
    if (agent != 0) T = 2;
    else if (opp == 1) T = 1;
    else if (opp == 2) T = h2 ? 1 : (mt ? 2 : (k.N <= 131072 ? 2 : 1));   // 'two_min_dist' at max_depth 5 / 6: the reference's loops, one lane per game
    else T = mt ? 1 : rollout_threads_per_game(k.N);
    return EWN_OK;
}

// geometries without a table image (cube_layer 4 / 5, boards of 9x9 .. 11x11) and RandomAgent / the four evaluate() heuristics as the
// opponent: the generic K-step kernel (k_rollout_generic); un-shaped, RandomAgent / sample agents, MT19937-compat dice only without auto-reset
static int generic_rollout_plan(const ewn_config *cfg, const Geom &g, int agent_kind)
{
    if (fast_tables_bytes(g.S, g.L) > 0 || cfg->shaped) return EWN_EUNSUPPORTED;
    if (cfg->opponent_kind == EWN_OPP_MINIMAX) { if (cfg->heuristic == EWN_H_SIM_WINRATE) return EWN_EUNSUPPORTED; }
    else if (cfg->opponent_kind != EWN_OPP_RANDOM) return EWN_EUNSUPPORTED;
    if (agent_kind != EWN_AGENT_RANDOM && agent_kind != EWN_AGENT_SAMPLE) return agent_kind == EWN_AGENT_MINIMAX ? EWN_EUNSUPPORTED : EWN_EINVAL;
    if (cfg->rng_kind == EWN_RNG_MT19937 && cfg->autoreset) return EWN_EUNSUPPORTED;
    return EWN_OK;
}

// the flat Monte-Carlo opponent inside ewn_step_k (k_rollout_mcts): byte-per-cube playouts (cube_layer <= 3, boards <= 8x8), un-shaped,
// RandomAgent / sample agents; MT19937-compat dice only without auto-reset (as every K-step kernel)
static int mcts_rollout_plan(const ewn_config *cfg, const Geom &g, int agent_kind)
{
    if (cfg->opponent_kind != EWN_OPP_MCTS || cfg->shaped || g.CN > 6 || g.S > 8) return EWN_EUNSUPPORTED;
    if (agent_kind != EWN_AGENT_RANDOM && agent_kind != EWN_AGENT_SAMPLE) return agent_kind == EWN_AGENT_MINIMAX ? EWN_EUNSUPPORTED : EWN_EINVAL;
    if (cfg->rng_kind == EWN_RNG_MT19937 && cfg->autoreset) return EWN_EUNSUPPORTED;
    return EWN_OK;
}

int ewn_lanes_per_game(const ewn_config *cfg, int entry)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (entry == 1) {
        int T, opp, agent;
        rc = rollout_plan(cfg, g, k, EWN_AGENT_RANDOM, 0, T, opp, agent);
        return rc == EWN_OK ? T : 0;
    }
    if (entry != 0) return EWN_EINVAL;
    const bool tab = fast_tables_bytes(g.S, g.L) > 0;
    const bool fast = tab && cfg->opponent_kind == EWN_OPP_MINIMAX && cfg->max_depth <= 6 && fast_heur_step(cfg->heuristic);
    const bool lean_random = tab && cfg->opponent_kind == EWN_OPP_RANDOM;
    if (!(fast || lean_random) || d3_threads_per_game(k.N) <= 0) return 0;
    if (lean_random || cfg->max_depth < 3) return 1;
    if (cfg->max_depth > 4) return (k.N <= 131072 && d3_threads_per_game(k.N) != 1) ? 2 : 1;
    return d3_threads_per_game(k.N);
}

int ewn_step_k_supported(const ewn_config *cfg, int agent_kind, int agent_max_depth)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (agent_kind == EWN_AGENT_MLP) { rc = ewn_policy_supported(cfg, g); return rc == EWN_OK ? 1 : (rc == EWN_EUNSUPPORTED ? 0 : rc); } // ewn_step_k_policy
    if (cfg->opponent_kind == EWN_OPP_MCTS) { rc = mcts_rollout_plan(cfg, g, agent_kind); return rc == EWN_OK ? 1 : (rc == EWN_EUNSUPPORTED ? 0 : rc); }
    if (generic_rollout_plan(cfg, g, agent_kind) == EWN_OK) return 1;
    int T, opp, agent;
    rc = rollout_plan(cfg, g, k, agent_kind, agent_max_depth, T, opp, agent);
    return rc == EWN_OK ? 1 : (rc == EWN_EUNSUPPORTED ? 0 : rc);
}

int ewn_step_k(const ewn_config *cfg, const ewn_state *st, int K, int agent_kind, int agent_max_depth, const ewn_rollout_out *out,
               void *stream)
{
    Geom g; KCfg k;
    int rc = check_cfg(cfg, g, k);
    if (rc) return rc;
    if (K < 1) return EWN_EINVAL;
    if (cfg->opponent_kind == EWN_OPP_MCTS) {
        if (!st || !st->board || !st->dice || !st->done || !st->rng) return EWN_ENULL;
        rc = mcts_rollout_plan(cfg, g, agent_kind);
        if (rc) return rc;
        RollBuf rb;
        memset(&rb, 0, sizeof(rb));
        if (out) {
            rb.t_board = out->board; rb.t_dice = out->dice; rb.t_action = out->action; rb.t_reward = out->reward;
            rb.t_term = out->terminated; rb.t_trunc = out->truncated; rb.t_info = out->info; rb.t_rec = out->record;
            rb.ret_sum = out->return_sum; rb.n_steps = out->n_steps; rb.n_episodes = out->n_episodes; rb.n_wins = out->n_wins;
        }
        // games per block: about 2 048 blocks (measured, us per step: 65 536 lanes MCTS(10 x 5) 473 / 426 / 449 / 485 at 16 / 32 / 64 / 128 games
        // per block; 32 768 lanes 7x7 400 playouts 1 822 / 2 020 / 2 303 / 3 587) -- fewer, larger blocks starve the SIMDs of waves, smaller
        // ones leave too few cells per barrier to balance.  EWN_MCTS_GPB overrides (tuning).
        static const int gpb_env = [] { const char *e = getenv("EWN_MCTS_GPB"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= MR_GPB ? v : 0; }();
        int gpb = 8;
        while (gpb < MR_GPB && (long long)k.N / (2 * gpb) >= 2048) gpb *= 2;
        if (gpb_env) gpb = gpb_env;
        MctsRoll mr = { K, k.nsim_total, playout_group_log2(k.nsim_total), agent_kind == EWN_AGENT_SAMPLE ? 1 : 0, (g.cells + 6 + 15) & ~15, gpb };
        k_rollout_mcts<<<dim3((unsigned)((k.N + mr.gpb - 1) / mr.gpb)), BS, (size_t)mr.gpb * mr.strd, (hipStream_t)stream>>>(g, k, kstate(st), mr, rb);
        return launch_status();
    }
    if (generic_rollout_plan(cfg, g, agent_kind) == EWN_OK) {
        if (!st || !st->board || !st->dice || !st->done || !st->rng) return EWN_ENULL;
        RollBuf rb;
        memset(&rb, 0, sizeof(rb));
        if (out) {
            rb.t_board = out->board; rb.t_dice = out->dice; rb.t_action = out->action; rb.t_reward = out->reward;
            rb.t_term = out->terminated; rb.t_trunc = out->truncated; rb.t_info = out->info; rb.t_rec = out->record;
            rb.ret_sum = out->return_sum; rb.n_steps = out->n_steps; rb.n_episodes = out->n_episodes; rb.n_wins = out->n_wins;
        }
        GenRoll gr = { K, agent_kind == EWN_AGENT_SAMPLE ? 1 : 0, (g.cells + 6 + 15) & ~15 };
        const KState ks = kstate(st);
        const size_t lds = (size_t)BS * gr.strd;
        hipStream_t s = (hipStream_t)stream;
        BY_NW(g, (k_rollout_generic<NWV><<<GRID(k.N), BS, lds, s>>>(g, k, ks, gr, rb)));
        return launch_status();
    }
    if (!st || !st->board || !st->dice || !st->done || !st->rng || !st->tables) return EWN_ENULL;
    int T, opp, agent;
    rc = rollout_plan(cfg, g, k, agent_kind, agent_max_depth, T, opp, agent);
    if (rc) return rc;
    RollCfg rcf = { k.N, k.autoreset, k.lane_offset, k.depth, agent_max_depth, K, agent_kind == EWN_AGENT_SAMPLE ? 1 : 0, k.seed_stride, k.W, k.reward, k.key };
    RollBuf rb;
    memset(&rb, 0, sizeof(rb));
    rb.board = st->board; rb.dice = st->dice; rb.done = st->done; rb.rng = st->rng;
    rb.tables = opp == 1 ? st->tables : fast_image(st->tables, g.S, g.L, cfg->max_depth, cfg->heuristic);
    rb.agent_tables = agent == 0 ? rb.tables : fast_image(st->tables, g.S, g.L, agent_max_depth, EWN_H_HYBRID);
    if (out) {
        rb.t_board = out->board; rb.t_dice = out->dice; rb.t_action = out->action; rb.t_reward = out->reward;
        rb.t_term = out->terminated; rb.t_trunc = out->truncated; rb.t_info = out->info;
        rb.ret_sum = out->return_sum; rb.n_steps = out->n_steps; rb.n_episodes = out->n_episodes; rb.n_wins = out->n_wins;
        rb.t_rec = out->record;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool h2 = opp != 1 && cfg->heuristic == EWN_H_TWO_MIN_DIST;
    switch (g.S) {
    case 5: return ewn_launch_rollout_s5(rcf, rb, T, opp, k.rng_kind, agent, h2, s);
    case 6: return ewn_launch_rollout_s6(rcf, rb, T, opp, k.rng_kind, agent, h2, s);
    case 7: return ewn_launch_rollout_s7(rcf, rb, T, opp, k.rng_kind, agent, h2, s);
    default: return ewn_launch_rollout_s8(rcf, rb, T, opp, k.rng_kind, agent, h2, s);
    }
}

static int query_geom(int S, int L, int M, const void *boards, Geom &g)
{
    if (S > EWN_MAX_BOARD && L >= 1 && L < S - 1) return EWN_EUNSUPPORTED;
    if (!make_geom(S, L, g) || M < 0) return EWN_EINVAL;
    if (M > 0 && !boards) return EWN_ENULL;
    return EWN_OK;
}

int ewn_legal_actions(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int player, int8_t *acts,
                      int8_t *n_acts, int8_t *cube_small, int8_t *cube_large, uint8_t *win, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (player != 1 && player != 2) return EWN_EINVAL; // Player.get_opponent raises ValueError otherwise, constants/player.py:17
    if (M == 0) return EWN_OK;
    if (!dice) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    BY_NW(g, (k_legal<NWV><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, player, acts, n_acts, cube_small, cube_large, win)));
    return launch_status();
}

int ewn_apply_action(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int player,
                     const int8_t *actions, int8_t *new_boards, uint8_t *valid, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (player != 1 && player != 2) return EWN_EINVAL;
    if (M == 0) return EWN_OK;
    if (!dice || !actions || !new_boards) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    BY_NW(g, (k_apply_action<NWV><<<GRID(M), BS, 0, s>>>(g, M, boards, dice, player, actions, new_boards, valid)));
    return launch_status();
}

int ewn_playout_wins(int board_size, int cube_layer, int M, const int8_t *boards, int first_player, int n_sims, uint64_t key,
                     int32_t *wins, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if ((first_player != 1 && first_player != 2) || n_sims < 1) return EWN_EINVAL;
    if (g.CN < 6) return EWN_EUNSUPPORTED; // dice 1..6 hard-coded upstream
    if (M == 0) return EWN_OK;
    if (!wins) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(wins, 0, (size_t)M * sizeof(int32_t), s) != hipSuccess) return EWN_ELAUNCH;
    if (g.CN <= 6 && g.S <= 8) {
        const int gl = playout_group_log2(n_sims);
        const long long threads = (long long)M << gl;
        if (threads > 0x7fffffffll * BS) return EWN_EINVAL;
        k_playout_wins_lean<<<GRID(threads), BS, 0, s>>>(g, M, n_sims, gl, boards, first_player, key, wins);
        return launch_status();
    }
    const long long threads = (long long)M * n_sims;
    if (threads > 0x7fffffffll * BS) return EWN_EINVAL;
    BY_NW(g, (k_playout_wins<NWV><<<GRID(threads), BS, 0, s>>>(g, M, n_sims, boards, first_player, key, wins)));
    return launch_status();
}

int ewn_evaluate(int board_size, int cube_layer, int M, const int8_t *boards, int heuristic, double *out, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (heuristic < 0 || heuristic > EWN_H_ATTK) return EWN_EUNSUPPORTED;
    if (M == 0) return EWN_OK;
    if (!out) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    BY_NW(g, (k_evaluate<NWV><<<GRID(M), BS, 0, s>>>(g, M, boards, heuristic, out)));
    return launch_status();
}

int ewn_predict_minimax(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int max_depth,
                        int heuristic, int8_t *actions, double *values, const void *tables, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (max_depth < 1) return EWN_EINVAL;
    if (max_depth > EWN_MAX_DEPTH || heuristic < 0 || heuristic > EWN_H_SIM_WINRATE || g.CN < 6) return EWN_EUNSUPPORTED;
    if (heuristic == EWN_H_SIM_WINRATE && max_depth > EWN_SIM_WINRATE_MAX_DEPTH) return EWN_EUNSUPPORTED;
    if (M == 0) return EWN_OK;
    if (!dice || !actions) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    if (heuristic == EWN_H_SIM_WINRATE) {
        // the playout randomness: one generator per observation from (index, sim_key); ewn_predict_minimax has no key argument,
        // ewn_predict_minimax_sim (below) takes one
        BY_NW(g, rc = launch_minimax_sim<NWV>(g, M, boards, dice, nullptr, nullptr, 0, max_depth, actions, values, s));
        return rc;
    }
    if (tables && fast_tables_bytes(g.S, g.L) > 0 && max_depth <= 6 && fast_heur_image(heuristic) >= 0) {
        tables = fast_image(tables, g.S, g.L, max_depth, heuristic);
#define PMF(SS) do { if (heuristic == EWN_H_TWO_MIN_DIST) k_predict_minimax_fast<SS, true><<<GRID(M), BS, FAST_TAB_BYTES(SS), s>>>(g, M, boards, dice, max_depth, actions, values, tables); \
                    else k_predict_minimax_fast<SS, false><<<GRID(M), BS, FAST_TAB_BYTES(SS), s>>>(g, M, boards, dice, max_depth, actions, values, tables); } while (0)
        switch (g.S) {
        case 5: PMF(5); break;
        case 6: PMF(6); break;
        case 7: PMF(7); break;
        default: PMF(8); break;
        }
#undef PMF
        return launch_status();
    }
    BY_NW(g, launch_minimax<NWV>(g, M, boards, dice, max_depth, heuristic, actions, values, s));
    return launch_status();
}

int ewn_predict_minimax_sim(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int max_depth, uint64_t key,
                            const uint32_t *obs_id, int8_t *actions, double *values, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (max_depth < 1) return EWN_EINVAL;
    if (max_depth > EWN_SIM_WINRATE_MAX_DEPTH || g.CN < 6) return EWN_EUNSUPPORTED;
    if (M == 0) return EWN_OK;
    if (!dice || !actions) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    BY_NW(g, rc = launch_minimax_sim<NWV>(g, M, boards, dice, nullptr, obs_id, key, max_depth, actions, values, s));
    return rc;
}

int ewn_predict_random(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, uint64_t key,
                       uint32_t step, const uint32_t *step_dev, int32_t lane_offset, int8_t *actions, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (M == 0) return EWN_OK;
    if (!dice || !actions) return EWN_ENULL;
    hipStream_t s = (hipStream_t)stream;
    BY_NW(g, (k_predict_random<NWV><<<GRID(M), BS, (size_t)BS * g.cells, s>>>(g, M, boards, dice, key, step, step_dev, lane_offset, actions)));
    return launch_status();
}

int ewn_predict_mcts(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int num_simulations,
                     int num_env_copies, uint64_t key, const uint32_t *obs_id, int8_t *actions, int32_t *wins, void *stream)
{
    Geom g;
    int rc = query_geom(board_size, cube_layer, M, boards, g);
    if (rc) return rc;
    if (num_simulations < 1 || num_env_copies < 1) return EWN_EINVAL;
    if (g.CN < 6) return EWN_EUNSUPPORTED;
    if (M == 0) return EWN_OK;
    if (!dice || !actions || !wins) return EWN_ENULL;
    return mcts_launch(g, M, boards, dice, nullptr, num_simulations * num_env_copies, key, obs_id, actions, wins, (hipStream_t)stream);
}

} // extern "C"
