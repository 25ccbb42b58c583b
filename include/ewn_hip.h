/*
 * include/ewn_hip.h -- C ABI of libewn_hip.so: the MI355X (gfx950) vectorised
 * EinStein-wuerfelt-nicht environment step and opponent-search engine.
 *
 * The reference (jchen8tw/ewn-gym) is pure Python and has no FFI: its boundary
 * for this path is the Python class surface of envs/ewn.py, envs/minimax_ewn.py,
 * envs/training_ewn.py and classical_policies/{random_policy,minimax,mcts}.py.
 * Each entry point below names the reference method(s) it replaces, batched over
 * N independent games ("lanes").  The Python mirror of those classes (packages
 * envs/, classical_policies/, constants/ at the repo root) binds this ABI with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - Plain C: pointers + sizes only.  Every buffer pointer is a DEVICE pointer
 *    owned by the caller (e.g. torch.Tensor.data_ptr()); nothing is allocated,
 *    freed or synchronised inside a call, so calls are hipGraph-capturable.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *  - Return value: 0 on success, a negative EWN_E* code otherwise (never throws,
 *    never aborts).  ewn_strerror() maps a code to text.
 *  - No hidden global state; a call is thread-safe w.r.t. other calls that do
 *    not share buffers.
 *  - Boards are int8, row-major [lane][row][col], value k>0 = TOP_LEFT cube k,
 *    k<0 = BOTTOM_RIGHT cube |k|, 0 = empty (envs/ewn.py:49-58, 94-107).
 *  - Actions are int8 [lane][2] = {chose_larger in {0,1}, direction in {0,1,2}}
 *    (envs/ewn.py:61-62, 436-442); action buffers are 2-byte aligned, board / rng
 *    / table buffers 16-byte aligned (any torch allocation is).
 */
#ifndef EWN_HIP_H
#define EWN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EWN_ABI_VERSION 4 /* 2: ewn_step_k, ewn_predict_minimax_sim, ewn_lanes_per_game; six table images; boards up to 11x11.  3: EWN_AGENT_SAMPLE; 32 KB table images.
                             4: ewn_rollout_out.record, ewn_roll_dice, EWN_AGENT_MLP / ewn_policy, shaped env and MCTS opponent in ewn_step_k, ewn_a2c_* */

/* error codes */
#define EWN_OK 0
#define EWN_EINVAL (-1)      /* bad argument / unsupported configuration (the reference asserts, envs/ewn.py:47) */
#define EWN_ENULL (-2)       /* required pointer is NULL */
#define EWN_ELAUNCH (-3)     /* kernel launch failed (hipGetLastError != hipSuccess) */
#define EWN_EUNSUPPORTED (-4)/* valid in the reference, not built here (e.g. board_size > 11, max_depth > 6) */

/* opponent_kind: constants/policy.py:4-10 (uct / alpha_zero are out of scope) */
#define EWN_OPP_RANDOM 0
#define EWN_OPP_MINIMAX 1
#define EWN_OPP_MCTS 2

/* rng_kind */
#define EWN_RNG_MT19937 0 /* bit-exact numpy legacy global stream, one per lane (envs/ewn.py:91,490) */
#define EWN_RNG_PHILOX 1  /* Philox4x32-10 counter RNG, same masked-rejection randint */

/* heuristic: envs/minimax_ewn.py:29-38 */
#define EWN_H_HYBRID 0
#define EWN_H_MIN_DIST 1
#define EWN_H_TWO_MIN_DIST 2
#define EWN_H_ATTK 3
#define EWN_H_SIM_WINRATE 4 /* MinimaxEnv.simulate as the search leaf (envs/minimax_ewn.py:36-37, 215-238): 100 random playouts per
                               leaf; searches only (ewn_evaluate answers it through ewn_playout_wins); max_depth 5 / 6 run for seconds */
#define EWN_SIM_WINRATE_PLAYOUTS 100 /* MinimaxEnv.num_simulations, envs/minimax_ewn.py:20 */
#define EWN_SIM_WINRATE_MAX_DEPTH 6

/* info codes of ewn_step: the messages of envs/ewn.py:448,454,473,478 and envs/training_ewn.py:56 */
#define EWN_INFO_NONE 0
#define EWN_INFO_INVALID_PLAYER 1 /* "Invalid move for player! End the game." */
#define EWN_INFO_WON 2            /* "You won!" */
#define EWN_INFO_INVALID_OPP 3    /* "Invalid move for opponent! End the game." */
#define EWN_INFO_LOST 4           /* "You lost!" */
#define EWN_INFO_TOLERANCE 5      /* "Invalid move for player! Tolerance left {n}." */

#define EWN_MAX_BOARD 11  /* up to 8x8: 64-bit occupancy masks and the table-driven kernels; 9x9 .. 11x11: the generic kernels on a
                             mask-free state (7-bit positions) */
#define EWN_MAX_CUBES 15  /* cube_layer <= 5 */
#define EWN_MAX_DEPTH 6
#define EWN_MT_WINDOW_MAX 227 /* MT19937 outputs computable from the seeded state alone */
#define EWN_RNG_HEADER_WORDS 4

/* Replaces the constructor arguments of EinsteinWuerfeltNichtEnv (envs/ewn.py:35-42),
 * MiniMaxHeuristicEnv (envs/training_ewn.py:19-29) and the opponent policy ctor
 * kwargs (classical_policies/minimax.py:10-11, mcts.py:11-13). */
typedef struct ewn_config {
    int32_t board_size;             /* S, 3..EWN_MAX_BOARD */
    int32_t cube_layer;             /* L, cube_num = L(L+1)/2, L < S-1 */
    int32_t n_lanes;                /* N parallel games handled by this call */
    int32_t opponent_kind;          /* EWN_OPP_* */
    int32_t max_depth;              /* ExpectiMinimaxAgent.max_depth, 1..EWN_MAX_DEPTH */
    int32_t heuristic;              /* EWN_H_* */
    int32_t num_simulations;        /* MctsAgent.num_simulations */
    int32_t num_env_copies;         /* MctsAgent.num_env_copies */
    int32_t rng_kind;               /* EWN_RNG_* */
    int32_t shaped;                 /* 0: EinsteinWuerfeltNichtEnv.step, 1: MiniMaxHeuristicEnv.step */
    int32_t illegal_move_tolerance; /* initial tolerance (training_ewn.py:38); used by ewn_init_aux */
    int32_t autoreset;              /* 1: a terminated lane is reset (next seed) inside ewn_step */
    int32_t shaped_refresh_on_reset;/* 0 = reference behaviour: prev_score is set in the ctor only */
    int32_t lane_offset;            /* global id of lane 0 (multi-GPU sharding); enters the Philox agent/MCTS streams */
    uint32_t seed_stride;           /* auto-reset: seed += seed_stride per episode */
    uint32_t mt_window;             /* MT19937 outputs precomputed per episode, 16..EWN_MT_WINDOW_MAX (0 = 128) */
    double reward;                  /* envs/ewn.py:38 (goal_reward of the shaped env) */
    double illegal_move_reward;     /* training_ewn.py:28 */
    uint64_t philox_key;
} ewn_config;

/* Per-lane state, structure-of-arrays, all device pointers, caller-owned. */
typedef struct ewn_state {
    int8_t *board;       /* [N][S*S]  the observation "board" (agent = TOP_LEFT to move) */
    int8_t *dice;        /* [N]       the observation "dice_roll" */
    uint8_t *done;       /* [N]       1 = terminated and not yet reset (lane frozen) */
    uint32_t *rng;       /* N*ewn_rng_words() words: [N][4] headers {seed, draw index, next_seed, flags}, then (MT kind)
                            [N][3][W] windows of precomputed MT19937 outputs (this episode's and the next two) and
                            [N] reset epochs.  Opaque to the caller; zero-initialise, then ewn_init_aux. */
    double *prev_score;  /* [N]  shaped env only (training_ewn.py:35), may be NULL otherwise */
    int32_t *tolerance;  /* [N]  shaped env only (training_ewn.py:38), may be NULL otherwise */
    const void *tables;  /* device copy of ewn_build_tables() output, or NULL.  When present and the config is
                            (cube_layer 3, un-shaped; RandomAgent opponent, or minimax with max_depth 1..6 and
                            'hybrid') ewn_step runs the specialised table-driven kernel; results are identical
                            either way. */
} ewn_state;

/* Outputs of one step, device pointers, caller-owned.  The post-step observation is
 * written in place to ewn_state.board / .dice. */
typedef struct ewn_step_out {
    double *reward;        /* [N] */
    uint8_t *terminated;   /* [N] */
    uint8_t *truncated;    /* [N] */
    uint8_t *info;         /* [N] EWN_INFO_* */
    int8_t *terminal_board;/* [N][S*S] observation before auto-reset (SB3 "terminal_observation"); NULL to skip */
    int8_t *terminal_dice; /* [N] ; NULL to skip */
    int8_t *random_action; /* [N][2] ; NULL to skip.  RandomAgent.predict (classical_policies/random_policy.py:11-15) on the
                              POST-step observation, fused into the step: a uniformly random legal action of the agent,
                              drawn from a hash of (episode seed, draws so far, global lane id, philox_key).  May alias
                              `actions` (each lane reads its action before it writes the next one). */
} ewn_step_out;

int ewn_abi_version(void);
const char *ewn_strerror(int code);

/* number of uint32 words per lane in ewn_state.rng for this config (<0 on error) */
int ewn_rng_words(const ewn_config *cfg);
/* bytes of device scratch ewn_step needs for this config (0 if none; <0 on error): the MCTS split-phase buffers, or
 * (MT kind with auto-reset) the queue through which one step launch hands window refills to the next.  The scratch
 * must be zero-initialised once and then left alone between calls. */
int64_t ewn_step_scratch_bytes(const ewn_config *cfg);

/* Search and move-selection tables of the specialised kernel (leaf-value ranks, ring-order geometry, dice -> cube
 * selectors; two images: max_depth 1-3 and 5, and max_depth 4 and 6, whose leaves are six-dice averages of evaluate();
 * ewn_gym_amd/csrc/ewn_fast.hpp).  Pure host computation: ewn_tables_bytes() gives the size
 * (0 = no specialised kernel for this geometry), ewn_build_tables() fills a HOST buffer the
 * caller then copies to the device and passes as ewn_state.tables / the `tables` argument. */
int64_t ewn_tables_bytes(int board_size, int cube_layer);
int ewn_build_tables(int board_size, int cube_layer, void *host_out);

/* Constructor-time state that reset() does not touch in the reference:
 * prev_score = evaluate(initial board) (training_ewn.py:35) and the tolerance
 * counter (:38); also clears `done` and zeroes the rng header. */
int ewn_init_aux(const ewn_config *cfg, const ewn_state *st, void *stream);

/* EinsteinWuerfeltNichtEnv.reset(seed) (envs/ewn.py:488-494) + setup_game (:94-108)
 * for every lane with lane_mask[i] != 0 (NULL = all lanes).  seeds[i] (NULL = the
 * lane's stored next_seed) is the argument of np.random.seed. */
int ewn_reset(const ewn_config *cfg, const ewn_state *st, const uint32_t *seeds, const uint8_t *lane_mask, void *stream);

/* EinsteinWuerfeltNichtEnv.roll_dice (envs/ewn.py:90-92): dice_roll = np.random.randint(1, cube_num + 1), one draw from the
 * lane's own dice stream, for every lane with lane_mask[i] != 0 (NULL = all lanes) whose game is not finished.  The step
 * and reset entry points roll their dice themselves; this is the public method on its own. */
int ewn_roll_dice(const ewn_config *cfg, const ewn_state *st, const uint8_t *lane_mask, void *stream);

/* EinsteinWuerfeltNichtEnv.step (envs/ewn.py:436-486) or, with cfg->shaped,
 * MiniMaxHeuristicEnv.step (envs/training_ewn.py:43-99): agent move, win test,
 * opponent dice + reply (opponent_action, ewn.py:289-296, by cfg->opponent_kind),
 * win test, next dice; optional auto-reset.  `scratch` = ewn_step_scratch_bytes() bytes. */
int ewn_step(const ewn_config *cfg, const ewn_state *st, const int8_t *actions, const ewn_step_out *out,
             void *scratch, void *stream);

/* ---- K env steps per launch, the agent played by the engine too --------------------------------------------------
 * Replaces the evaluation loop of eval_minimax.py:16-50 / eval_pairs.py:10-35
 *     while not done: action, _ = agent.predict(obs); obs, reward, done, trunc, info = env.step(action)
 * (and a rollout collector's inner loop) when `agent` is one of the classical policies: the state is read once, stays in
 * registers for K steps and is written once; results are identical, step for step, to K ewn_step calls with the agent's
 * action fed back.  Available for the table-driven configurations (cube_layer 3, board sizes 5..8, ewn_state.tables set,
 * un-shaped, opponent RandomAgent or 'hybrid' minimax); MT19937-compat dice only without auto-reset. */
#define EWN_AGENT_RANDOM 0  /* RandomAgent.predict (classical_policies/random_policy.py:11-15): the hash-driven uniform legal pick
                               of ewn_step_out.random_action, same stream */
#define EWN_AGENT_MINIMAX 1 /* ExpectiMinimaxAgent(agent_max_depth, 'hybrid').predict (classical_policies/minimax.py:89-93) */
#define EWN_AGENT_SAMPLE 2  /* env.action_space.sample() on MultiDiscrete([2, 3]) (envs/ewn.py:59): uniform over all six actions, illegal ones
                             * included -- what an untrained policy plays; hash-driven like EWN_AGENT_RANDOM, agent_max_depth ignored */

typedef struct ewn_rollout_out {
    /* trajectory, row k = step k of this call; every pointer may be NULL (that column is not written) */
    int8_t *board;       /* [K][N][S*S] observation after step k (after the auto-reset, like ewn_state.board after ewn_step) */
    int8_t *dice;        /* [K][N] */
    int8_t *action;      /* [K][N][2] the action the agent played at step k */
    double *reward;      /* [K][N] */
    uint8_t *terminated; /* [K][N] */
    uint8_t *truncated;  /* [K][N] */
    uint8_t *info;       /* [K][N] EWN_INFO_* */
    /* per-lane totals over the K steps, ADDED to what the buffers hold; every pointer may be NULL */
    double *return_sum;  /* [N] sum of rewards */
    int32_t *n_steps;    /* [N] steps played (a finished, un-reset lane plays none) */
    int32_t *n_episodes; /* [N] episodes finished */
    int32_t *n_wins;     /* [N] of which won ("You won!", envs/ewn.py:454) */
    /* The same trajectory as ONE record per lane-step, [K][N][EWN_TRAJ_RECORD_STRIDE(S)] bytes, 16-byte aligned; may be NULL.
     * Record = board int8 [S*S] | dice | action[2] | terminated | truncated | info | zero padding: a lane-step is one or two whole
     * 32-byte sectors written with 16-byte stores (32 bytes for 5x5, 64 for 7x7) instead of a 25-byte row at an odd offset plus
     * five 1-2 byte columns.  The reward stays in its own f64 column.  Independent of the columns above (any subset may be asked for). */
    uint8_t *record;
} ewn_rollout_out;
#define EWN_TRAJ_RECORD_STRIDE(S) ((((S) * (S)) + 6 + 15) & ~15)
#define EWN_TRAJ_REC_DICE(S) ((S) * (S))          /* byte offsets inside a record */
#define EWN_TRAJ_REC_ACTION(S) ((S) * (S) + 1)
#define EWN_TRAJ_REC_TERMINATED(S) ((S) * (S) + 3)
#define EWN_TRAJ_REC_TRUNCATED(S) ((S) * (S) + 4)
#define EWN_TRAJ_REC_INFO(S) ((S) * (S) + 5)

/* Introspection: how many lanes of a wavefront share one game in the table-driven kernel this configuration runs -- entry 0:
 * ewn_step (given ewn_state.tables), entry 1: ewn_step_k with the RandomAgent agent; 0 = a generic kernel (one thread per game).
 * The choice depends on n_lanes only (measured thresholds) unless the tuning variables EWN_D3_T /
 * EWN_ROLLOUT_T are set in the environment; tests pin the defaults. */
int ewn_lanes_per_game(const ewn_config *cfg, int entry);

/* 1 if ewn_step_k serves this configuration and agent, 0 if not, < 0 on an invalid configuration.  Geometries without a table
 * image (cube_layer 4 / 5, boards of 9x9 .. 11x11) are served by the generic one-thread-per-game K-step kernel for the RandomAgent /
 * sample agents against RandomAgent or minimax opponents of the four evaluate() heuristics (no ewn_state.tables needed). */
int ewn_step_k_supported(const ewn_config *cfg, int agent_kind, int agent_max_depth);
/* K >= 1 steps of every lane; out may be NULL (only the state advances).  No scratch; one kernel launch. */
int ewn_step_k(const ewn_config *cfg, const ewn_state *st, int K, int agent_kind, int agent_max_depth,
               const ewn_rollout_out *out, void *stream);

/* ---- K env steps per launch with the TRAINED policy as the agent: the rollout collector of train.py:35-63, 134, 148 -----------
 * (SB3 A2C("MultiInputPolicy", env, policy_kwargs=dict(activation_fn=Tanh)).learn -> collect_rollouts over SubprocVecEnv workers)
 * as one kernel: observation -> features (S*S board cells as floats ++ one_hot(dice_roll - 1), width cube_num + 1 = 7,
 * envs/ewn.py:66-68) -> policy network (SB3's default net_arch: two separate hidden-64-64 tanh bodies, 5 logits for
 * MultiDiscrete([2, 3]) and a scalar value) on the matrix cores in exact fp32 -> Gumbel-max sample -> env step (plain or
 * cfg->shaped: envs/training_ewn.py:43-99) -> opponent reply -> auto-reset.  Served for cube_layer 3, board sizes 5 and 7,
 * opponent RandomAgent or minimax max_depth 1..4 with a (level, count) heuristic image, Philox dice. */
#define EWN_AGENT_MLP 3     /* agent_kind of ewn_step_k_supported for this path (ewn_step_k itself takes no parameters: use ewn_step_k_policy) */
#define EWN_POLICY_HIDDEN 64
#define EWN_POLICY_LOGITS 5

typedef struct ewn_policy {
    const float *params;        /* [ewn_policy_param_count()] fp32, device: body pi {W1 [64][F], b1 [64], W2 [64][64], b2 [64]}, body vf
                                   {same}, action head {W [5][64], b [5]}, value head {W [1][64], b [1]}; F = S*S + 7; row-major
                                   [out][in] like torch.nn.Linear.weight -- the order of a2c.ActorCritic.parameters() */
    int32_t deterministic;      /* 1: argmax of the logits (model.predict(deterministic=True), train.py:93) instead of sampling */
    int32_t record_initial_obs; /* 1: ewn_rollout_out.record has K + 1 rows, row 0 = the observation before step 0 (meta bytes:
                                   its dice, zeros) and row k + 1 = step k; what an n-step update needs (s_0 .. s_K) */
    uint64_t noise_key;         /* keys the sampling noise together with cfg->philox_key, the global lane id, the episode seed
                                   and the episode's draw count (the hash stream of ewn_step_out.random_action) */
    /* per-step outputs of the policy, [K][N]...; each may be NULL */
    float *logits;              /* [K][N][5] */
    float *value;               /* [K][N]; non-NULL makes the kernel evaluate the value body too */
    float *noise;               /* [K][N][5] the uniforms u in (0, 1) behind the Gumbel noise -log(-log u) of that step */
} ewn_policy;

/* number of fp32 parameters of the actor-critic for this geometry (< 0: not served) */
int64_t ewn_policy_param_count(int board_size, int cube_layer);
/* K >= 1 steps of every lane, the agent's action sampled from the policy; out may be NULL.  One kernel launch, no scratch. */
int ewn_step_k_policy(const ewn_config *cfg, const ewn_state *st, int K, const ewn_policy *pol, const ewn_rollout_out *out, void *stream);

/* ---- the A2C update on the records of ewn_step_k_policy: stable_baselines3 A2C.train as train.py:35-63, 148 configures it ----
 * (n-step returns = GAE with lambda 1, no advantage normalisation; loss = policy gradient + vf_coef * MSE(returns, values) +
 * ent_coef * (-entropy), a mean over the n_steps x lanes batch; clip_grad_norm_(max_grad_norm); RMSprop(alpha, eps)).  SB3 is not
 * vendored: parity with it is unpinned, the arithmetic is checked against torch autograd of the same loss.  Forward (recomputed from
 * the records: the parameters have not changed since the rollout) and backward run on the bf16 matrix pipe with every operand split
 * into three bf16 parts (six products per multiply: fp32 accuracy, measured against float64 -- tools/a2c_accuracy.py). */
typedef struct ewn_a2c_hyper {
    float gamma;            /* 0.99 */
    float vf_coef;          /* 0.5 */
    float ent_coef;         /* 0.0 */
    float max_grad_norm;    /* 0.5; <= 0: no clipping */
    float learning_rate;    /* SB3 A2C default 7e-4; train.py passes its own (3e-4) */
    float rms_alpha;        /* 0.99 */
    float rms_eps;          /* 1e-5 */
    int32_t world_size;     /* ewn_a2c_apply divides the (all-reduced, summed) gradient by it */
} ewn_a2c_hyper;

/* bytes of device scratch ewn_a2c_grad needs (advantages [K][N] + per-block partial gradients); < 0: configuration not served */
int64_t ewn_a2c_scratch_bytes(const ewn_config *cfg, int K);
/* record: [K + 1][N][EWN_TRAJ_RECORD_STRIDE(S)] written by ewn_step_k_policy with record_initial_obs = 1; reward [K][N].
 * grad [ewn_policy_param_count() + 8]: the gradient of THIS rank's mean loss in the layout of ewn_policy.params, then the loss sums
 * {policy, value, entropy, 0} of the policy pass and of the value pass (divide by K * N for means).  Three launches. */
int ewn_a2c_grad(const ewn_config *cfg, int K, const uint8_t *record, const double *reward, const float *params, const ewn_a2c_hyper *hp,
                 float *grad, void *scratch, void *stream);
/* clip by the global norm, then one RMSprop step on params / sq_avg (both [param count], in place); grad_norm_out (may be NULL)
 * receives the norm before clipping.  A multi-GPU job all-reduces (sums) grad between the two calls -- the one collective. */
int ewn_a2c_apply(const ewn_config *cfg, float *params, float *sq_avg, const float *grad, const ewn_a2c_hyper *hp, float *grad_norm_out,
                  void *stream);

/* ---- stateless policy / rule queries on M given observations (canonical: TOP_LEFT to move) ---- */

/* get_legal_actions (envs/ewn.py:338-375), find_cube_to_move (:178-215), check_win (:131-142).
 * player: 1 TOP_LEFT, 2 BOTTOM_RIGHT (constants/player.py).  Outputs (each may be NULL):
 * acts [M][6][2] (-1 padded, reference order), n_acts [M], cube_small/cube_large [M]
 * (cube NUMBER moved with flag 0 / 1), win [M]. */
int ewn_legal_actions(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int player,
                      int8_t *acts, int8_t *n_acts, int8_t *cube_small, int8_t *cube_large, uint8_t *win, void *stream);

/* make_simulated_action (envs/ewn.py:377-412) on M positions: player (1 TOP_LEFT / 2 BOTTOM_RIGHT) moves the cube its
 * dice selects ([flag, dir] as in step).  new_boards [M][S*S] receives the position after the move (unchanged when the
 * move leaves the board); valid [M] (may be NULL) is 1 for a legal move.  The host keeps the undo history. */
int ewn_apply_action(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int player,
                     const int8_t *actions, int8_t *new_boards, uint8_t *valid, void *stream);

/* MinimaxEnv.simulate (envs/minimax_ewn.py:215-238), the 'sim_winrate' heuristic: n_sims uniformly random playouts
 * from each position, `first_player` (1/2) moving first; wins [M] = playouts TOP_LEFT won.  Statistical parity only
 * (the reference draws from an unseeded Python `random`); randomness as in ewn_predict_mcts with block {0, m, 0, 'SIMU'}
 * and playout number r. */
int ewn_playout_wins(int board_size, int cube_layer, int M, const int8_t *boards, int first_player, int n_sims,
                     uint64_t key, int32_t *wins, void *stream);

/* MinimaxEnv.evaluate(heuristic) (envs/minimax_ewn.py:29-213) */
int ewn_evaluate(int board_size, int cube_layer, int M, const int8_t *boards, int heuristic, double *out, void *stream);

/* ExpectiMinimaxAgent.predict (classical_policies/minimax.py:89-93): actions [M][2];
 * values [M] = root value (may be NULL); tables: see ewn_build_tables (may be NULL).
 * A position that is already won/lost returns action {-1,-1} and value = evaluate(). */
int ewn_predict_minimax(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int max_depth,
                        int heuristic, int8_t *actions, double *values, const void *tables, void *stream);

/* The same with heuristic = 'sim_winrate' and an explicit key / per-observation stream id for the playouts' randomness (one
 * Philox block {0, obs_id (NULL = index), 0, 'SIMW'} per observation seeds one generator, drawn from in the order of the
 * depth-first search).  values [M] = root value (may be NULL).  Statistical parity with the reference (unseeded Python
 * `random`, envs/minimax_ewn.py:222-225); bit-exact with oracle/ewn_oracle.c, which mirrors it. */
int ewn_predict_minimax_sim(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, int max_depth, uint64_t key,
                            const uint32_t *obs_id, int8_t *actions, double *values, void *stream);

/* RandomAgent.predict (classical_policies/random_policy.py:11-15) as a stateless policy:
 * uniform legal action from Philox ctr={step + (step_dev ? *step_dev : 0), lane_offset+i, 'AGNT', 0}, key.
 * step_dev (device pointer, may be NULL) lets a captured hipGraph advance the stream between replays. */
int ewn_predict_random(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice, uint64_t key,
                       uint32_t step, const uint32_t *step_dev, int32_t lane_offset, int8_t *actions, void *stream);

/* MctsAgent.predict (classical_policies/mcts.py:102-106, flat Monte-Carlo :47-69, rollouts :21-45).
 * wins [M][6] int32 is REQUIRED scratch/output (win count per root move, -1 = no such move).
 * obs_id [M] (NULL = 0..M-1) and key select the rollout randomness: one Philox block {0, obs_id, 0, 'MCTS'} per observation
 * gives a word; rollout r of root move i runs a 32-bit LCG started at fmix32(word + (i * total + r) * 0x9E3779B9), one draw
 * per ply (dice and move index).  The result does not depend on how rollouts are distributed over lanes.  Statistical parity
 * with the reference (never-seeded Python `random`, mcts.py:29-32); bit-exact with oracle/ewn_oracle.c, which mirrors it. */
int ewn_predict_mcts(int board_size, int cube_layer, int M, const int8_t *boards, const int8_t *dice,
                     int num_simulations, int num_env_copies, uint64_t key, const uint32_t *obs_id, int8_t *actions,
                     int32_t *wins, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EWN_HIP_H */
