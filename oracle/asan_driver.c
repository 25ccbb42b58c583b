/* asan_driver.c -- runs the CPU oracle (ewn_oracle.c, included as a translation unit) through every entry point under
 * AddressSanitizer + UndefinedBehaviorSanitizer.  TEST INFRASTRUCTURE: `make -C oracle asan-run`, tests/test_oracle_sanitizers.py.
 * (GPU AddressSanitizer is not available on the pool; the HIP side has the guard-zone test instead.) */
#include "ewn_oracle.c"
#include <stdio.h>

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static int run_env(int S, int L, int opp, int depth, int heur, int rng, int shaped, int n, int steps)
{
    oracle_cfg_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.board_size = S; cfg.cube_layer = L; cfg.n_lanes = n; cfg.opponent_kind = opp; cfg.max_depth = depth; cfg.heuristic = heur;
    cfg.num_simulations = 3; cfg.num_env_copies = 2; cfg.rng_kind = rng; cfg.shaped = shaped; cfg.illegal_move_tolerance = 3;
    cfg.autoreset = 1; cfg.shaped_refresh_on_reset = shaped; cfg.lane_offset = 5; cfg.seed_stride = (uint32_t)n;
    cfg.reward = 1.0; cfg.illegal_move_reward = -1.0; cfg.philox_key = 0x1234567ull;
    void *h = ewn_oracle_create(&cfg);
    if (!h) return 1;
    uint32_t *seeds = malloc(sizeof(uint32_t) * n);
    for (int i = 0; i < n; i++) seeds[i] = 0xFFFFFF00u + (uint32_t)i * 7u; /* wraps past 2^32 within a few episodes */
    ewn_oracle_reset(h, seeds, NULL);
    int8_t *b = malloc((size_t)n * S * S), *d = malloc(n), *a = malloc(2 * n), *tb = malloc((size_t)n * S * S), *td = malloc(n);
    double *r = malloc(sizeof(double) * n);
    uint8_t *te = malloc(n), *tr = malloc(n), *info = malloc(n);
    uint32_t s = 99;
    long terms = 0;
    for (int t = 0; t < steps; t++) {
        if (t % 3 == 0) ewn_oracle_random_actions(h, a);
        else if (t % 3 == 1) ewn_oracle_sample_legal_actions(h, (uint32_t)t, a);
        else for (int i = 0; i < n; i++) { a[2 * i] = (int8_t)(lcg(&s) % 3) - 0; a[2 * i + 1] = (int8_t)(lcg(&s) % 4); } /* illegal ones included */
        ewn_oracle_step(h, a, b, d, r, te, tr, info, tb, td);
        for (int i = 0; i < n; i++) terms += te[i];
    }
    double *ps = malloc(sizeof(double) * n);
    int32_t *tol = malloc(sizeof(int32_t) * n);
    uint64_t *dr = malloc(sizeof(uint64_t) * n);
    ewn_oracle_get_aux(h, ps, tol, dr);
    ewn_oracle_get_obs(h, b, d);
    ewn_oracle_set_obs(h, b, d);
    /* the stateless entry points on the positions reached */
    int8_t *acts = malloc((size_t)n * 12), *na = malloc(n), *cs = malloc(n), *cl = malloc(n);
    uint8_t *win = malloc(n);
    for (int pl = 1; pl <= 2; pl++) ewn_oracle_legal_actions(S, L, n, b, d, pl, acts, na, cs, cl, win);
    for (int i = 0; i < n; i++) if (d[i] > 6) d[i] = 6;
    if (L >= 3) {
        double *v = malloc(sizeof(double) * n);
        uint64_t *lv = malloc(sizeof(uint64_t) * n);
        int32_t *w6 = malloc(sizeof(int32_t) * 6 * n), *w1 = malloc(sizeof(int32_t) * n);
        for (int hh = 0; hh < 4; hh++) { ewn_oracle_evaluate(S, L, n, b, hh, v); ewn_oracle_predict_minimax(S, L, n, b, d, hh == 0 ? 3 : 2, hh, a, v, lv); }
        ewn_oracle_predict_minimax_sim(S, L, n < 4 ? n : 4, b, d, 2, 77, NULL, 10, a, v);
        ewn_oracle_predict_mcts(S, L, n, b, d, 3, 2, 5, NULL, a, w6);
        ewn_oracle_playout_wins(S, L, n, b, 2, 7, 9, w1);
        free(v); free(lv); free(w6); free(w1);
    }
    ewn_oracle_destroy(h);
    free(seeds); free(b); free(d); free(a); free(tb); free(td); free(r); free(te); free(tr); free(info); free(ps); free(tol); free(dr);
    free(acts); free(na); free(cs); free(cl); free(win);
    printf("S=%d L=%d opp=%d depth=%d heur=%d rng=%d shaped=%d: %ld terminations\n", S, L, opp, depth, heur, rng, shaped, terms);
    return 0;
}

int main(void)
{
    int rc = 0;
    rc |= run_env(5, 3, 0, 3, 0, 0, 0, 64, 60);
    rc |= run_env(5, 3, 1, 3, 0, 1, 1, 48, 30);
    rc |= run_env(5, 3, 1, 2, 4, 1, 0, 6, 6);      /* sim_winrate opponent */
    rc |= run_env(5, 3, 2, 3, 0, 0, 0, 16, 12);     /* flat Monte-Carlo opponent */
    rc |= run_env(5, 3, 1, 5, 0, 1, 0, 4, 5);
    rc |= run_env(7, 5, 1, 2, 2, 0, 1, 24, 30);
    rc |= run_env(4, 1, 0, 1, 0, 0, 0, 32, 40);
    rc |= run_env(8, 3, 1, 3, 3, 1, 0, 24, 20);
    rc |= run_env(11, 5, 1, 2, 1, 0, 0, 16, 40);
    rc |= run_env(9, 3, 2, 1, 0, 1, 0, 8, 10);
    uint8_t dd[64];
    uint32_t ff[64];
    ewn_oracle_prng_draws(1, 2, 3, 4, 64, dd, ff);
    uint32_t out[40];
    ewn_oracle_mt_outputs(9487, 40, out);
    int32_t lo[5] = { 1, 0, 1, 0, 3 }, hi[5] = { 7, 3, 2, 1, 4 }, o5[5];
    ewn_oracle_np_randint_seq(0, 5, lo, hi, o5);
    printf(rc ? "FAILED\n" : "sanitizer run complete\n");
    return rc;
}
