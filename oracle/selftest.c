/* Sanitizer self-test of the CPU oracle (test infrastructure): built with -fsanitize=address,undefined by
 * `make -C oracle selftest` and executed by tests/test_oracle_properties.py.  Walks every env class, every word
 * source and the observation restatements over many random steps; any out-of-bounds access or UB aborts. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "susnet_oracle.h"

static void fill_grid(so_config *c, int n, int walls) {
    c->grid_n = n;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) c->grid[i][j] = 1;
    if (walls) {
        int w = (n - 1) / 2;
        for (int i = 0; i < n; i++)
            if (i != (w - 1) / 2 && i != w + 1 + (n - 1 - w) / 2) c->grid[i][w] = c->grid[w][i] = 0;
    }
}

int main(void) {
    long total = 0;
    for (int variant = 0; variant < 3; variant++)
        for (int n = 9; n <= 16; n += 7)
            for (int kind = 0; kind < 3; kind++) {
                so_config c;
                memset(&c, 0, sizeof(c));
                c.variant = variant;
                c.n_imposters = variant == SO_VARIANT_ITG ? 1 : 3;
                c.n_crew = variant == SO_VARIANT_ITG ? 5 : 13;
                c.n_jobs = 16;
                fill_grid(&c, n, 1);
                c.kill_reward = -5; c.complete_job_reward = 3; c.sabotage_reward = 3; c.game_end_reward = 10;
                c.dead_penalty = -2; c.vote_reward = 3; c.max_time_steps = 50;
                c.is_action_order_random = 1; c.shuffle_imposter_index = 1; c.tag_reset_interval = 4;
                so_env *e = malloc(sizeof(so_env));
                if (so_env_init(e, &c) != SO_OK) { printf("init failed\n"); return 1; }
                uint32_t *tape = NULL;
                if (kind == SO_RNG_TAPE) {
                    so_seed_mt(e, 7);
                    tape = malloc(sizeof(uint32_t) * 200000);
                    for (int k = 0; k < 200000; k++) tape[k] = so_next_u32(e);
                    so_set_tape(e, tape, 200000);
                } else if (kind == SO_RNG_PHILOX) so_set_philox(e, 99, 12345678901ull, 0);
                else so_seed_mt(e, 7);
                so_reset(e);
                int32_t act[SO_MAX_AGENTS];
                double rew[SO_MAX_AGENTS], raw[5 * SO_MAX_AGENTS + 3 * SO_MAX_JOBS + 1];
                float sp[(SO_MAX_AGENTS + 2) * SO_MAX_GRID * SO_MAX_GRID], ns[3 * SO_MAX_AGENTS], flat[1024];
                int32_t comps[6] = {SO_F_ONEHOT_POS, SO_F_COORD_POS, SO_F_ALIVE_CREW, SO_F_WALLS3X3, SO_F_DIST_TO_IMP, SO_F_SCENT};
                for (int s = 0; s < 3000; s++) {
                    int32_t d, t;
                    so_sample_actions(e, act);
                    if (so_step(e, act, rew, &d, &t) != SO_OK) { printf("step failed\n"); return 1; }
                    so_obs_raw(e, raw);
                    so_obs_planes(e, sp, ns);
                    if (so_obs_flat(e, comps, 6, flat) < 0) { printf("flat failed\n"); return 1; }
                    if (d || t) so_reset(e);
                    total++;
                }
                if (e->rng.overflow) { printf("tape overflow\n"); return 1; }
                free(tape);
                free(e);
            }
    printf("selftest ok %ld steps\n", total);
    return 0;
}
