/*
 * susnet_oracle.h -- CPU restatement of the reference environment step/reset path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The product path (sus-net_amd/) never imports it.
 *
 * Parity pin: the reference has no tests of its own, so this restatement is pinned
 * against golden traces produced by running the unmodified reference in the build container
 * (tests/golden/generate_golden.py -> tests/golden/ *.npz); tests/test_oracle_golden.py replays every one
 * of them from the recorded numpy seed alone.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef SUSNET_ORACLE_H
#define SUSNET_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO_MAX_AGENTS 16
#define SO_MAX_JOBS 16
#define SO_MAX_GRID 16
#define SO_N_METRICS 13

/* variant: which reference class */
#define SO_VARIANT_BASE 0    /* FourRoomEnv            src/environment/base.py:102      */
#define SO_VARIANT_ITG 1     /* ImposterTrainingGround src/environment/pred_prey.py:20  */
#define SO_VARIANT_TAGGING 2 /* FourRoomEnvWithTagging src/environment/tagging.py:9     */

/* SusMetrics order (src/metrics.py:7-20) */
enum {
    SO_M_IMP_KILLED_CREW = 0,
    SO_M_IMP_VOTED_OUT = 1,
    SO_M_CREW_VOTED_OUT = 2,
    SO_M_SABOTAGED_JOBS = 3,
    SO_M_COMPLETED_JOBS = 4,
    SO_M_TOTAL_STALEMATES = 5,
    SO_M_TOTAL_TIME_STEPS = 6,
    SO_M_IMPOSTER_WON = 7,
    SO_M_CREW_WON = 8,
    SO_M_AVG_CREW_RETURNS = 9,
    SO_M_AVG_IMPOSTER_RETURNS = 10,
    SO_M_CREW_LOSS = 11,
    SO_M_IMPOSTER_LOSS = 12
};

/* Action enum values (src/environment/base.py:46-58) */
enum { SO_STAY = 0, SO_UP = 1, SO_DOWN = 2, SO_LEFT = 3, SO_RIGHT = 4, SO_KILL = 5, SO_FIX = 6, SO_SABOTAGE = 7 };

/* word sources */
#define SO_RNG_MT19937 0 /* numpy legacy global RandomState stream (reference behaviour) */
#define SO_RNG_TAPE 1    /* caller-supplied raw 32-bit words (what the HIP parity mode consumes) */
#define SO_RNG_PHILOX 2  /* Philox4x32-10 keyed (seed, env_id); the product's production stream */

typedef struct so_config {
    int32_t variant;
    int32_t n_imposters, n_crew, n_jobs;
    int32_t grid_n;                               /* N (reference: 9) */
    uint8_t grid[SO_MAX_GRID][SO_MAX_GRID];       /* grid[i][j] != 0 <=> free cell (base.py:195-197) */
    double kill_reward, complete_job_reward, sabotage_reward, time_step_reward;
    double game_end_reward, dead_penalty, vote_reward;
    int32_t max_time_steps;
    int32_t is_action_order_random;
    int32_t shuffle_imposter_index;
    int32_t tag_reset_interval;
} so_config;

typedef struct so_rng {
    int32_t kind;
    uint32_t mt[624];
    int32_t mti;
    const uint32_t *tape;
    int64_t tape_len;
    uint64_t seed;    /* philox key */
    uint64_t env_id;  /* philox counter hi */
    uint64_t cursor;  /* words consumed so far (all kinds) */
    uint64_t tick;    /* steps taken: index of the production ACTION stream */
    int32_t overflow; /* tape exhausted */
    uint32_t episode; /* resets drawn so far: index of the production RESET stream (Philox kind only) */
    uint32_t reset_pos; /* word position inside the reset being drawn (Philox kind only) */
    int32_t in_reset;
} so_rng;

typedef struct so_env {
    so_config cfg;
    so_rng rng;
    int32_t A, J;
    int32_t n_valid;
    uint8_t valid[SO_MAX_GRID * SO_MAX_GRID][2]; /* np.argwhere(grid) row-major (base.py:199) */
    int32_t pos[SO_MAX_AGENTS][2];               /* (x, y) */
    int32_t alive[SO_MAX_AGENTS];
    int32_t imp_mask[SO_MAX_AGENTS];
    int32_t imp_idxs[SO_MAX_AGENTS];             /* in the order numpy produced them */
    int32_t jobpos[SO_MAX_JOBS][2];
    int32_t jobdone[SO_MAX_JOBS];
    int32_t used[SO_MAX_AGENTS];                 /* tagging.py:30 */
    int32_t counts[SO_MAX_AGENTS];               /* tagging.py:29 */
    int32_t timer;                               /* tagging.py:31 */
    int32_t t;
    int32_t n_role_actions[SO_MAX_AGENTS];       /* len of the role part of agent_action_map[i] */
    int64_t metrics[SO_N_METRICS];
    double rewards[SO_MAX_AGENTS];
    int32_t order[SO_MAX_AGENTS];                /* order used by the last step */
    /* production (Philox) action stream: static packing of a tick's bounded draws into 32-bit words (so_action_layout) */
    int32_t aw_W;                                /* words a tick owns (A == 2: one word serves aw_tpw ticks) */
    int32_t aw_tpw;                              /* ticks per word: 3 for the 1v1 game, else 1 */
    uint8_t aw_word[2 * SO_MAX_AGENTS];          /* draw d (A action draws, then the A - 1 shuffle draws) -> word of the tick */
} so_env;

/* return codes of so_step */
#define SO_OK 0
#define SO_ERR_ASSERT (-1) /* reference AssertionError: action >= action_space.n (base.py:360) */
#define SO_ERR_INDEX (-2)  /* reference IndexError: role-invalid action index (base.py:381)     */
#define SO_ERR_CONFIG (-3)

int so_env_init(so_env *e, const so_config *cfg);
void so_seed_mt(so_env *e, uint32_t seed);                        /* np.random.seed(seed) */
void so_set_tape(so_env *e, const uint32_t *words, int64_t n_words);
void so_set_philox(so_env *e, uint64_t seed, uint64_t env_id, uint64_t cursor);
void so_set_tick(so_env *e, uint64_t tick);
void so_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]); /* raw block (KAT) */
uint32_t so_next_u32(so_env *e);                                  /* one raw word */
void so_reset(so_env *e);                                         /* base.py:251 / tagging.py:62 */
void so_sample_actions(so_env *e, int32_t *actions);              /* base.py:326 */
int so_step(so_env *e, const int32_t *actions, double *rewards, int32_t *done, int32_t *truncated);
int so_n_actions(const so_env *e, int agent);                     /* len(agent_action_map[agent]) */
int so_sizeof_env(void);

/* batched helpers (array of so_env); `threads` <= 0 means all cores (OpenMP) */
void so_batch_reset(so_env *envs, int64_t B, int threads);
int so_batch_step(so_env *envs, int64_t B, const int32_t *actions /*[B][A]*/, double *rewards /*[B][A]*/,
                  uint8_t *done, uint8_t *trunc, int threads);
void so_batch_sample_actions(so_env *envs, int64_t B, int32_t *actions);
void so_batch_reset_masked(so_env *envs, int64_t B, const uint8_t *mask);
void so_batch_obs_raw(const so_env *envs, int64_t B, uint8_t *out);
void so_batch_export(const so_env *envs, int64_t B, int32_t *pos, uint8_t *alive, uint8_t *imp, int32_t *jobpos,
                     uint8_t *jobdone, uint8_t *used, int32_t *counts, int32_t *timer, int32_t *t, int64_t *metrics,
                     uint64_t *cursor);
void so_batch_export_episode(const so_env *envs, int64_t B, uint32_t *episode); /* resets drawn so far (production RESET stream) */
void so_set_episode(so_env *e, uint32_t episode);
/* populate()-shaped random rollout (replay_memory.py:96-143 minus the buffer): per env
 * reset; repeat {sample_actions; step; reset on done|trunc}; returns env-steps taken (B*steps). */
int64_t so_batch_random_rollout(so_env *envs, int64_t B, int64_t steps, int threads, int64_t *episodes_out,
                                double *reward_sum_out);

/* observation restatements (src/features/component.py); out buffers are float32 */
int so_obs_flat_size(const so_env *e, const int32_t *components, int n_components);
int so_obs_flat(const so_env *e, const int32_t *components, int n_components, float *out);
void so_obs_planes(const so_env *e, float *spatial /*[A+2][N][N]*/, float *non_spatial /*[A(+A)+J]*/);
void so_obs_raw(const so_env *e, double *out /* flatten_state: 3A+3J (+2A+1 tagging) */);
int so_obs_raw_size(const so_env *e);

/* flat component ids (src/features/component.py line of the class) */
enum {
    SO_F_ONEHOT_POS = 0,   /* OneHotAgentPositionFeaturizer      component.py:221 */
    SO_F_COORD_POS = 1,    /* CoordinateAgentPositionsFeaturizer component.py:384 */
    SO_F_ALIVE_CREW = 2,   /* AliveCrewFeaturizer                component.py:406 */
    SO_F_L1_CREW = 3,      /* L1CrewFeaturizer                   component.py:428 */
    SO_F_CLOSEST_CREW = 4, /* ClosestAliveCrewFeaturizer         component.py:455 */
    SO_F_WALLS3X3 = 5,     /* WallsFeaturizer                    component.py:281 */
    SO_F_DIST_TO_IMP = 6,  /* DistanceToImposterFeaturizer       component.py:250 */
    SO_F_ROOM_LOC = 7,     /* ImposterVSCrewRoomLocaionFeaturizer component.py:303 (9x9 only) */
    SO_F_SCENT = 8         /* ImposterScentFeaturizer            component.py:339 */
};

#ifdef __cplusplus
}
#endif
#endif
