/*
 * susnet.h -- C ABI of libsusnet_hip.so: batched, MI355X-native (gfx950) implementation of the Sus-Net
 * grid-world environment hot path (reset / sample_actions / step / fused random rollout / observation
 * extraction), stepping `batch` independent episodes in lockstep, one wavefront lane per environment.
 *
 * The reference is a pure-Python gymnasium.Env (no FFI of its own), so every entry point names the
 * reference METHOD it replaces (paths relative to the reference repo root):
 *
 *   susnet_create          FourRoomEnv.__init__            src/environment/base.py:103-228
 *                          ImposterTrainingGround.__init__ src/environment/pred_prey.py:26-76
 *                          FourRoomEnvWithTagging.__init__ src/environment/tagging.py:10-60
 *   susnet_reset           FourRoomEnv.reset               src/environment/base.py:251-324 (tagging.py:62-101)
 *   susnet_sample_actions  FourRoomEnv.sample_actions      src/environment/base.py:326-330
 *   susnet_policy_actions  run_game's acting step          src/visualize.py:547-562 (train.py:355-381, epsilon = 0)
 *   susnet_qnet_*          MLP.forward on the flat features src/models/dqn.py:72-108, 322-329
 *   susnet_policy_step     argmax per team + env.step        src/visualize.py:547-582
 *   susnet_step            FourRoomEnv.step                src/environment/base.py:332-407 (tagging.py:120-235)
 *                          + _agent_step 462-533, check_win_condition 409-460 (pred_prey.py:78-99),
 *                            _merge_rewards 553-563, EnvMetricHandler src/metrics.py:35-64
 *   susnet_rollout         ReplayBuffer.populate's loop    src/replay_memory.py:96-143 (minus the buffer)
 *   susnet_ring_append     ReplayBuffer.add x (T x B)      src/replay_memory.py:50-73, with populate's sequence window (108-136)
 *   susnet_observe         flatten_state base.py:234; FlatFeaturizer / GlobalFeaturizer
 *                          src/features/model_ready.py:219-370, src/features/component.py:83-482
 *   susnet_featurize       SequenceStateFeaturizer.fit(state_sequence[B,T,S]) on windows / replay batches
 *                          src/features/model_ready.py:41-57, 254-289, 338-354 (callers: src/train.py:345-347)
 *   susnet_scent           ImposterScentFeaturizer.extract_features   src/features/component.py:336-380
 *   susnet_seed / _tick    np.random.seed(seed)            src/environment/base.py:126,267 (production stream)
 *   susnet_device_tick     (new) step counter in device memory: captured launches replay as a hipGraph
 *   susnet_bind_tape       (numpy's own MT19937 words: decisions equal the reference's for that seed)
 *   susnet_export_state    the state tuple step()/reset() return (base.py:317-324, 397-402)
 *   susnet_import_state    direct assignment to env.agent_positions etc. (what callers/tests do)
 *
 * Conventions
 *   - plain C types only; every device buffer is a caller-owned pointer (the Python host passes
 *     torch-ROCm tensor data_ptr()s); the library never allocates or frees device memory.
 *   - every call returns 0 on success or a negative SUSNET_E_* code; susnet_last_error() gives text.
 *   - all launches are asynchronous on the hipStream_t passed as `stream` (void*; NULL = default).
 *   - per-environment input errors (reference AssertionError / IndexError) are recorded in a device
 *     error word and reported by susnet_poll_errors(); an environment with a bad action is not stepped.
 *   - a handle is owned by one host thread; distinct handles are independent.
 */
#ifndef SUSNET_H
#define SUSNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUSNET_ABI_VERSION 6

#define SUSNET_MAX_AGENTS 16
#define SUSNET_MAX_JOBS 16
#define SUSNET_MAX_GRID 16
#define SUSNET_N_METRICS 13 /* SusMetrics, src/metrics.py:7-20, in declaration order */
#define SUSNET_N_LIFETIME 12

/* variant = reference class */
#define SUSNET_VARIANT_BASE 0
#define SUSNET_VARIANT_ITG 1
#define SUSNET_VARIANT_TAGGING 2

/* word source of all random decisions */
#define SUSNET_RNG_TAPE 1   /* caller-supplied raw 32-bit words consumed with numpy-legacy semantics:
                               given numpy's MT19937 output for seed s, results equal the reference's */
#define SUSNET_RNG_PHILOX 2 /* Philox4x32-10 keyed (seed, global env id): production stream */

/* error codes */
#define SUSNET_OK 0
#define SUSNET_E_INVALID (-1)  /* bad argument / config (reference ctor asserts, base.py:243-249) */
#define SUSNET_E_HIP (-2)      /* a HIP call failed */
#define SUSNET_E_STATE (-3)    /* state not bound */
#define SUSNET_E_ACTION_ASSERT (-4) /* some env got action >= action_space.n (base.py:360-362) */
#define SUSNET_E_ACTION_INDEX (-5)  /* some env got a role-invalid action index (base.py:379-382) */
#define SUSNET_E_TAPE (-6)     /* random tape exhausted */
#define SUSNET_E_ROW (-7)      /* susnet_featurize: a state row holds a coordinate outside the grid */

/* bits of the device error word */
#define SUSNET_ERRBIT_ASSERT 1u
#define SUSNET_ERRBIT_INDEX 2u
#define SUSNET_ERRBIT_TAPE 4u
#define SUSNET_ERRBIT_ROW 8u

/* dtypes of caller buffers */
#define SUSNET_U8 0
#define SUSNET_I32 1
#define SUSNET_I64 2
#define SUSNET_F32 3
#define SUSNET_F64 4

/* action / reward buffer layouts */
#define SUSNET_LAYOUT_AB 0 /* [A][B]  agent-major (SoA, what the kernels prefer) */
#define SUSNET_LAYOUT_BA 1 /* [B][A]  env-major (what reference callers index as actions[b]) */

/* observation modes (susnet_obs_spec.mode) */
#define SUSNET_OBS_NONE 0
#define SUSNET_OBS_RAW 1    /* flatten_state: [x0,y0,..., alive.., jobxy.., jobdone.. (, used, counts, timer_left)] */
#define SUSNET_OBS_FLAT 2   /* concatenation of flat components, FlatFeaturizer */
#define SUSNET_OBS_PLANES 3 /* GlobalFeaturizer: spatial [A+2][N][N] + non_spatial [A(+A)+J] */
#define SUSNET_OBS_PERSP 4  /* PerspectiveFeaturizer (model_ready.py:82-216): per agent i the planes with the agent channels in
                               the order i, 0, .., i-1, i+1, .., A-1 (then the two job channels): spatial [A][A+2][N][N];
                               non_spatial [A][A(+A)+J] = alive (and tag counts) in that same agent order, then job status */

/* flat components (src/features/component.py) */
#define SUSNET_F_ONEHOT_POS 0
#define SUSNET_F_COORD_POS 1
#define SUSNET_F_ALIVE_CREW 2
#define SUSNET_F_L1_CREW 3
#define SUSNET_F_CLOSEST_CREW 4
#define SUSNET_F_WALLS3X3 5
#define SUSNET_F_DIST_TO_IMP 6
#define SUSNET_F_ROOM_LOC 7

/* lifetime accumulator rows (per env, summed over finished episodes) */
#define SUSNET_L_EPISODES 0
#define SUSNET_L_CREW_WON 1
#define SUSNET_L_IMPOSTER_WON 2
#define SUSNET_L_TRUNCATED 3
#define SUSNET_L_KILLS 4
#define SUSNET_L_COMPLETED_JOBS 5
#define SUSNET_L_SABOTAGED_JOBS 6
#define SUSNET_L_IMP_VOTED_OUT 7
#define SUSNET_L_CREW_VOTED_OUT 8
#define SUSNET_L_EPISODE_STEPS 9
#define SUSNET_L_ENV_STEPS 10 /* every step the env took, finished episode or not (k_step: +1, a fused rollout: +n_ticks) */
#define SUSNET_L_RESERVED 11

typedef struct susnet_env susnet_env; /* opaque host-side handle */

/* Constructor arguments: the reference's ctor kwargs (base.py:103-120; pred_prey.py:26-38;
 * tagging.py:10-12) plus what a batched device env needs. */
typedef struct susnet_config {
    uint32_t struct_bytes; /* = sizeof(susnet_config) */
    uint32_t abi_version;  /* = SUSNET_ABI_VERSION */
    int32_t variant;
    int32_t batch;         /* environments on THIS device */
    int32_t n_imposters, n_crew, n_jobs;
    int32_t grid_n;        /* N; reference hard-codes 9 (base.py:195,206-207) */
    uint16_t grid_rows[SUSNET_MAX_GRID]; /* bit j of grid_rows[i] = grid[i][j], 1 = free (base.py:195-197) */
    double kill_reward, complete_job_reward, sabotage_reward, time_step_reward;
    double game_end_reward, dead_penalty, vote_reward;
    int32_t max_time_steps;
    int32_t is_action_order_random;
    int32_t shuffle_imposter_index;
    int32_t tag_reset_interval;
    int32_t auto_reset;    /* 1: an env that ends (done|truncated) is reset inside the same launch */
    int32_t rng_mode;      /* SUSNET_RNG_* */
    uint64_t seed;         /* Philox key */
    uint64_t env_id_base;  /* global id of local env 0 (sharding across GPUs keeps streams identical) */
    int32_t device;        /* HIP device ordinal the buffers live on */
    int32_t reserved;
} susnet_config;

/* Sizes the caller must allocate. All device buffers are raw bytes. */
typedef struct susnet_layout {
    uint64_t state_bytes;     /* the compact SoA state blob (see DESIGN.md "Data layout in HBM") */
    uint64_t state_align;     /* required alignment of the blob (256) */
    int32_t batch_padded;     /* SoA row stride in elements */
    int32_t n_agents;
    int32_t n_actions_imposter; /* len(agent_action_map[i]) for an imposter / a crew member */
    int32_t n_actions_crew;
    int32_t action_space_n;   /* Discrete(n) of the reference (8, or 8 + A with tagging) */
    int32_t obs_raw_size;     /* flattened_state_size (base.py:230-232) */
    int32_t envs_per_wave;    /* environments a wavefront of the fused rollout serves (64, 32 or 16, by batch size) */
    uint32_t test_overrides;  /* SUSNET_OVERRIDE_* bits: which test hooks were found in the environment at susnet_create */
} susnet_layout;

/* Test hooks.  Four environment variables change how a handle launches its kernels (never what it computes); they are read
 * ONCE, in susnet_create, recorded in susnet_layout.test_overrides, and named in susnet_last_error() messages of that handle:
 *   SUSNET_FORCE_GENERIC=1      every configuration runs the generic (LDS-table) kernels, none of the compiled-in ones
 *   SUSNET_EPW=16|32|64         environments per wave of the fused rollout
 *   SUSNET_TRAJ_MAX_BYTES=n     default of susnet_set_launch_limit()
 *   SUSNET_RING_TILE=0|8|16|32  susnet_ring_append: environments of a wave's (ticks x envs) tile; 0 (default) = 64 consecutive rows per wave */
#define SUSNET_OVERRIDE_FORCE_GENERIC 1u
#define SUSNET_OVERRIDE_EPW 2u
#define SUSNET_OVERRIDE_TRAJ_MAX_BYTES 4u
#define SUSNET_OVERRIDE_RING_TILE 8u

typedef struct susnet_obs_spec {
    int32_t mode;                 /* SUSNET_OBS_* */
    int32_t dtype;                /* SUSNET_F32 (reference layouts) or SUSNET_U8 (compact) */
    int32_t n_components;         /* FLAT only */
    int32_t components[16];       /* SUSNET_F_* in concatenation order */
    void *out;                    /* [B][obs_size]; PLANES: spatial [B][A+2][N][N]; PERSP: [B][A][A+2][N][N] */
    void *out2;                   /* PLANES: non_spatial [B][A(+A)+J]; PERSP: [B][A][A(+A)+J]; else NULL */
} susnet_obs_spec;

typedef struct susnet_step_io {
    const void *actions;  /* role-relative action indices (base.py:379-382) */
    int32_t actions_dtype;  /* SUSNET_U8 / I32 / I64 */
    int32_t actions_layout; /* SUSNET_LAYOUT_* */
    void *rewards;        /* out, one per agent */
    int32_t rewards_dtype;  /* SUSNET_F32 / F64 (reference: float64, base.py:369) */
    int32_t rewards_layout;
    uint8_t *done;        /* out [B]  (base.py:384-385) */
    uint8_t *truncated;   /* out [B]  (base.py:392-395) */
    const susnet_obs_spec *obs; /* optional fused observation of the post-step (post-auto-reset) state */
    /* replay feed of ONE tick (what susnet_ring_append needs besides actions / rewards / done / truncated / the raw uint8 observation
     * when the trajectory is collected tick by tick -- the trainer's loop, train.py:345-399 -- instead of by a fused rollout); both
     * optional: */
    uint8_t *term_obs;    /* out [B][obs_raw_size] u8, written ONLY where the episode ended at this step: its true terminal state
                           * (flatten_state order), before the auto-reset replaces it */
    uint16_t *roles;      /* out [B]: imposter bitmask (bit i = agent i) of the episode that acted at this step */
} susnet_step_io;

/* Fused random rollout: T lockstep ticks in one launch; per tick every env samples uniform role-valid
 * actions (sample_actions), steps, and auto-resets on done|truncated.  Every trajectory pointer is
 * optional (NULL = not stored).  PHILOX handles: any subset of outputs.  TAPE handles (numpy parity): the full trajectory
 * with the raw uint8 observation, as separate tensors or as packed records. */
typedef struct susnet_rollout_io {
    int32_t n_ticks;
    uint8_t *actions;   /* out [T][B][A] u8  (env-major: what reference callers index as actions[b]) */
    float *rewards;     /* out [T][B][A] f32 */
    uint8_t *done;      /* out [T][B] */
    uint8_t *truncated; /* out [T][B] */
    const susnet_obs_spec *obs; /* out pointer is [T][B][obs_size]; observation AFTER each tick */
    void *record;       /* alternative to ALL of the above (which must then be NULL): one packed record per env-step,
                         * [T][B][record_bytes] -- the same fields, laid out for one wide store per lane
                         * (susnet_record_layout); only for the compiled-in configurations */
    /* replay feed (susnet_ring_append); both optional, both need obs = RAW / U8 next to the full trajectory (term_obs also goes with
     * `record` on handles whose records are whole -- susnet_record_layout_t.planar == 0) */
    uint8_t *term_obs;  /* out [T][B][obs_raw_size] u8, written ONLY at (tick, env) where the episode ended: the true
                         * post-step state (the obs row of such a tick already holds the next episode's first state) */
    uint16_t *roles;    /* out [T][B]: imposter bitmask (bit i = agent i) of the episode that acted at the tick */
    int32_t record_format; /* which record `record` receives: SUSNET_RECORD_DEFAULT, or SUSNET_RECORD_COMPACT where the handle has one
                            * (susnet_record_layout_of) */
} susnet_rollout_io;
#define SUSNET_RECORD_DEFAULT 0
/* the 1v1 game on a grid without walls (BASELINE configs[1]): 16 bytes -- rewards f32[2] | raw observation u8[6] | ONE byte holding
 * actions and flags (susnet_record_layout_t.flags_packed) | 0 -- so that an env-step is one 16-byte store and a wave writes 1 KiB of
 * consecutive bytes per tick (+8 % env-steps/s over the 20-byte default, whose bytes stay as they are) */
#define SUSNET_RECORD_COMPACT 1

/* Device replay ring fed from a fused rollout: the reference's ReplayBuffer tensors (src/replay_memory.py:33-44) written by
 * ONE launch from the [T][B] trajectory of susnet_rollout, with ReplayBuffer.populate's semantics (replay_memory.py:96-143):
 * row n = tick * B + env (the order B sequential `add` calls per tick would produce) lands at ring position (idx + n) %
 * max_size; states[row] is the env's window of its last `trajectory_size` flattened states before the tick (after a reset:
 * the episode's first state repeated, replay_memory.py:108-113), next_states[row] the window rolled by one with the TRUE
 * post-step state appended (np.roll, replay_memory.py:120-127 -- also for a transition that ends its episode: term_obs),
 * dones[row] = done (not truncated), imposters[row] = ascending agent indices of the acting episode's imposters.  If the
 * launch holds more than max_size transitions only the last max_size are written.  `window` carries every env's window
 * across launches: in = before the first tick, out = before the tick that follows the last one. */
typedef struct susnet_ring_io {
    int32_t n_ticks;            /* T of the rollout */
    int32_t trajectory_size;    /* window length (ReplayBuffer.trajectory_size) */
    const uint8_t *actions;     /* [T][B][A] u8 */
    const float *rewards;       /* [T][B][A] */
    const uint8_t *done;        /* [T][B] */
    const uint8_t *truncated;   /* [T][B] */
    const uint8_t *obs;         /* [T][B][S] raw u8: state after each tick (after the in-launch reset, if any) */
    const uint8_t *term_obs;    /* [T][B][S] raw u8, read where done | truncated */
    const uint16_t *roles;      /* [T][B], or NULL: the imposters are agents [0, n_imposters) */
    uint8_t *window;            /* in/out [B][trajectory_size][S] u8 */
    int64_t max_size, idx;      /* ring capacity (rows), position of the next row */
    float *states;              /* [max_size][trajectory_size][S] */
    float *next_states;         /* [max_size][trajectory_size][S] */
    int64_t *ring_actions;      /* [max_size][A] */
    float *ring_rewards;        /* [max_size][A] */
    uint8_t *ring_dones;        /* [max_size][1] */
    int16_t *ring_imposters;    /* [max_size][n_imposters] */
    /* alternative to actions / rewards / done / truncated / obs (which must then be NULL): the [T][B] packed records a fused rollout
     * wrote (susnet_rollout_io.record, format record_format), read in place -- for handles that store whole records
     * (susnet_record_layout_t.planar == 0: the 1v1 kernels); term_obs as above */
    const uint8_t *record;
    int32_t record_format;
} susnet_ring_io;
int susnet_ring_append(susnet_env *env, const susnet_ring_io *io, void *stream);

/* Packed trajectory record of susnet_rollout_io.record: rewards f32[A], actions u8[A], done u8, truncated u8 and the raw
 * observation u8[obs_raw_size] (flatten_state order) of one env-step, zero-padded to a multiple of 4 bytes.  The ORDER of the
 * fields depends on the kernel that serves the handle (the multi-agent kernels store the observation right behind the actions
 * and done / truncated behind it, so that its words go out unshifted): read the byte offsets below, do not assume them.
 * record_bytes = 0 when the handle's configuration has no packed mode (not one of the compiled-in games). */
typedef struct susnet_record_layout_t {
    int32_t record_bytes;
    int32_t off_rewards, off_actions, off_done, off_truncated, off_obs;
    /* planar = 0: the record array is [T][B][record_bytes].
     * planar = 1 (the multi-agent kernels): the same record, cut into pieces of 16 bytes -- the last one(s) 8 and / or 4 bytes when
     * record_bytes is not a multiple of 16 -- and stored piece by piece, [T][piece][B][piece width]: every store instruction of a
     * wave then writes 1 KiB of consecutive bytes (40 partially written cache lines at a 40-byte record stride otherwise).  Byte o
     * of env b's record of tick t lives at
     *     t * B * record_bytes + B * P(o) + b * W(o) + (o - P(o)),
     * P(o) = the start of o's piece, W(o) its width: with full = record_bytes / 16 * 16, P(o) = o / 16 * 16, W = 16 for o < full;
     * then an 8-byte piece when record_bytes - full >= 8, then a 4-byte piece when record_bytes - full is 4 or 12. */
    int32_t planar;
    /* The raw observation (flatten_state order, obs_raw_size bytes) as the concatenation of n_obs_segments byte ranges of the record.
     * One segment (off_obs, obs_raw_size) for the fully compiled-in games; the multi-agent FAMILY kernels (any job count up to 8)
     * keep a record layout that does not depend on the job count -- eight job slots -- and report up to four segments: cells + alive,
     * job cells, job status, the tagging tail.  off_obs = the first segment's offset. */
    int32_t n_obs_segments;
    struct { int32_t off, len; } obs_segments[4];
    /* flags_packed = 1 (SUSNET_RECORD_COMPACT, two agents): off_actions = off_done = off_truncated = ONE byte
     * a0 | a1 << 3 | done << 6 | truncated << 7 (role-relative action indices below 8). */
    int32_t flags_packed;
} susnet_record_layout_t;
int susnet_record_layout(const susnet_env *env, susnet_record_layout_t *out);        /* the default record */
int susnet_record_layout_of(const susnet_env *env, int32_t record_format, susnet_record_layout_t *out); /* record_bytes = 0: the handle has no such record */

/* The trajectory modes of susnet_rollout address every output array with 32-bit offsets: one launch covers at most as many ticks
 * as keep each array below `bytes` (default and maximum 2^31 - 1); longer requests run as consecutive launches.  bytes = 0
 * restores the default.  (No reference counterpart: launch plumbing.) */
int susnet_set_launch_limit(susnet_env *env, uint64_t bytes);

/* Reference-layout views of the state (export / import). NULL pointers are skipped. */
typedef struct susnet_state_view {
    int32_t *agent_positions; /* [B][A][2] (x, y)            base.py:291 */
    uint8_t *alive_agents;    /* [B][A]                      base.py:301 */
    uint8_t *imposter_mask;   /* [B][A]                      base.py:280-281 */
    int32_t *job_positions;   /* [B][J][2]                   base.py:299 */
    uint8_t *completed_jobs;  /* [B][J]                      base.py:302 */
    uint8_t *used_tag_actions;/* [B][A]                      tagging.py:30 */
    int32_t *tag_counts;      /* [B][A]                      tagging.py:29 */
    int32_t *tag_reset_timer; /* [B]                         tagging.py:31 */
    int32_t *t;               /* [B]                         base.py:315 */
    int64_t *metrics;         /* [B][13] info counters       metrics.py:35-64 */
    uint64_t *rng_cursor;     /* [B] words consumed so far (PHILOX: the event stream = the kill draws; TAPE: every draw) */
    uint32_t *lifetime;       /* [SUSNET_N_LIFETIME][B] per-env sums over finished episodes */
    uint32_t *episode_index;  /* [B] resets drawn so far = the index of the env's next reset in the RESET stream (PHILOX handles:
                                 a reset's draws are a function of (seed, global env id, this index) alone) */
} susnet_state_view;

int susnet_abi_version(void);
const char *susnet_last_error(void);

int susnet_create(const susnet_config *cfg, susnet_env **out);
void susnet_destroy(susnet_env *env);
int susnet_get_layout(const susnet_env *env, susnet_layout *out);

/* Bind the caller-allocated state blob (zero-filled by the callee, asynchronously, on `stream`). */
int susnet_bind_state(susnet_env *env, void *state_blob, uint64_t bytes, void *stream);
/* TAPE mode: per-env raw 32-bit words, tape[b][0..words_per_env); cursors restart at 0. */
int susnet_bind_tape(susnet_env *env, const uint32_t *tape, int64_t words_per_env);
/* PHILOX mode: reseed / reposition every env's stream (also rewinds the handle's tick counter to 0). */
int susnet_seed(susnet_env *env, uint64_t seed, uint64_t cursor, void *stream);
/* PHILOX mode draws agent actions from a second stream indexed by the number of steps the handle has taken
 * ("tick"; advanced by susnet_step / susnet_rollout, NOT by susnet_sample_actions -- sampling twice before a
 * step returns the same actions).  Set and/or read it (either pointer may be NULL).  With the counter in device memory
 * (susnet_device_tick) both are ordered on `stream`; reading synchronises `stream`. */
int susnet_tick(susnet_env *env, const uint64_t *set, uint64_t *get, void *stream);

int susnet_reset(susnet_env *env, const uint8_t *mask /* [B] or NULL = all */, const susnet_obs_spec *obs,
                 void *stream);
int susnet_sample_actions(susnet_env *env, void *actions_out, int32_t dtype, int32_t layout, void *stream);

/* Greedy policy actions of one tick -- the acting step of the reference's policy loops (run_game, src/visualize.py:547-562:
 * every agent takes the argmax of ITS TEAM's Q-network; train.py:355-381 with epsilon = 0):
 *   actions[b][i] = argmax_k q_imposter[b][k]   if agent i is an imposter of environment b
 *                 = argmax_k q_crew[b][k]       otherwise
 * q_imposter / q_crew: float32 [B][n_actions_imposter] / [B][n_actions_crew], one row per environment (all agents of an
 * environment share the flat observation, FlatFeaturizer model_ready.py:356-367); ties go to the lowest index, like
 * torch.argmax / np.argmax.  q_crew = NULL: the crew draws uniformly random role-valid indices from the production action
 * stream -- exactly what susnet_sample_actions would return for them (PHILOX handles only).  One launch instead of the
 * role export + argmax + sample + dtype copy + where of the eager loop; actions_out as for susnet_sample_actions. */
typedef struct susnet_policy_opts {
    float epsilon;     /* epsilon-greedy, train.py:355-381: with this probability an agent takes its uniformly random role-valid draw (what
                        * susnet_sample_actions returns for it) instead of the argmax; the decision is word tick * A + i of a third
                        * stream of the handle's Philox key (PHILOX handles only); 0 = greedy */
    int32_t mask_dead; /* != 0: dead agents get index 0 (train.py sets only the living agents' actions); 0: they act like everybody
                        * else (run_game, visualize.py:547-560) -- the step ignores a dead agent's action either way */
    /* ABI 6 -- the CREW's Q-network inside the one-kernel tick (susnet_qnet_policy_step / susnet_qnet_policy_rollout only; the other
     * calls take the crew's Q rows as an argument and refuse these fields): run_game drives both teams by their networks
     * (visualize.py:547-562), train_crew exists (notebooks/experiment.ipynb cell 5).  crew_packed: a susnet_qnet_pack image for the SAME
     * components as the imposters' network (device, 16-byte aligned), crew_dims / crew_n_dims its layer widths (last = the crew's action
     * count), crew_q_out: [B][n] float32 (rollout: [T][B][n]) or NULL.  NULL crew_packed = a uniformly random crew.  With both networks and
     * epsilon = 0 nothing is drawn from the action stream, so numpy-tape (TAPE) handles are served as well. */
    const float *crew_packed;
    const int32_t *crew_dims;
    int32_t crew_n_dims, pad_;
    float *crew_q_out;
} susnet_policy_opts;
int susnet_policy_actions(susnet_env *env, const float *q_imposter, const float *q_crew, const susnet_policy_opts *opts /* or NULL */,
                          void *actions_out, int32_t dtype, int32_t layout, void *stream);

/* The policy loop's Q-network -- the reference's MLP (src/models/dqn.py:72-108: make_mlp 322-329, Linear + nn.PReLU() with ONE
 * slope per layer, no activation after the last Linear) on the FlatFeaturizer observation of the handle's CURRENT environments
 * (model_ready.py:356-367) -- as one kernel: MLP.forward(spatial, env.obs) without the observation or any activation ever
 * being written to memory (float32 throughout; layers 2.. on the f32-input matrix instructions, whose result is a k-ordered fmaf
 * chain, so values agree with torch's to float32 summation-order differences).  Served: five Linear layers
 * dims = [F, <=256, <=128, <=64, <=32, <=32] (the reference's [F, 256, 128, 64, 16, n_actions], notebooks/experiment_1v1.ipynb cell 1) on
 * the compiled-in feature layouts: components {ONEHOT_POS} (F = 36: experiments no_wall / one_hot_wall of notebooks/experiment_1v1.ipynb)
 * or {COORD_POS} (F = 4, CoordinateAgentPositionsFeaturizer, component.py:384-403: no_wall_coord_features / wall_coord_features) on the
 * 2-agent 9x9 game, either map, and {ONEHOT_POS, ALIVE_CREW, CLOSEST_CREW} on the 3-agent 14x14 game (F = 88).
 *   susnet_qnet_packed_floats  size of the packed weight image (floats), or SUSNET_E_INVALID when the handle / components / dims are
 *                              not served (callers then run the network themselves and hand Q rows to susnet_policy_actions);
 *   susnet_qnet_pack           HOST: torch-layout weights[l] ([dims[l+1]][dims[l]] row-major), biases[l] ([dims[l+1]]), slopes
 *                              ([n_dims - 2], the PReLU weights) -> packed (host buffer of that size; the caller copies it to the device:
 *                              the library never allocates device memory);
 *   susnet_qnet_forward        q_out [B][dims[n_dims-1]] float32 (device) from the packed image (device, 16-byte aligned). */
int64_t susnet_qnet_packed_floats(const susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims,
                                  int32_t n_dims);
int susnet_qnet_pack(const susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                     const float *const *weights, const float *const *biases, const float *slopes, float *packed);
int susnet_qnet_forward(susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                        const float *packed, float *q_out, void *stream);
int susnet_step(susnet_env *env, const susnet_step_io *io, void *stream);
/* susnet_policy_actions and susnet_step in ONE launch -- a tick of the acting loop (visualize.py:547-582: argmax per team, env.step):
 * the stepping lane takes the teams' greedy actions from the Q rows itself (q_imposter / q_crew as for susnet_policy_actions; at most
 * 16 actions per team) and steps with them; io->actions is an OUTPUT here (the actions taken, same dtypes / layouts; NULL: not kept),
 * every other field of io as for susnet_step. */
int susnet_policy_step(susnet_env *env, const float *q_imposter, const float *q_crew, const susnet_policy_opts *opts /* or NULL */,
                       const susnet_step_io *io, void *stream);
/* susnet_qnet_forward and susnet_policy_step (imposters by the network; the crew random, or by ITS network: opts->crew_packed) as ONE
 * kernel -- a whole tick of the acting loop in one launch: the wave that computed its 64 environments' Q rows takes their argmax in
 * registers and steps them (with a crew network the workgroup swaps the LDS image between the two passes).  Arguments as for
 * the two calls; q_out may be NULL (Q rows not kept).  Served: the two compiled-in games the network kernel knows the feature layout of
 * (2-agent 9x9 ImposterTrainingGround; 1v2 14x14 FourRoomEnv with 4 jobs), PHILOX handles; otherwise SUSNET_E_INVALID and the caller
 * uses the two calls. */
int susnet_qnet_policy_step(susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                            const float *packed, float *q_out, const susnet_policy_opts *opts /* or NULL */, const susnet_step_io *io, void *stream);

/* The policy tick (susnet_qnet_policy_step) repeated n_ticks times inside ONE launch: the acting loop of the trainer over a block of
 * ticks with fixed weights (train.py:345-399 between two optimizer steps) -- the network image is loaded once, a launch is paid once,
 * and what replay_buffer.add needs of every tick lands in slot t of [T][B] arrays, ready for susnet_ring_append.  Every pointer is
 * optional.  Served where susnet_qnet_policy_step is (the two compiled-in games, PHILOX handles, a random crew); the handle must
 * auto-reset.  Advances the handle by n_ticks steps. */
typedef struct susnet_feed_io {
    uint8_t *actions;    /* out [T][B][A] u8: the actions taken */
    float *rewards;      /* out [T][B][A] f32 */
    uint8_t *done;       /* out [T][B] */
    uint8_t *truncated;  /* out [T][B] */
    uint8_t *obs;        /* out [T][B][obs_raw_size] u8: the state after each tick (after the auto-reset where the episode ended); 16-byte aligned
                          * slots (n_ticks > 1: B * obs_raw_size a multiple of 16, else SUSNET_E_INVALID) */
    uint8_t *term_obs;   /* out [T][B][obs_raw_size] u8, written only where an episode ended: its terminal state */
    uint16_t *roles;     /* out [T][B]: imposter bitmask of the episode that acted */
    float *q;            /* out [T][B][n_actions_imposter] f32: the imposters' Q rows, or NULL */
} susnet_feed_io;
int susnet_qnet_policy_rollout(susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                               const float *packed, const susnet_policy_opts *opts, const susnet_feed_io *feed, int32_t n_ticks, void *stream);

int susnet_rollout(susnet_env *env, const susnet_rollout_io *io, void *stream);
int susnet_observe(susnet_env *env, const susnet_obs_spec *obs, void *stream);
int susnet_obs_size(const susnet_env *env, const susnet_obs_spec *obs, int32_t *size_out, int32_t *size2_out);

/* Observation of n_rows FLATTENED states instead of the handle's own environments: what the reference's
 * sequence featurizers do with a state window or a replay batch -- SequenceStateFeaturizer.fit(state_sequence
 * [B, T, S]) un-flattens every state and featurises it (src/features/model_ready.py:41-57; FlatFeaturizer.fit
 * 338-354, GlobalFeaturizer.fit 254-289; the trainer calls it on windows train.py:345-347 and on replay batches).
 * rows: device pointer, [n_rows][S] contiguous, S = susnet_layout.obs_raw_size (flatten_state order, base.py:234:
 * x0,y0,.. alive.. jobxy.. jobdone.. (, used, counts, timer_left)), rows_dtype one of SUSNET_U8 / I32 / I64 / F32 /
 * F64 (integer-valued).  obs->out (/out2) receive [n_rows][obs_size] in the layouts of susnet_observe; obs->mode
 * FLAT or PLANES.  A row with a coordinate outside the grid is written as all zeros and reported by
 * susnet_poll_errors (SUSNET_E_ROW).  The handle supplies the configuration (A, J, N, wall map) only; its
 * environments are not touched. */
int susnet_featurize(susnet_env *env, const void *rows, int32_t rows_dtype, int64_t n_rows, const susnet_obs_spec *obs,
                     void *stream);

/* ImposterScentFeaturizer (src/features/component.py:336-380; the one real-valued feature of the reference) of n_rows
 * FLATTENED states (layout and dtypes as susnet_featurize): out [n_rows][4] float32, 16-byte aligned. */
int susnet_scent(susnet_env *env, const void *rows, int32_t rows_dtype, int64_t n_rows, float *out, void *stream);

int susnet_export_state(susnet_env *env, const susnet_state_view *view, void *stream);
int susnet_import_state(susnet_env *env, const susnet_state_view *view, void *stream);

/* Sum the per-env lifetime accumulators into out[SUSNET_N_LIFETIME] (device int64 buffer): the vector
 * the multi-GPU host all-gathers.  One small launch. */
int susnet_reduce_lifetime(susnet_env *env, int64_t *out_device, void *stream);

/* Graph-replayable launches.  By default the step counter that indexes the production action stream (susnet_tick)
 * travels as a kernel argument, so a captured launch would replay with a stale value.  enable = 1 moves the counter
 * into device memory (inside the bound state blob), ONE COPY PER ENVIRONMENT: sample_actions / step / rollout read the
 * env's own copy and step / rollout advance it in the same kernel (the lane that owns the env; no extra launch, no word
 * one workgroup writes while another reads), so a hipGraph captured from these calls (e.g. {susnet_sample_actions;
 * susnet_step}) can be replayed any number of times.  enable = 0 reads it back (synchronises `stream`).  susnet_tick()
 * keeps working in both modes. */
int susnet_device_tick(susnet_env *env, int32_t enable, void *stream);

/* Synchronises `stream`, reads and clears the device error word. Returns 0 or the most severe
 * SUSNET_E_ACTION_* / SUSNET_E_TAPE / SUSNET_E_ROW code; *bits_out receives the raw bits. */
int susnet_poll_errors(susnet_env *env, uint32_t *bits_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SUSNET_H */
