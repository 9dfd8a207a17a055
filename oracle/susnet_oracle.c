/*
 * susnet_oracle.c -- plain-C restatement of the reference env hot path (see susnet_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: checker for tests/, smoke() and bench.py's cpu_baseline leg.
 * Parity pin: tests/golden/ *.npz (generated from the unmodified reference; see generate_golden.py).
 *
 * Third-party algorithm restated here because the reference's randomness lives in it:
 *   numpy legacy RandomState (numpy is un-pinned in /root/reference/environment.yml:7; the legacy stream
 *   is frozen by numpy's compatibility policy, NEP 19).  Restated from the published algorithm:
 *   MT19937 (Matsumoto & Nishimura 1998) with Knuth seeding; bounded integers by masked rejection on
 *   32-bit words; Fisher-Yates from the top index down.  Anchored on the reference's call sites
 *   base.py:126,267,274-276,288-290,295-297,329,374,497 and tagging.py:167 and pinned by the `words`
 *   column of every golden trace (cumulative raw words numpy consumed).
 */
#include "susnet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * word sources
 * ---------------------------------------------------------------------------------------------- */
static void mt_seed(so_rng *r, uint32_t seed) {
    /* numpy legacy seeding for an int seed == init_genrand (Knuth LCG 1812433253) */
    for (int i = 0; i < 624; i++) {
        r->mt[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    r->mti = 624;
}

static void mt_twist(so_rng *r) {
    uint32_t *mt = r->mt;
    int kk;
    uint32_t y;
    for (kk = 0; kk < 624 - 397; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < 623; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r->mti = 0;
}

static uint32_t mt_next(so_rng *r) {
    if (r->mti >= 624) mt_twist(r);
    uint32_t y = r->mt[r->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* Philox4x32-10 (Salmon et al., SC'11).  The product's production stream:
 *   key = (seed_lo, seed_hi); counter = (block_lo, block_hi, env_lo, env_hi); block = cursor >> 2;
 *   word = out[cursor & 3].  The HIP kernels implement the identical mapping (csrc/susnet_rng.h). */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

void so_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; i++) out[i] = c[i];
}

/* Streams that share the key and the env id: the per-env EVENT stream (step-internal draws: the kill draws; tag 0), the
 * ACTION stream (tag bit 31 of counter word 1), whose word index is tick * A + agent with `tick` the number of
 * steps taken -- identical for all envs stepped in lockstep, so a wave generates action blocks under uniform
 * control flow and every generated word is used -- and the RESET stream (tag bit 29): word j of the env's n-th reset
 * (n = resets drawn since seeding) is word j & 3 of the block with counter (n, tag | j >> 2, env).  A reset's draws are
 * thereby a function of (seed, env, n) alone -- not of how many words the episodes before it consumed -- which is
 * what lets the fused rollouts draw an environment's NEXT episode ahead of time, all lanes of a wave at once. */
#define SO_ACTION_STREAM_TAG 0x80000000u
#define SO_RESET_STREAM_TAG 0x20000000u
static uint32_t philox_reset_word(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t j) {
    uint32_t c[4] = {episode, SO_RESET_STREAM_TAG | (j >> 2), (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return c[j & 3];
}
static uint32_t philox_word_tagged(uint64_t seed, uint64_t env_id, uint64_t cursor, uint32_t tag) {
    uint64_t block = cursor >> 2;
    uint32_t c[4] = {(uint32_t)block, (uint32_t)(block >> 32) | tag, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return c[cursor & 3];
}

static uint32_t philox_word(uint64_t seed, uint64_t env_id, uint64_t cursor) {
    uint64_t block = cursor >> 2;
    uint32_t c[4] = {(uint32_t)block, (uint32_t)(block >> 32), (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return c[cursor & 3];
}

uint32_t so_next_u32(so_env *e) {
    so_rng *r = &e->rng;
    uint32_t w;
    switch (r->kind) {
    case SO_RNG_TAPE:
        if ((int64_t)r->cursor >= r->tape_len) {
            r->overflow = 1;
            w = 0;
        } else {
            w = r->tape[r->cursor];
        }
        break;
    case SO_RNG_PHILOX:
        if (r->in_reset) return philox_reset_word(r->seed, r->env_id, r->episode, r->reset_pos++); /* (the event cursor stays) */
        w = philox_word(r->seed, r->env_id, r->cursor);
        break;
    default:
        w = mt_next(r);
        break;
    }
    r->cursor++;
    return w;
}

void so_seed_mt(so_env *e, uint32_t seed) {
    e->rng.kind = SO_RNG_MT19937;
    mt_seed(&e->rng, seed);
    e->rng.cursor = 0;
    e->rng.overflow = 0;
}

void so_set_tape(so_env *e, const uint32_t *words, int64_t n_words) {
    e->rng.kind = SO_RNG_TAPE;
    e->rng.tape = words;
    e->rng.tape_len = n_words;
    e->rng.cursor = 0;
    e->rng.overflow = 0;
}

void so_set_philox(so_env *e, uint64_t seed, uint64_t env_id, uint64_t cursor) {
    e->rng.kind = SO_RNG_PHILOX;
    e->rng.seed = seed;
    e->rng.env_id = env_id;
    e->rng.cursor = cursor;
    e->rng.overflow = 0;
    e->rng.tick = 0;
    e->rng.episode = 0;
    e->rng.reset_pos = 0;
    e->rng.in_reset = 0;
}
void so_set_episode(so_env *e, uint32_t episode) { e->rng.episode = episode; }

void so_set_tick(so_env *e, uint64_t tick) { e->rng.tick = tick; }

/* numpy legacy bounded draw on [0, max]: smallest all-ones mask >= max, reject until <= max.
 * max == 0 consumes nothing (observed: choice([x]) / shuffle of one item draw no word). */
static uint32_t rk_interval(so_env *e, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
        v = so_next_u32(e) & mask;
    } while (v > max);
    return v;
}

/* np.random.randint(0, n) / np.random.choice(n) one element */
static int np_randint(so_env *e, int n) { return (int)rk_interval(e, (uint32_t)(n - 1)); }

/* ---- draw protocol ---------------------------------------------------------------------------------
 * MT19937 / TAPE : numpy-legacy semantics (reference behaviour).
 * PHILOX         : the PRODUCT's production protocol, restated here only so that the HIP kernels can be
 *                  checked bit for bit in that mode too (it is not reference behaviour; it draws from the
 *                  same distributions): the event cursor is aligned to a 4-word Philox block at the start of
 *                  a step, every bounded draw consumes exactly one word (also for n == 1) and maps it with a
 *                  multiply-shift, "without replacement" is sequential rejection of duplicates instead of a
 *                  full permutation, and a reset draws from the RESET stream (above), word 0 onwards. */
static int draw_bounded(so_env *e, int n) {
    if (e->rng.kind == SO_RNG_PHILOX) return (int)(((uint64_t)so_next_u32(e) * (uint64_t)(uint32_t)n) >> 32);
    return np_randint(e, n);
}
static void draw_align(so_env *e) {
    if (e->rng.kind == SO_RNG_PHILOX) e->rng.cursor = (e->rng.cursor + 3ull) & ~3ull;
}

/* np.random.shuffle / permutation: for i = n-1 .. 1: j = interval(i); swap(x[i], x[j]) */
static void np_shuffle(so_env *e, int32_t *x, int n) {
    for (int i = n - 1; i >= 1; i--) {
        int j = draw_bounded(e, i + 1);
        int32_t t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
}

/* ------------------------------------------------------------------------------------------------
 * construction  (base.py:103-228, pred_prey.py:26-76, tagging.py:10-60)
 * ---------------------------------------------------------------------------------------------- */
int so_sizeof_env(void) { return (int)sizeof(so_env); }

/* PRODUCT protocol (not reference behaviour), restated so that the kernels can be checked bit for bit on the Philox
 * stream: how a tick's bounded draws -- A action draws in agent order (base.py:326-330), then, with a shuffled action
 * order (np.random.shuffle, base.py:372-374), the A - 1 placement draws k = 1 .. A-1 (range k + 1, see so_step) -- are
 * packed into 32-bit words of the ACTION stream.  Consecutive draws share a word by nested multiply-shift (digit = hi32(w * n), w = lo32(w * n):
 * the mixed-radix digits of w * n1 * n2 .. / 2^32, joint bias <= n1 * n2 .. * 2^-32); a word is closed as soon as the
 * product of the draws' LARGEST possible ranges (imposter action count; i + 1 for a shuffle draw) would pass 2^16, so
 * the packing is static and the bias of any word stays below 2^-16.  A tick owns aw_W consecutive words (tick t: words
 * t * aw_W ..), NOT rounded up to Philox blocks.  The 1v1 game (A == 2: ranges 6 and 5 whichever agent is the
 * imposter, product 30) packs THREE ticks into a word (30^3 = 27000): tick t uses word t / 3 after t % 3 earlier ticks'
 * worth of digits have been taken, i.e. starts from lo32(word * 30^(t % 3)). */
#define SO_AW_CAP 65536u
static int role_action_count(const so_env *e, int is_imp);
static void so_action_layout(so_env *e) {
    const int A = e->A;
    const uint32_t R = (uint32_t)role_action_count(e, 1) + (e->cfg.variant == SO_VARIANT_TAGGING ? (uint32_t)(A - 1) : 0u);
    int n = 0, word = 0;
    uint32_t prod = 1;
    for (int i = 0; i < A; i++) {
        if (prod * R > SO_AW_CAP) { word++; prod = 1; }
        prod *= R;
        e->aw_word[n++] = (uint8_t)word;
    }
    if (e->cfg.is_action_order_random)
        for (int k = 1; k < A; k++) {
            uint32_t radix = (uint32_t)k + 1u;
            if (prod * radix > SO_AW_CAP) { word++; prod = 1; }
            prod *= radix;
            e->aw_word[n++] = (uint8_t)word;
        }
    e->aw_W = word + 1;
    e->aw_tpw = 1;
    if (A == 2) { e->aw_W = 1; e->aw_tpw = 3; }
}

int so_env_init(so_env *e, const so_config *cfg) {
    memset(e, 0, sizeof(*e));
    e->cfg = *cfg;
    if (cfg->grid_n < 1 || cfg->grid_n > SO_MAX_GRID) return SO_ERR_CONFIG;
    if (cfg->variant == SO_VARIANT_ITG) {
        /* pred_prey.py:52-66: exactly one imposter, dead_penalty 0, fixed order; validation relaxed (75-76) */
        e->cfg.n_imposters = 1;
        e->cfg.dead_penalty = 0.0;
        e->cfg.is_action_order_random = 0;
        if (cfg->n_crew <= 0) return SO_ERR_CONFIG;
    } else {
        /* base.py:243-249 */
        if (cfg->n_imposters <= 0 || cfg->n_crew <= 0 || cfg->n_jobs < 0) return SO_ERR_CONFIG;
        if (!(cfg->n_imposters < cfg->n_crew)) return SO_ERR_CONFIG;
    }
    e->A = e->cfg.n_imposters + e->cfg.n_crew;
    e->J = e->cfg.n_jobs;
    if (e->A > SO_MAX_AGENTS || e->J > SO_MAX_JOBS) return SO_ERR_CONFIG;
    /* valid_positions = np.argwhere(grid): row-major (i, j) with grid[i][j] true (base.py:199);
     * a spawned agent at row (i, j) has x = i, y = j (base.py:291) */
    int n = cfg->grid_n;
    e->n_valid = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (cfg->grid[i][j]) {
                e->valid[e->n_valid][0] = (uint8_t)i;
                e->valid[e->n_valid][1] = (uint8_t)j;
                e->n_valid++;
            }
    if (e->n_valid < e->J || e->n_valid < 1) return SO_ERR_CONFIG;
    so_action_layout(e);
    so_seed_mt(e, 0);
    return SO_OK;
}

static int role_action_count(const so_env *e, int is_imp) {
    /* base.py:82-99 (crew 6, imposter 7); pred_prey.py:4-19 (crew 5, imposter 6) */
    if (e->cfg.variant == SO_VARIANT_ITG) return is_imp ? 6 : 5;
    return is_imp ? 7 : 6;
}

int so_n_actions(const so_env *e, int agent) {
    int n = e->n_role_actions[agent];
    if (e->cfg.variant == SO_VARIANT_TAGGING) n += e->A - 1; /* tagging.py:68-75 */
    return n;
}

/* role-relative index -> Action enum value (or -1 if out of the role list) */
static int role_action(const so_env *e, int is_imp, int idx) {
    if (idx < 0) return -1;
    if (idx <= 4) return idx; /* STAY, UP, DOWN, LEFT, RIGHT */
    if (e->cfg.variant == SO_VARIANT_ITG) {
        if (is_imp && idx == 5) return SO_KILL; /* pred_prey.py:12-19 */
        return -1;
    }
    if (is_imp) {
        if (idx == 5) return SO_SABOTAGE; /* base.py:91-99 */
        if (idx == 6) return SO_KILL;
        return -1;
    }
    if (idx == 5) return SO_FIX; /* base.py:82-89 */
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * reset  (base.py:251-324; tagging.py:62-101)
 * ---------------------------------------------------------------------------------------------- */
void so_reset(so_env *e) {
    const int A = e->A, J = e->J;
    memset(e->metrics, 0, sizeof(e->metrics)); /* base.py:270 */

    const int philox = e->rng.kind == SO_RNG_PHILOX;
    if (philox) { /* production protocol: this reset's own words of the RESET stream */
        e->rng.in_reset = 1;
        e->rng.reset_pos = 0;
    }
    /* base.py:273-278 */
    if (e->cfg.shuffle_imposter_index && philox) {
        for (int i = 0; i < A; i++) e->imp_mask[i] = 0;
        for (int k = 0; k < e->cfg.n_imposters; k++) {
            int pick;
            do { pick = draw_bounded(e, A); } while (e->imp_mask[pick]);
            e->imp_mask[pick] = 1;
            e->imp_idxs[k] = pick;
        }
    } else if (e->cfg.shuffle_imposter_index) {
        int32_t perm[SO_MAX_AGENTS];
        for (int i = 0; i < A; i++) perm[i] = i;
        np_shuffle(e, perm, A); /* choice(range(A), n_imp, replace=False) == permutation(A)[:n_imp] */
        for (int k = 0; k < e->cfg.n_imposters; k++) e->imp_idxs[k] = perm[k];
    } else {
        for (int k = 0; k < e->cfg.n_imposters; k++) e->imp_idxs[k] = k;
    }
    for (int i = 0; i < A; i++) e->imp_mask[i] = 0;
    for (int k = 0; k < e->cfg.n_imposters; k++) e->imp_mask[e->imp_idxs[k]] = 1;

    /* base.py:288-291: agent cells, with replacement */
    for (int i = 0; i < A; i++) {
        int c = draw_bounded(e, e->n_valid);
        e->pos[i][0] = e->valid[c][0];
        e->pos[i][1] = e->valid[c][1];
    }
    /* base.py:295-299: job cells, without replacement == permutation(n_valid)[:J]; the full
     * permutation is drawn even when J == 0 */
    if (philox) {
        for (int j = 0; j < J; j++) {
            int c, dup;
            do {
                c = draw_bounded(e, e->n_valid);
                dup = 0;
                for (int k = 0; k < j; k++)
                    dup |= (e->jobpos[k][0] == e->valid[c][0] && e->jobpos[k][1] == e->valid[c][1]);
            } while (dup);
            e->jobpos[j][0] = e->valid[c][0];
            e->jobpos[j][1] = e->valid[c][1];
        }
    } else {
        int32_t perm[SO_MAX_GRID * SO_MAX_GRID];
        for (int i = 0; i < e->n_valid; i++) perm[i] = i;
        np_shuffle(e, perm, e->n_valid);
        for (int j = 0; j < J; j++) {
            e->jobpos[j][0] = e->valid[perm[j]][0];
            e->jobpos[j][1] = e->valid[perm[j]][1];
        }
    }
    if (philox) {
        e->rng.in_reset = 0;
        e->rng.episode += 1u;
    }
    for (int i = 0; i < A; i++) e->alive[i] = 1;     /* base.py:301 */
    for (int j = 0; j < J; j++) e->jobdone[j] = 0;   /* base.py:302 */
    for (int i = 0; i < A; i++) e->n_role_actions[i] = role_action_count(e, e->imp_mask[i]); /* 306-312 */
    e->t = 0;                                        /* base.py:315 */
    /* tagging.py:64-66 */
    for (int i = 0; i < A; i++) {
        e->used[i] = 0;
        e->counts[i] = 0;
    }
    e->timer = 0;
    for (int i = 0; i < A; i++) {
        e->order[i] = i;
        e->rewards[i] = 0.0;
    }
}

/* base.py:326-330: one randint(len(agent_action_map[i])) per agent in index order */
/* word `index` of the production ACTION stream */
static uint32_t action_word(const so_env *e, uint64_t index) {
    return philox_word_tagged(e->rng.seed, e->rng.env_id, index, SO_ACTION_STREAM_TAG);
}

void so_sample_actions(so_env *e, int32_t *actions) {
    if (e->rng.kind == SO_RNG_PHILOX) {
        /* production protocol (so_action_layout): the actions of step `tick` are digits of the tick's action-stream words;
         * sampling does not advance anything (the step does), so repeated calls before a step return the same actions */
        if (e->A == 2) {
            uint32_t w = action_word(e, e->rng.tick / 3u);
            for (uint64_t s = 0; s < e->rng.tick % 3u; s++) w *= 30u; /* digits of the earlier ticks in this word */
            uint64_t p = (uint64_t)w * (uint64_t)(uint32_t)so_n_actions(e, 0);
            actions[0] = (int)(p >> 32);
            actions[1] = (int)(((uint64_t)(uint32_t)p * (uint64_t)(uint32_t)so_n_actions(e, 1)) >> 32);
            return;
        }
        uint32_t w = 0;
        for (int i = 0; i < e->A; i++) {
            if (i == 0 || e->aw_word[i] != e->aw_word[i - 1])
                w = action_word(e, e->rng.tick * (uint64_t)e->aw_W + e->aw_word[i]);
            uint64_t p = (uint64_t)w * (uint64_t)(uint32_t)so_n_actions(e, i);
            actions[i] = (int)(p >> 32);
            w = (uint32_t)p;
        }
        return;
    }
    for (int i = 0; i < e->A; i++) actions[i] = draw_bounded(e, so_n_actions(e, i));
}

/* ------------------------------------------------------------------------------------------------
 * step
 * ---------------------------------------------------------------------------------------------- */
/* base.py:548-551 -- note the TRANSPOSED lookup grid[pos[1], pos[0]] */
static int is_valid_position(const so_env *e, int x, int y) {
    int n = e->cfg.grid_n;
    if (x < 0 || y < 0 || x >= n || y >= n) return 0;
    return e->cfg.grid[y][x] != 0;
}

/* base.py:544-546: first job on the cell, or -1 */
static int job_at(const so_env *e, int x, int y) {
    for (int j = 0; j < e->J; j++)
        if (e->jobpos[j][0] == x && e->jobpos[j][1] == y) return j;
    return -1;
}

/* base.py:462-533 */
static void agent_step(so_env *e, int idx, int action) {
    if (!e->alive[idx]) return; /* base.py:477 */
    int x = e->pos[idx][0], y = e->pos[idx][1];
    if (action <= SO_RIGHT) { /* is_move_action (base.py:60-62); move() base.py:69-79 */
        int nx = x, ny = y;
        if (action == SO_UP) ny = y + 1;
        else if (action == SO_DOWN) ny = y - 1;
        else if (action == SO_LEFT) nx = x - 1;
        else if (action == SO_RIGHT) nx = x + 1;
        if (is_valid_position(e, nx, ny)) { /* base.py:486-487 */
            e->pos[idx][0] = nx;
            e->pos[idx][1] = ny;
        }
    } else if (action == SO_KILL) {
        /* base.py:493, 535-542: alive crew on the same cell, ascending index */
        int cands[SO_MAX_AGENTS], nc = 0;
        for (int i = 0; i < e->A; i++)
            if (e->alive[i] && !e->imp_mask[i] && e->pos[i][0] == x && e->pos[i][1] == y) cands[nc++] = i;
        if (nc > 0) {
            int victim = cands[draw_bounded(e, nc)]; /* base.py:497; no word drawn when nc == 1 */
            e->metrics[SO_M_IMP_KILLED_CREW] += 1; /* base.py:508 */
            e->alive[victim] = 0;                  /* base.py:511 */
            e->rewards[victim] = e->cfg.kill_reward; /* base.py:514-515: ASSIGNED, not added */
            e->rewards[idx] = e->cfg.kill_reward;
        }
    } else if (action == SO_FIX) { /* base.py:518-524 */
        int j = job_at(e, x, y);
        if (j >= 0 && !e->jobdone[j]) {
            e->jobdone[j] = 1;
            e->metrics[SO_M_COMPLETED_JOBS] += 1;
            e->rewards[idx] = e->cfg.complete_job_reward;
        }
    } else if (action == SO_SABOTAGE) { /* base.py:527-533 */
        int j = job_at(e, x, y);
        if (j >= 0 && e->jobdone[j]) {
            e->jobdone[j] = 0;
            e->metrics[SO_M_SABOTAGED_JOBS] += 1;
            e->rewards[idx] = -1.0 * e->cfg.sabotage_reward;
        }
    }
}

/* base.py:409-460 / pred_prey.py:78-99 */
static int check_win(so_env *e, double *team_reward) {
    int alive_imp = 0, alive_all = 0, done_jobs = 0;
    for (int i = 0; i < e->A; i++) {
        alive_all += e->alive[i];
        if (e->imp_mask[i]) alive_imp += e->alive[i];
    }
    for (int j = 0; j < e->J; j++) done_jobs += e->jobdone[j];
    if (e->cfg.variant == SO_VARIANT_ITG) {
        if (e->J != 0 && done_jobs == e->J) { /* pred_prey.py:88-91 */
            e->metrics[SO_M_CREW_WON] = 1;
            *team_reward = e->cfg.game_end_reward;
            return 1;
        }
        if (alive_all - alive_imp == 0) { /* pred_prey.py:94-97 */
            e->metrics[SO_M_IMPOSTER_WON] = 1;
            *team_reward = -1.0 * e->cfg.game_end_reward;
            return 1;
        }
        *team_reward = 0.0;
        return 0;
    }
    if (alive_imp == 0 || done_jobs == e->J) { /* base.py:428-435 (true every step when J == 0) */
        e->metrics[SO_M_CREW_WON] = 1;
        *team_reward = e->cfg.game_end_reward;
        return 1;
    }
    if (alive_all - alive_imp <= alive_imp) { /* base.py:438-446 */
        e->metrics[SO_M_IMPOSTER_WON] = 1;
        *team_reward = -1.0 * e->cfg.game_end_reward;
        return 1;
    }
    *team_reward = 0.0;
    return 0;
}

/* base.py:553-563 */
static void merge_rewards(so_env *e, double team_reward) {
    for (int i = 0; i < e->A; i++) e->rewards[i] += team_reward;
    for (int i = 0; i < e->cfg.n_imposters && i < e->A; i++) e->rewards[i] *= -1.0; /* indices [:n_imp], NOT the mask */
    for (int i = 0; i < e->A; i++)
        if (!e->alive[i]) e->rewards[i] = e->cfg.dead_penalty;
}

int so_step(so_env *e, const int32_t *actions, double *rewards, int32_t *done_out, int32_t *trunc_out) {
    const int A = e->A;
    const int tagging = e->cfg.variant == SO_VARIANT_TAGGING;
    /* base.py:360-362 / tagging.py:148-150: action < action_space.n (8, or 8 + A with tagging) */
    int space_n = 8 + (tagging ? A : 0);
    for (int i = 0; i < A; i++)
        if (actions[i] >= space_n) return SO_ERR_ASSERT;
    /* base.py:379-382: role-relative index must exist in agent_action_map[i] (IndexError otherwise).
     * The reference raises in the middle of the agent loop; this restatement (and the product) checks
     * before mutating anything.  Negative indices (Python wrap-around) are rejected too. */
    for (int i = 0; i < A; i++)
        if (actions[i] < 0 || actions[i] >= so_n_actions(e, i)) return SO_ERR_INDEX;

    e->metrics[SO_M_TOTAL_TIME_STEPS] += 1; /* base.py:366 / tagging.py:152 */
    double team_reward = 0.0;
    for (int i = 0; i < A; i++) /* base.py:369 zeros; tagging.py:162 ones*time_step_reward */
        e->rewards[i] = tagging ? 1.0 * e->cfg.time_step_reward : 0.0;

    for (int i = 0; i < A; i++) e->order[i] = i;
    draw_align(e);
    if (e->cfg.is_action_order_random) { /* base.py:372-374: np.random.shuffle(agent order) */
        if (e->rng.kind == SO_RNG_PHILOX) {
            /* production protocol (so_action_layout): a uniform random order built as turn RANKS from the digits that
             * FOLLOW the tick's action draws in its action-stream words.  Agents are placed one after the other: draw
             * k = 1 .. A-1 (range k + 1) is the slot agent k takes among agents 0 .. k, every earlier agent at a slot >= it
             * moves one up.  A draw that shares a word with action draws continues from what those left, lo32(word *
             * product of their ranges).  (Same distribution as np.random.shuffle, different mapping from the words.) */
            int rank[SO_MAX_AGENTS];
            uint32_t w = 0;
            rank[0] = 0;
            for (int k = 1, d = A; k < A; k++, d++) {
                if (e->aw_word[d] != e->aw_word[d - 1]) {
                    w = action_word(e, e->rng.tick * (uint64_t)e->aw_W + e->aw_word[d]);
                } else if (d == A) {
                    w = action_word(e, e->rng.tick * (uint64_t)e->aw_W + e->aw_word[d]);
                    for (int q = 0; q < A; q++)
                        if (e->aw_word[q] == e->aw_word[d]) w *= (uint32_t)so_n_actions(e, q);
                }
                uint64_t p = (uint64_t)w * (uint64_t)(uint32_t)(k + 1);
                int slot = (int)(p >> 32);
                w = (uint32_t)p;
                for (int q = 0; q < k; q++)
                    if (rank[q] >= slot) rank[q]++;
                rank[k] = slot;
            }
            for (int i = 0; i < A; i++) e->order[rank[i]] = i;
        } else {
            np_shuffle(e, e->order, A);
        }
    }

    for (int k = 0; k < A; k++) {
        int idx = e->order[k];
        int a = actions[idx];
        int nr = e->n_role_actions[idx];
        if (tagging && a >= nr) {
            /* tagging.py:68-75,103-110: k-th OTHER agent ascending; actor's own aliveness is not checked */
            int target = a - nr;
            if (target >= idx) target += 1;
            if (e->used[idx] == 0 && e->alive[target] > 0) {
                e->counts[target] += 1;
                e->used[idx] = 1;
            }
        } else {
            agent_step(e, idx, role_action(e, e->imp_mask[idx], a));
        }
    }

    if (tagging) {
        for (int i = 0; i < A; i++) e->counts[i] *= e->alive[i]; /* tagging.py:180 */
        e->timer += 1;                                           /* tagging.py:182 */
        if (e->timer >= e->cfg.tag_reset_interval) {             /* tagging.py:184-207 */
            int best = 0;
            for (int i = 1; i < A; i++)
                if (e->counts[i] > e->counts[best]) best = i; /* np.argmax: first maximum */
            int highest = e->counts[best];
            int alive_sum = 0;
            for (int i = 0; i < A; i++) alive_sum += e->alive[i];
            int quorum = (alive_sum + 1) / 2;
            if (highest >= quorum) {
                e->alive[best] = 0;
                int is_imp = e->imp_mask[best];
                team_reward += e->cfg.vote_reward * (is_imp ? -1 : 1); /* tagging.py:196, sign as coded */
                if (is_imp) e->metrics[SO_M_IMP_VOTED_OUT] += 1;
                else e->metrics[SO_M_CREW_VOTED_OUT] += 1;
            }
            for (int i = 0; i < A; i++) { /* tagging.py:237-241 */
                e->counts[i] = 0;
                e->used[i] = 0;
            }
            e->timer = 0;
        }
    }

    double win_reward = 0.0;
    int done = check_win(e, &win_reward); /* base.py:384 / tagging.py:209 */
    team_reward += win_reward;
    merge_rewards(e, team_reward); /* base.py:387 / tagging.py:213 */
    if (!tagging) {
        /* base.py:389-390 (also catches -0.0); tagging.py has no such fill */
        for (int i = 0; i < A; i++)
            if (e->rewards[i] == 0.0) e->rewards[i] = e->cfg.time_step_reward;
    }
    int truncated = 0;
    if (e->t == e->cfg.max_time_steps - 1) truncated = 1; /* base.py:392-395: t saturates */
    else e->t += 1;

    e->rng.tick += 1; /* production action stream: one tick per step taken */
    if (rewards)
        for (int i = 0; i < A; i++) rewards[i] = e->rewards[i];
    if (done_out) *done_out = done;
    if (trunc_out) *trunc_out = truncated;
    return SO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * batched helpers
 * ---------------------------------------------------------------------------------------------- */
static int pick_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) return omp_get_max_threads();
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

void so_batch_reset(so_env *envs, int64_t B, int threads) {
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t b = 0; b < B; b++) so_reset(&envs[b]);
}

int so_batch_step(so_env *envs, int64_t B, const int32_t *actions, double *rewards, uint8_t *done,
                  uint8_t *trunc, int threads) {
    int nt = pick_threads(threads);
    (void)nt;
    int worst = 0;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(min : worst)
    for (int64_t b = 0; b < B; b++) {
        int A = envs[b].A;
        int32_t d = 0, t = 0;
        int rc = so_step(&envs[b], actions + b * A, rewards ? rewards + b * A : NULL, &d, &t);
        if (rc < worst) worst = rc;
        if (done) done[b] = (uint8_t)d;
        if (trunc) trunc[b] = (uint8_t)t;
    }
    return worst;
}

void so_batch_sample_actions(so_env *envs, int64_t B, int32_t *actions /*[B][A]*/) {
    for (int64_t b = 0; b < B; b++) so_sample_actions(&envs[b], actions + b * envs[b].A);
}

void so_batch_reset_masked(so_env *envs, int64_t B, const uint8_t *mask) {
    for (int64_t b = 0; b < B; b++)
        if (mask[b]) so_reset(&envs[b]);
}

void so_batch_obs_raw(const so_env *envs, int64_t B, uint8_t *out /*[B][F]*/) {
    int F = so_obs_raw_size(&envs[0]);
    double tmp[5 * SO_MAX_AGENTS + 3 * SO_MAX_JOBS + 1];
    for (int64_t b = 0; b < B; b++) {
        so_obs_raw(&envs[b], tmp);
        for (int k = 0; k < F; k++) out[b * F + k] = (uint8_t)tmp[k];
    }
}

void so_batch_export_episode(const so_env *envs, int64_t B, uint32_t *episode) {
    for (int64_t b = 0; b < B; b++) episode[b] = envs[b].rng.episode;
}

/* dense export of the batched state (test convenience; avoids per-env Python loops) */
void so_batch_export(const so_env *envs, int64_t B, int32_t *pos, uint8_t *alive, uint8_t *imp, int32_t *jobpos,
                     uint8_t *jobdone, uint8_t *used, int32_t *counts, int32_t *timer, int32_t *t, int64_t *metrics,
                     uint64_t *cursor) {
    for (int64_t b = 0; b < B; b++) {
        const so_env *e = &envs[b];
        const int A = e->A, J = e->J;
        for (int i = 0; i < A; i++) {
            if (pos) { pos[(b * A + i) * 2] = e->pos[i][0]; pos[(b * A + i) * 2 + 1] = e->pos[i][1]; }
            if (alive) alive[b * A + i] = (uint8_t)e->alive[i];
            if (imp) imp[b * A + i] = (uint8_t)e->imp_mask[i];
            if (used) used[b * A + i] = (uint8_t)e->used[i];
            if (counts) counts[b * A + i] = e->counts[i];
        }
        for (int j = 0; j < J; j++) {
            if (jobpos) { jobpos[(b * J + j) * 2] = e->jobpos[j][0]; jobpos[(b * J + j) * 2 + 1] = e->jobpos[j][1]; }
            if (jobdone) jobdone[b * J + j] = (uint8_t)e->jobdone[j];
        }
        if (timer) timer[b] = e->timer;
        if (t) t[b] = e->t;
        if (metrics) for (int k = 0; k < SO_N_METRICS; k++) metrics[b * SO_N_METRICS + k] = e->metrics[k];
        if (cursor) cursor[b] = e->rng.cursor;
    }
}

int64_t so_batch_random_rollout(so_env *envs, int64_t B, int64_t steps, int threads, int64_t *episodes_out,
                                double *reward_sum_out) {
    int nt = pick_threads(threads);
    (void)nt;
    int64_t episodes = 0;
    double rsum = 0.0;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : episodes, rsum)
    for (int64_t b = 0; b < B; b++) {
        so_env *e = &envs[b];
        int32_t act[SO_MAX_AGENTS];
        double rew[SO_MAX_AGENTS];
        so_reset(e);
        for (int64_t s = 0; s < steps; s++) {
            int32_t d = 0, t = 0;
            so_sample_actions(e, act);
            so_step(e, act, rew, &d, &t);
            for (int i = 0; i < e->A; i++) rsum += rew[i];
            if (d || t) {
                episodes++;
                so_reset(e);
            }
        }
    }
    if (episodes_out) *episodes_out = episodes;
    if (reward_sum_out) *reward_sum_out = rsum;
    return B * steps;
}

/* ------------------------------------------------------------------------------------------------
 * observations
 * ---------------------------------------------------------------------------------------------- */
/* base.py:234-235 flatten_state over observation_space (base.py:211-228; tagging.py:42-60) */
int so_obs_raw_size(const so_env *e) {
    int n = 3 * e->A;
    if (e->cfg.variant == SO_VARIANT_TAGGING) return n + 3 * e->J + 2 * e->A + 1;
    if (e->J > 0) n += 3 * e->J;
    return n;
}

void so_obs_raw(const so_env *e, double *out) {
    int k = 0;
    for (int i = 0; i < e->A; i++) {
        out[k++] = e->pos[i][0];
        out[k++] = e->pos[i][1];
    }
    for (int i = 0; i < e->A; i++) out[k++] = e->alive[i];
    for (int j = 0; j < e->J; j++) {
        out[k++] = e->jobpos[j][0];
        out[k++] = e->jobpos[j][1];
    }
    for (int j = 0; j < e->J; j++) out[k++] = e->jobdone[j];
    if (e->cfg.variant == SO_VARIANT_TAGGING) {
        for (int i = 0; i < e->A; i++) out[k++] = e->used[i];
        for (int i = 0; i < e->A; i++) out[k++] = e->counts[i];
        out[k++] = e->cfg.tag_reset_interval - e->timer;
    }
}

static int flat_component_size(const so_env *e, int c) {
    int A = e->A, N = e->cfg.grid_n;
    switch (c) {
    case SO_F_ONEHOT_POS: return A * 2 * N;    /* component.py:245-247 */
    case SO_F_COORD_POS: return 2 * A;         /* component.py:402-403 */
    case SO_F_ALIVE_CREW: return A - 1;        /* component.py:424-425 */
    case SO_F_L1_CREW: return e->cfg.n_crew;   /* component.py:451-452 */
    case SO_F_CLOSEST_CREW: return e->cfg.n_crew;
    case SO_F_WALLS3X3: return 9;
    case SO_F_DIST_TO_IMP: return (A - 1) * 2;
    case SO_F_ROOM_LOC: return 8;
    case SO_F_SCENT: return 4;
    default: return -1;
    }
}

int so_obs_flat_size(const so_env *e, const int32_t *components, int n_components) {
    int n = 0;
    for (int c = 0; c < n_components; c++) {
        int s = flat_component_size(e, components[c]);
        if (s < 0) return -1;
        n += s;
    }
    return n;
}

/* FlatFeaturizer over a CompositeFeaturizer (model_ready.py:309-354, component.py:134-149): the
 * components' vectors concatenated in list order.  All "imposter" components assume agent 0 is the
 * imposter, exactly as the reference does (component.py:262,289,354,439,466). */
int so_obs_flat(const so_env *e, const int32_t *components, int n_components, float *out) {
    const int A = e->A, N = e->cfg.grid_n, NC = e->cfg.n_crew;
    const int ix = e->pos[0][0], iy = e->pos[0][1];
    int k = 0;
    for (int c = 0; c < n_components; c++) {
        int sz = flat_component_size(e, components[c]);
        if (sz < 0) return -1;
        float *o = out + k;
        for (int i = 0; i < sz; i++) o[i] = 0.0f;
        switch (components[c]) {
        case SO_F_ONEHOT_POS: /* component.py:226-240 */
            for (int i = 0; i < A; i++)
                if (e->alive[i]) {
                    o[i * 2 * N + e->pos[i][0]] = 1.0f;
                    o[i * 2 * N + N + e->pos[i][1]] = 1.0f;
                }
            break;
        case SO_F_COORD_POS: /* component.py:389-399 */
            for (int i = 0; i < A; i++) {
                o[2 * i] = (float)e->pos[i][0];
                o[2 * i + 1] = (float)e->pos[i][1];
            }
            break;
        case SO_F_ALIVE_CREW: /* component.py:411-421 */
            for (int i = 1; i < A; i++)
                if (e->alive[i]) o[i - 1] = 1.0f;
            break;
        case SO_F_L1_CREW: /* component.py:433-448 (valid when A-1 == n_crew) */
            if (A - 1 != NC) return -2;
            for (int i = 0; i < NC; i++) o[i] = -1.0f;
            for (int i = 1; i < A; i++)
                if (e->alive[i]) o[i - 1] = (float)(abs(ix - e->pos[i][0]) + abs(iy - e->pos[i][1]));
            break;
        case SO_F_CLOSEST_CREW: { /* component.py:460-478 */
            if (A - 1 != NC) return -2;
            float l1[SO_MAX_AGENTS];
            for (int i = 0; i < NC; i++) l1[i] = (float)(N + N); /* ones*n_cols + n_rows */
            for (int i = 1; i < A; i++)
                if (e->alive[i]) l1[i - 1] = (float)(abs(ix - e->pos[i][0]) + abs(iy - e->pos[i][1]));
            int best = 0;
            for (int i = 1; i < NC; i++)
                if (l1[i] < l1[best]) best = i; /* torch.argmin: first minimum */
            o[best] = 1.0f;
            break;
        }
        case SO_F_WALLS3X3: /* component.py:286-296: zero-padded grid[x,y], 3x3 around agent 0 */
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    int gx = ix + a - 1, gy = iy + b - 1;
                    int v = (gx >= 0 && gy >= 0 && gx < N && gy < N) ? (e->cfg.grid[gx][gy] != 0) : 0;
                    o[a * 3 + b] = (float)v;
                }
            break;
        case SO_F_DIST_TO_IMP: { /* component.py:255-273: alive non-0 agents packed left */
            int p = 0;
            for (int i = 1; i < A; i++)
                if (e->alive[i]) {
                    o[p] = (float)(ix - e->pos[i][0]);
                    o[p + 1] = (float)(iy - e->pos[i][1]);
                    p += 2;
                }
            break;
        }
        case SO_F_ROOM_LOC: /* component.py:8-17,308-329: fixed 9x9 quadrant masks */
            if (N != 9) return -2;
            for (int i = 0; i < A; i++) {
                if (!e->alive[i]) continue;
                int x = e->pos[i][0], y = e->pos[i][1];
                int rooms[4] = {x < 5 && y < 5, x < 5 && y >= 5, x >= 5 && y >= 5, x >= 5 && y < 5};
                for (int r = 0; r < 4; r++) o[(i == 0 ? 0 : 4) + r] += (float)rooms[r];
            }
            break;
        case SO_F_SCENT: /* component.py:344-375: python float64 scent, accumulated into a float32 tensor */
            for (int i = 1; i < A; i++) {
                if (!e->alive[i]) continue;
                int dx = e->pos[i][0] - ix, dy = e->pos[i][1] - iy;
                double xs = (double)(N - dx) / (double)N;
                double ys = (double)(N - dy) / (double)N;
                if (xs > 0) o[0] = o[0] + (float)xs;
                else o[1] = o[1] + (float)xs;
                if (ys > 0) o[2] = o[2] + (float)ys;
                else o[3] = o[3] + (float)ys;
            }
            break;
        }
        k += sz;
    }
    return k;
}

/* GlobalFeaturizer planes (model_ready.py:230-247; component.py:90-100,116-127):
 * spatial[i][x][y] = 1 iff agent i alive there; spatial[A + done][x][y] = 1 per job;
 * non_spatial = [alive(A), (tag_counts(A) with tagging), job_status(J)] */
void so_obs_planes(const so_env *e, float *spatial, float *non_spatial) {
    const int A = e->A, J = e->J, N = e->cfg.grid_n;
    for (int i = 0; i < (A + 2) * N * N; i++) spatial[i] = 0.0f;
    for (int i = 0; i < A; i++)
        if (e->alive[i]) spatial[(i * N + e->pos[i][0]) * N + e->pos[i][1]] = 1.0f;
    for (int j = 0; j < J; j++)
        spatial[((A + (e->jobdone[j] ? 1 : 0)) * N + e->jobpos[j][0]) * N + e->jobpos[j][1]] = 1.0f;
    int k = 0;
    for (int i = 0; i < A; i++) non_spatial[k++] = (float)e->alive[i];
    if (e->cfg.variant == SO_VARIANT_TAGGING)
        for (int i = 0; i < A; i++) non_spatial[k++] = (float)e->counts[i];
    for (int j = 0; j < J; j++) non_spatial[k++] = (float)e->jobdone[j];
}
