// susnet_capi.hip -- kernels + the C ABI declared in include/susnet.h (libsusnet_hip.so, gfx950 only).
//
// Build:  hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libsusnet_hip.so susnet_capi.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "susnet_kernels.h"
#include "susnet_family.h"
#include "susnet_qnet.h"

namespace susnet {
SUSNET_DECLARE(GenericSpec) SUSNET_DECLARE(SpecCfg2) SUSNET_DECLARE(SpecCfg3) SUSNET_DECLARE(SpecCfg4) SUSNET_DECLARE(SpecTag5)
SUSNET_DECLARE(SpecA<2>)
#define X SUSNET_FAMILY_DECLARE
SUSNET_FAMILY(X)
#undef X
} // namespace susnet
using namespace susnet;

// the byte-parallel family (susnet_family.h): which instantiation serves (agents, variant, order, imposters), and its launchers
struct FamilyEntry {
    int A, variant, order_random, n_imp;
    void (*rollout)(bool, int, dim3, dim3, size_t, hipStream_t, const Consts &, const State &, const RolloutArgs &, const ObsArgs &);
    void (*step)(bool, dim3, dim3, size_t, hipStream_t, const Consts &, const State &, const StepArgs &, const ObsArgs &);
    int record_dwords, head_dwords, tagging;
};
#define X(A_, V_, O_, N_) {A_, V_, O_, N_, &launch_rollout<SpecFam<A_, V_, O_, N_>>, &launch_step<SpecFam<A_, V_, O_, N_>>, \
                           FamRecord<SpecFam<A_, V_, O_, N_>>::kDwords, FamRecord<SpecFam<A_, V_, O_, N_>>::kHeadDwords, V_ == SUSNET_VARIANT_TAGGING},
static const FamilyEntry kFamily[] = {SUSNET_FAMILY(X)};
#undef X
constexpr int kFamilySpecBase = 100; // pick_spec() values kFamilySpecBase + i = kFamily[i]
static inline bool is_family(int spec) { return spec >= kFamilySpecBase; }
static inline bool is_swar_spec(int spec) { return spec == 3 || spec == 4 || spec == 6 || is_family(spec); }

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
template <class RNG>
__global__ __launch_bounds__(kBlock) void k_reset(Consts c, State s, const uint8_t *mask, ObsArgs o) {
    using S = GenericSpec;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kBlock, b = b0 + tid;
    const bool active = b < c.B;
    LdsStore st;
    Tables T = setup_lds<S, false>(c, smem, tid, st);
    if (tid < 16) T.comp[tid] = (uint32_t)o.comp[tid];
    wave_lds_fence();
    Env e = {};
    if (active) {
        load_env<S>(c, s, st, b, e);
        if (!mask || mask[b]) {
            RNG rng = make_rng<RNG>(c, s, b);
            reset_env<S>(c, T, st, tid, e, rng);
            zero_metrics(e); // metrics.reset(), base.py:270
            store_env<S>(c, s, st, b, e, true);
            finish_rng(s, b, rng);
        }
    }
    int nrows = (int)((c.B - b0) < kBlock ? (c.B - b0) : kBlock);
    write_obs<S>(c, o, T, st, tid, e, active, b0, nrows, 0);
}

template <class RNG>
__global__ __launch_bounds__(kBlock) void k_sample(Consts c, State s, void *out, int32_t dtype, int64_t sa, int64_t sb, uint64_t tick) {
    using S = GenericSpec;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * kBlock + tid;
    LdsStore st;
    setup_lds<S, false>(c, smem, tid, st);
    if (b >= c.B) return;
    Env e = {};
    load_env<S>(c, s, st, b, e);
    RNG rng = make_rng<RNG>(c, s, b);
    ActionStream as;
    as.init();
    if (c.dev_tick) tick = uniform64(s.tickw[b]);
    sample_actions_env<S>(c, st, e, rng, as, tick);
    for (int i = 0; i < c.A; i++) store_action(out, dtype, (int64_t)i * sa + b * sb, st.act(i));
    finish_rng(s, b, rng);
}

// sample_actions() on the production stream.  The action stream is indexed by (env id, tick) alone, so all the kernel needs of
// an environment is its roles (which agents draw from the imposter's action list) and the tick: no LDS tables, no state load
// beyond the role bit of the agent words, the draws go straight to the output.  (One launch per tick is latency-bound: the
// generic k_sample above spends three dependent memory round trips before its first draw.)
struct ActionSink {
    void *out;
    int32_t dtype;
    int64_t sa, k0;
    __device__ __forceinline__ void set_act(int i, uint32_t a) const { store_action(out, dtype, (int64_t)i * sa + k0, a); }
};
__global__ __launch_bounds__(kBlock) void k_sample_philox(Consts c, State s, void *out, int32_t dtype, int64_t sa, int64_t sb, uint64_t tick) {
    using S = GenericSpec;
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; // (< Bp: rows are padded, every lane may load)
    warm_kernargs<sizeof(Consts) + sizeof(State) + 40>();
    uint32_t roles = (1u << c.n_imp) - 1u; // fixed roles: the first n_imposters agents (base.py:286-290 without the shuffle)
    if (c.shuffle_imp) {                   // else bit 9 of every agent word; loads first, no branch between them
        uint32_t w[SUSNET_MAX_AGENTS];
#pragma unroll
        for (int i = 0; i < SUSNET_MAX_AGENTS; i++) w[i] = (uint32_t)s.agent[(size_t)(i < c.A ? i : c.A - 1) * c.Bp + b];
        roles = 0;
#pragma unroll
        for (int i = 0; i < SUSNET_MAX_AGENTS; i++) roles |= (i < c.A ? (w[i] >> 9) & 1u : 0u) << i;
    }
    uint64_t tick_word = tick;
    if (c.dev_tick) tick_word = s.tickw[b];
    if (b >= c.B) return;
    Env e = {};
    e.imp = roles;
    PhiloxRng rng;
    rng.init(c.seed, c.env_id_base + (uint64_t)b, 0ull);
    ActionStream as;
    as.init();
    ActionSink sink = {out, dtype, sa, b * sb};
    sample_actions_env<S>(c, sink, e, rng, as, uniform64(tick_word));
}

// susnet_policy_actions: the acting step of the policy loops (visualize.py:547-562) in one launch.  Like k_sample_philox it
// needs of an environment only its roles and the tick; an agent's action is the argmax of its team's Q row (first maximum,
// like torch.argmax), or -- q_crew == NULL -- the crew's draw from the action stream.  The draws are made for EVERY agent and
// the imposters' are overwritten: the digits of a word depend on the draws before them, so the crew's values are exactly the
// ones susnet_sample_actions returns.
struct PolicyActionSink {
    void *out;
    int32_t dtype;
    int64_t sa, k0;
    uint32_t roles, a_imp, a_crew; // a_crew = ~0u: keep the sampled index
    float eps;                     // epsilon-greedy / dead mask: as PolicyStepSink (susnet_kernels.h)
    uint32_t alive, mask_dead;
    const PhiloxRng *rng;
    ActionStream *xs;
    uint64_t xbase;
    __device__ __forceinline__ void set_act(int i, uint32_t sampled) const {
        uint32_t a = ((roles >> i) & 1u) ? a_imp : (a_crew != ~0u ? a_crew : sampled);
        if (eps > 0.0f) {
            const float u = (float)(xs->word(*rng, xbase + (uint64_t)i) >> 8) * 5.9604644775390625e-08f;
            a = u <= eps ? sampled : a;
        }
        if (mask_dead && !((alive >> i) & 1u)) a = 0u;
        store_action(out, dtype, (int64_t)i * sa + k0, a);
    }
};
__device__ __forceinline__ uint32_t argmax_row(const float *q, int n) {
    uint32_t best = 0;
    float hi = q[0];
    for (int k = 1; k < n; k++) {
        const float v = q[k];
        if (v > hi) { hi = v; best = (uint32_t)k; }
    }
    return best;
}
__global__ __launch_bounds__(kBlock) void k_policy_actions(Consts c, State s, const float *q_imp, const float *q_crew, int n_imp_actions,
                                                           int n_crew_actions, void *out, int32_t dtype, int64_t sa, int64_t sb, uint64_t tick, float epsilon,
                                                           int mask_dead) {
    using S = GenericSpec;
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; // (< Bp: rows are padded, every lane may load its roles)
    uint32_t roles = (1u << c.n_imp) - 1u; // fixed roles: the first n_imposters agents (base.py:286-290 without the shuffle)
    uint32_t alive = ~0u;
    if (c.shuffle_imp || mask_dead) {
        uint32_t w[SUSNET_MAX_AGENTS];
#pragma unroll
        for (int i = 0; i < SUSNET_MAX_AGENTS; i++) w[i] = (uint32_t)s.agent[(size_t)(i < c.A ? i : c.A - 1) * c.Bp + b];
        uint32_t r2 = 0;
        alive = 0;
#pragma unroll
        for (int i = 0; i < SUSNET_MAX_AGENTS; i++) {
            r2 |= (i < c.A ? (w[i] >> 9) & 1u : 0u) << i;
            alive |= (i < c.A ? (w[i] >> 8) & 1u : 0u) << i;
        }
        if (c.shuffle_imp) roles = r2;
    }
    uint64_t tick_word = tick;
    if (c.dev_tick) tick_word = s.tickw[b];
    if (b >= c.B) return;
    PhiloxRng rng;
    rng.init(c.seed, c.env_id_base + (uint64_t)b, 0ull);
    ActionStream as, xs;
    as.init();
    xs.init(kExploreStreamTag);
    const uint64_t tk = uniform64(tick_word);
    PolicyActionSink sink = {out, dtype, sa, b * sb, roles, argmax_row(q_imp + b * n_imp_actions, n_imp_actions),
                             q_crew ? argmax_row(q_crew + b * n_crew_actions, n_crew_actions) : ~0u, epsilon, alive, (uint32_t)mask_dead, &rng, &xs,
                             tk * (uint64_t)c.A};
    if (q_crew && !(epsilon > 0.0f)) {
        for (int i = 0; i < c.A; i++) sink.set_act(i, 0u);
        return;
    }
    Env e = {};
    e.imp = roles;
    sample_actions_env<S>(c, sink, e, rng, as, tk);
}

// Device-resident step counter (susnet_device_tick): one copy per environment, read and advanced by the lane that owns the
// environment inside the stepping kernels themselves -- no launch of its own, no word that one workgroup writes while another reads.
__global__ void k_fill_tick(Consts c, State s, uint64_t tick) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < c.Bp) s.tickw[b] = tick;
}

__global__ __launch_bounds__(kBlock) void k_observe(Consts c, State s, ObsArgs o) {
    using S = GenericSpec;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kBlock, b = b0 + tid;
    const bool active = b < c.B;
    LdsStore st;
    Tables T = setup_lds<S, false>(c, smem, tid, st);
    if (tid < 16) T.comp[tid] = (uint32_t)o.comp[tid];
    wave_lds_fence();
    Env e = {};
    if (active) load_env<S>(c, s, st, b, e);
    int nrows = (int)((c.B - b0) < kBlock ? (c.B - b0) : kBlock);
    write_obs<S>(c, o, T, st, tid, e, active, b0, nrows, 0);
}

// Observation of caller-supplied flattened states (flatten_state order) instead of the handle's environments:
// lane r parses row b0 + r into the same per-lane storage the state loader fills, then the wave runs the same
// cooperative writer.  A wave's 64 rows are contiguous in memory, so the per-lane element reads hit the same lines.
__device__ __forceinline__ int row_value(const void *rows, int dtype, int64_t k) {
    switch (dtype) {
    case SUSNET_U8: return (int)static_cast<const uint8_t *>(rows)[k];
    case SUSNET_I32: return static_cast<const int32_t *>(rows)[k];
    case SUSNET_I64: return (int)static_cast<const int64_t *>(rows)[k];
    case SUSNET_F32: return (int)static_cast<const float *>(rows)[k];
    default: return (int)static_cast<const double *>(rows)[k];
    }
}
__global__ __launch_bounds__(kBlock) void k_featurize(Consts c, const void *rows, int dtype, int64_t n_rows, int S_row, uint32_t *err, ObsArgs o) {
    using S = GenericSpec;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kBlock, b = b0 + tid;
    bool active = b < n_rows;
    LdsStore st;
    Tables T = setup_lds<S, false>(c, smem, tid, st);
    if (tid < 16) T.comp[tid] = (uint32_t)o.comp[tid];
    wave_lds_fence();
    Env e = {};
    if (active) {
        const int A = c.A, J = c.J, N = c.N;
        const bool tagging = c.variant == SUSNET_VARIANT_TAGGING;
        const int64_t base = b * S_row;
        bool ok = true;
        int k = 0;
        for (int i = 0; i < A; i++, k += 2) {
            const int x = row_value(rows, dtype, base + k), y = row_value(rows, dtype, base + k + 1);
            ok = ok && (unsigned)x < (unsigned)N && (unsigned)y < (unsigned)N;
            st.set_agent(i, (uint32_t)(x & 15) | ((uint32_t)(y & 15) << 4), 0u);
        }
        for (int i = 0; i < A; i++, k++) e.alive |= (row_value(rows, dtype, base + k) != 0 ? 1u : 0u) << i;
        if (J > 0 || tagging) {
            for (int j = 0; j < J; j++, k += 2) {
                const int x = row_value(rows, dtype, base + k), y = row_value(rows, dtype, base + k + 1);
                ok = ok && (unsigned)x < (unsigned)N && (unsigned)y < (unsigned)N;
                st.set_job(j, (uint32_t)(x & 15) | ((uint32_t)(y & 15) << 4));
            }
            for (int j = 0; j < J; j++, k++) e.jd |= (row_value(rows, dtype, base + k) != 0 ? 1u : 0u) << j;
        }
        if (tagging) { // tagging.py:220-230: used[A], tag_counts[A], interval - timer
            for (int i = 0; i < A; i++, k++) e.used |= (row_value(rows, dtype, base + k) != 0 ? 1u : 0u) << i;
            for (int i = 0; i < A; i++, k++) st.set_cnt(i, (uint32_t)row_value(rows, dtype, base + k) & 15u);
            e.timer = (uint32_t)(c.tag_interval - row_value(rows, dtype, base + k));
        }
        if (!ok) {
            atomicOr(err, SUSNET_ERRBIT_ROW);
            active = false; // the row stays all zeros
        }
    }
    const int nrows = (int)((n_rows - b0) < kBlock ? (n_rows - b0) : kBlock);
    write_obs<S>(c, o, T, st, tid, e, active, b0, nrows, 0);
}

// susnet_featurize for the two compiled-in float32 FlatFeaturizer layouts (susnet_flat.h): a lane parses the few fields its row
// needs (cells and alive flags lead every flattened state, base.py:234-235), builds the row's bit mask, and the wave writes its 64
// rows cooperatively.  The generic kernel above zero-fills a byte image, walks a run-time component list and expands the image;
// it keeps serving every other layout.
template <class ROW>
__global__ __launch_bounds__(kBlock) void k_featurize_flat(Consts c, const void *rows, int dtype, int64_t n_rows, int S_row, uint32_t *err, float *out) {
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kBlock, b = b0 + tid;
    const int nrows = (int)((n_rows - b0) < kBlock ? (n_rows - b0) : kBlock);
    constexpr int A = ROW::A, N = ROW::N;
    ROW row;
    row.clear();
    if (b < n_rows) {
        const int64_t base = b * S_row;
        int v[3 * A];
        if (dtype == SUSNET_U8) { // (wave-uniform: one branch for all the row's loads, which then issue back to back)
#pragma unroll
            for (int k = 0; k < 3 * A; k++) v[k] = (int)static_cast<const uint8_t *>(rows)[base + k];
        } else if (dtype == SUSNET_F32) {
#pragma unroll
            for (int k = 0; k < 3 * A; k++) v[k] = (int)static_cast<const float *>(rows)[base + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3 * A; k++) v[k] = row_value(rows, dtype, base + k);
        }
        uint32_t x[A], y[A], al[A];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < A; i++) {
            ok = ok && (unsigned)v[2 * i] < (unsigned)N && (unsigned)v[2 * i + 1] < (unsigned)N;
            x[i] = (uint32_t)v[2 * i] & 15u;
            y[i] = (uint32_t)v[2 * i + 1] & 15u;
            al[i] = v[2 * A + i] != 0 ? 1u : 0u;
        }
        if (ok) row.build(x, y, al);
        else atomicOr(err, SUSNET_ERRBIT_ROW); // the row stays all zeros, like the generic kernel's
    }
    // a buffer of this wave's rows only: no 2 GiB limit on the whole output
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + b0 * ROW::F, 0, nrows * ROW::F * 4, 0x00020000);
    flat_store_wave(row, smem, tid, nrows, r, 0u);
}

static int pick_spec(const Consts &c, bool float_exact, bool force_generic) {
    if (!float_exact) return 0; // compiled-in kernels do the reward arithmetic in float32
    if (force_generic) return 0; // (tests: the same fixtures through the generic LDS-table kernels)
    if (c.A == 2 && c.J == 0 && c.variant == SUSNET_VARIANT_ITG && !c.order_random && !c.shuffle_imp && c.n_imp == 1) return 2;
    if (c.A == 3 && c.J == 4 && c.variant == SUSNET_VARIANT_BASE && c.order_random && c.n_imp == 1) return 3;
    if (c.A == 8 && c.J == 4 && c.variant == SUSNET_VARIANT_BASE && c.order_random && c.n_imp == 2) return 4;
    if (c.A == 5 && c.J == 5 && c.variant == SUSNET_VARIANT_TAGGING && c.order_random && c.n_imp == 1) return 6;
    if (c.J <= 8) // the byte-parallel family: any job count up to 8, roles shuffled or not
        for (size_t i = 0; i < sizeof(kFamily) / sizeof(kFamily[0]); i++)
            if (kFamily[i].A == c.A && kFamily[i].variant == c.variant && kFamily[i].order_random == (c.order_random ? 1 : 0) && kFamily[i].n_imp == c.n_imp)
                return kFamilySpecBase + (int)i;
    // (every 3..8-agent game with at most 8 jobs is in the family -- one to three imposters: base.py:247-249 allows no more below nine agents --,
    // so the per-turn kernels with a compiled-in agent count, SpecA<3..8> of rounds 2-4 and their 400-500 spilled SGPRs, are gone)
    if (c.A == 2 && c.J <= 8) return 12; // SpecA<2>: the 2-agent games that are not the compiled-in one (jobs, shuffled roles) through the per-turn kernels
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_export(Consts c, State s, susnet_state_view v) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= c.B) return;
    const int A = c.A, J = c.J;
    for (int i = 0; i < A; i++) {
        uint32_t w = s.agent[(size_t)i * c.Bp + b];
        if (v.agent_positions) {
            v.agent_positions[(b * A + i) * 2] = (int32_t)(w & 15u);
            v.agent_positions[(b * A + i) * 2 + 1] = (int32_t)((w >> 4) & 15u);
        }
        if (v.alive_agents) v.alive_agents[b * A + i] = (w >> 8) & 1u;
        if (v.imposter_mask) v.imposter_mask[b * A + i] = (w >> 9) & 1u;
        if (v.used_tag_actions) v.used_tag_actions[b * A + i] = (w >> 10) & 1u;
        if (v.tag_counts) v.tag_counts[b * A + i] = (int32_t)((w >> 11) & 15u);
    }
    const uint32_t jd = s.jobdone[b];
    for (int j = 0; j < J; j++) {
        uint32_t w = s.job[(size_t)j * c.Bp + b];
        if (v.job_positions) {
            v.job_positions[(b * J + j) * 2] = (int32_t)(w & 15u);
            v.job_positions[(b * J + j) * 2 + 1] = (int32_t)((w >> 4) & 15u);
        }
        if (v.completed_jobs) v.completed_jobs[b * J + j] = (jd >> j) & 1u;
    }
    if (v.tag_reset_timer) v.tag_reset_timer[b] = s.timer[b];
    if (v.t) v.t[b] = s.t[b];
    if (v.rng_cursor) v.rng_cursor[b] = s.rng[b];
    if (v.episode_index) v.episode_index[b] = s.ep[b];
    if (v.metrics) {
        int64_t *m = v.metrics + b * SUSNET_N_METRICS;
        const uint32_t kv = s.m_kv[b], fl = s.flags[b];
        for (int k = 0; k < SUSNET_N_METRICS; k++) m[k] = 0;
        m[0] = kv & 0xffffu;          // IMP_KILLED_CREW
        m[1] = (kv >> 16) & 0xffu;    // IMP_VOTED_OUT
        m[2] = kv >> 24;              // CREW_VOTED_OUT
        m[3] = s.m_sab[b];            // SABOTAGED_JOBS
        m[4] = s.m_fix[b];            // COMPLETED_JOBS
        m[6] = s.m_steps[b];          // TOTAL_TIME_STEPS
        m[7] = (fl & FLAG_IMP_WON) ? 1 : 0;
        m[8] = (fl & FLAG_CREW_WON) ? 1 : 0;
    }
    if (v.lifetime)
        for (int k = 0; k < SUSNET_N_LIFETIME; k++) v.lifetime[(size_t)k * c.B + b] = s.life[(size_t)k * c.Bp + b];
}

__global__ __launch_bounds__(kBlock) void k_import(Consts c, State s, susnet_state_view v) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= c.B) return;
    const int A = c.A, J = c.J;
    for (int i = 0; i < A; i++) {
        uint32_t w = s.agent[(size_t)i * c.Bp + b];
        uint32_t xy = w & 0xffu, alive = (w >> 8) & 1u, imp = (w >> 9) & 1u, used = (w >> 10) & 1u, cnt = (w >> 11) & 15u;
        if (v.agent_positions)
            xy = ((uint32_t)v.agent_positions[(b * A + i) * 2] & 15u) | (((uint32_t)v.agent_positions[(b * A + i) * 2 + 1] & 15u) << 4);
        if (v.alive_agents) alive = v.alive_agents[b * A + i] ? 1u : 0u;
        if (v.imposter_mask) imp = v.imposter_mask[b * A + i] ? 1u : 0u;
        if (v.used_tag_actions) used = v.used_tag_actions[b * A + i] ? 1u : 0u;
        if (v.tag_counts) cnt = (uint32_t)v.tag_counts[b * A + i] & 15u;
        s.agent[(size_t)i * c.Bp + b] = (uint16_t)pack_agent(xy, alive, imp, used, cnt);
    }
    uint32_t jd = s.jobdone[b];
    for (int j = 0; j < J; j++) {
        if (v.job_positions)
            s.job[(size_t)j * c.Bp + b] = (uint8_t)(((uint32_t)v.job_positions[(b * J + j) * 2] & 15u) |
                                                     (((uint32_t)v.job_positions[(b * J + j) * 2 + 1] & 15u) << 4));
        if (v.completed_jobs) jd = (jd & ~(1u << j)) | ((v.completed_jobs[b * J + j] ? 1u : 0u) << j);
    }
    s.jobdone[b] = (uint16_t)jd;
    if (v.tag_reset_timer) s.timer[b] = (uint16_t)v.tag_reset_timer[b];
    if (v.t) s.t[b] = (uint16_t)v.t[b];
    if (v.rng_cursor) s.rng[b] = v.rng_cursor[b];
    if (v.episode_index) s.ep[b] = v.episode_index[b];
    if (v.metrics) {
        const int64_t *m = v.metrics + b * SUSNET_N_METRICS;
        s.m_kv[b] = ((uint32_t)m[0] & 0xffffu) | (((uint32_t)m[1] & 0xffu) << 16) | (((uint32_t)m[2] & 0xffu) << 24);
        s.m_sab[b] = (uint32_t)m[3];
        s.m_fix[b] = (uint32_t)m[4];
        s.m_steps[b] = (uint32_t)m[6];
        uint32_t fl = s.flags[b] & ~(FLAG_IMP_WON | FLAG_CREW_WON | FLAG_FRESH);
        if (m[7]) fl |= FLAG_IMP_WON;
        if (m[8]) fl |= FLAG_CREW_WON;
        s.flags[b] = (uint8_t)fl;
    }
}

__global__ void k_fill_cursor(Consts c, State s, uint64_t cursor) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < c.B) {
        s.rng[b] = cursor;
        s.ep[b] = 0u; // the RESET stream restarts with the seed
    }
}

// lifetime row sums: grid = (chunks, rows); 64-lane shuffles, one LDS hop across the block's 4 waves, then ONE
// atomic per block into the (pre-zeroed) 64-bit total
__global__ __launch_bounds__(256) void k_reduce_lifetime(Consts c, State s, int64_t *out) {
    __shared__ unsigned long long part[4];
    const int row = blockIdx.y;
    unsigned long long acc = 0;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < c.B; b += (int64_t)gridDim.x * 256) acc += s.life[(size_t)row * c.Bp + b];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(out) + row, part[0] + part[1] + part[2] + part[3]);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct susnet_env {
    susnet_config cfg;
    Consts c;
    State s;
    bool bound = false;
    bool float_exact = false;
    uint64_t ticks = 0; // steps taken (index of the production action stream)
    // test hooks, read ONCE at susnet_create (include/susnet.h SUSNET_OVERRIDE_*)
    bool force_generic = false;
    int ring_tile = 0;   // susnet_ring_append: environments of a wave's (ticks x envs) tile (8 / 16 / 32; 0: consecutive rows per wave)
    uint64_t launch_limit = (1ull << 31) - 1u, launch_limit_default = (1ull << 31) - 1u;
    int spec = 0; // pick_spec(): which compiled-in kernel family serves the handle (0 = generic)
    susnet_layout layout;
    uint64_t off_err, off_agent, off_job, off_jobdone, off_t, off_timer, off_flags, off_rng, off_msteps, off_mfix, off_msab,
        off_mkv, off_life, off_tickw, off_ep;
};

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
// (messages about a handle created under test hooks say so: a stray environment variable is then visible where it bites)
static int fail(const susnet_env *env, int code, const std::string &msg) {
    std::string m = msg;
    if (env && env->layout.test_overrides) {
        m += " [handle created with";
        if (env->layout.test_overrides & SUSNET_OVERRIDE_FORCE_GENERIC) m += " SUSNET_FORCE_GENERIC";
        if (env->layout.test_overrides & SUSNET_OVERRIDE_EPW) m += " SUSNET_EPW";
        if (env->layout.test_overrides & SUSNET_OVERRIDE_TRAJ_MAX_BYTES) m += " SUSNET_TRAJ_MAX_BYTES";
        if (env->layout.test_overrides & SUSNET_OVERRIDE_RING_TILE) m += " SUSNET_RING_TILE";
        m += "]";
    }
    return fail(code, m);
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(SUSNET_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static inline uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

extern "C" int susnet_abi_version(void) { return SUSNET_ABI_VERSION; }
extern "C" const char *susnet_last_error(void) { return g_err.c_str(); }

extern "C" int susnet_create(const susnet_config *cfg, susnet_env **out) {
    if (!cfg || !out) return fail(SUSNET_E_INVALID, "null argument");
    if (cfg->struct_bytes != sizeof(susnet_config) || cfg->abi_version != SUSNET_ABI_VERSION)
        return fail(SUSNET_E_INVALID, "susnet_config size/ABI mismatch");
    int n_imp = cfg->n_imposters;
    double dead = cfg->dead_penalty;
    int order_random = cfg->is_action_order_random;
    if (cfg->variant == SUSNET_VARIANT_ITG) { // pred_prey.py:52-66, 75-76
        n_imp = 1;
        dead = 0.0;
        order_random = 0;
        if (cfg->n_crew <= 0) return fail(SUSNET_E_INVALID, "Must have at least one crew member.");
    } else if (cfg->variant == SUSNET_VARIANT_BASE || cfg->variant == SUSNET_VARIANT_TAGGING) { // base.py:243-249
        if (n_imp <= 0) return fail(SUSNET_E_INVALID, "Must have at least one imposter.");
        if (cfg->n_crew <= 0) return fail(SUSNET_E_INVALID, "Must have at least one crew member.");
        if (cfg->n_jobs < 0) return fail(SUSNET_E_INVALID, "Must non-negative jobs.");
        if (!(n_imp < cfg->n_crew)) return fail(SUSNET_E_INVALID, "Must be more crew members than imposters.");
    } else {
        return fail(SUSNET_E_INVALID, "unknown variant");
    }
    const int A = n_imp + cfg->n_crew, J = cfg->n_jobs, N = cfg->grid_n;
    if (A > SUSNET_MAX_AGENTS || J > SUSNET_MAX_JOBS || J < 0) return fail(SUSNET_E_INVALID, "too many agents/jobs (max 16/16)");
    if (N < 1 || N > SUSNET_MAX_GRID) return fail(SUSNET_E_INVALID, "grid_n must be in [1,16]");
    if (cfg->batch < 1) return fail(SUSNET_E_INVALID, "batch must be >= 1");
    if (cfg->max_time_steps < 1 || cfg->max_time_steps > 65535) return fail(SUSNET_E_INVALID, "max_time_steps must be in [1,65535]");
    if (cfg->rng_mode != SUSNET_RNG_TAPE && cfg->rng_mode != SUSNET_RNG_PHILOX) return fail(SUSNET_E_INVALID, "unknown rng_mode");
    if (cfg->variant == SUSNET_VARIANT_TAGGING && (cfg->tag_reset_interval < 1 || cfg->tag_reset_interval > 255))
        return fail(SUSNET_E_INVALID, "tag_reset_interval must be in [1,255]");

    susnet_env *e = new (std::nothrow) susnet_env();
    if (!e) return fail(SUSNET_E_INVALID, "out of host memory");
    e->cfg = *cfg;
    Consts &c = e->c;
    std::memset(&c, 0, sizeof(c));
    c.B = cfg->batch;
    c.Bp = (int32_t)up((uint64_t)cfg->batch, 256);
    c.A = A; c.J = J; c.N = N; c.n_imp = n_imp; c.n_crew = cfg->n_crew; c.variant = cfg->variant;
    c.max_t = cfg->max_time_steps;
    c.order_random = order_random ? 1 : 0;
    c.shuffle_imp = cfg->shuffle_imposter_index ? 1 : 0;
    c.tag_interval = cfg->tag_reset_interval;
    c.auto_reset = cfg->auto_reset ? 1 : 0;
    // environments per wave in the fused rollout: fewer than 64 when the batch would otherwise leave SIMDs
    // idle (an MI355X has 1024 SIMDs; below one wave per SIMD the kernel is purely latency-bound)
    c.epw = cfg->batch >= 65536 ? 64 : cfg->batch >= 32768 ? 32 : 16;
    uint32_t overrides = 0;
    if (const char *ev = getenv("SUSNET_EPW")) {
        int v = atoi(ev);
        if (v == 16 || v == 32 || v == 64) { c.epw = v; overrides |= SUSNET_OVERRIDE_EPW; }
    }
    if (const char *ev = getenv("SUSNET_FORCE_GENERIC"))
        if (ev[0] == '1') { e->force_generic = true; overrides |= SUSNET_OVERRIDE_FORCE_GENERIC; }
    if (const char *ev = getenv("SUSNET_RING_TILE")) {
        const int v = atoi(ev);
        if (v == 0 || v == 8 || v == 16 || v == 32) { e->ring_tile = v; overrides |= SUSNET_OVERRIDE_RING_TILE; }
    }
    if (const char *ev = getenv("SUSNET_TRAJ_MAX_BYTES")) {
        const long long v = atoll(ev);
        if (v > 0 && (uint64_t)v < e->launch_limit) { e->launch_limit = e->launch_limit_default = (uint64_t)v; overrides |= SUSNET_OVERRIDE_TRAJ_MAX_BYTES; }
    }
    c.nr_imp = cfg->variant == SUSNET_VARIANT_ITG ? 6 : 7;  // pred_prey.py:12-19 / base.py:91-99
    c.nr_crew = cfg->variant == SUSNET_VARIANT_ITG ? 5 : 6; // pred_prey.py:4-10  / base.py:82-89
    { // action-stream word layout of the production protocol (same function the compiled-in kernels evaluate statically)
        const AwLayout L = make_aw_layout(A, (uint32_t)c.nr_imp + (cfg->variant == SUSNET_VARIANT_TAGGING ? (uint32_t)(A - 1) : 0u), order_random != 0);
        c.aw_W = L.W;
        for (int d = 0; d < 32; d++) c.aw_word[d] = L.word[d];
    }
    c.n_valid = 0;
    for (int i = 0; i < N; i++) {
        c.grid_rows[i] = cfg->grid_rows[i] & ((1u << N) - 1u);
        for (int j = 0; j < N; j++)
            if ((cfg->grid_rows[i] >> j) & 1u) reinterpret_cast<uint8_t *>(c.valid_xy)[c.n_valid++] = (uint8_t)(i | (j << 4)); // argwhere order; x = i, y = j
    }
    { // (action row, cell) -> next cell
        uint8_t *mt = reinterpret_cast<uint8_t *>(c.move_tab);
        for (int a = 0; a < 6; a++)
            for (int cell = 0; cell < 256; cell++) {
                int x = cell & 15, y = cell >> 4;
                int nx = x + (a == ACT_RIGHT) - (a == ACT_LEFT), ny = y + (a == ACT_UP) - (a == ACT_DOWN);
                bool ok = a >= 1 && a <= 4 && nx >= 0 && ny >= 0 && nx < N && ny < N && ((c.grid_rows[ny] >> nx) & 1u);
                mt[a * 256 + cell] = (uint8_t)(ok ? (nx | (ny << 4)) : cell);
            }
    }
    if (c.n_valid < 1 || c.n_valid < J) {
        delete e;
        return fail(SUSNET_E_INVALID, "grid has fewer free cells than jobs");
    }
    const double rewards[7] = {cfg->kill_reward, cfg->complete_job_reward, cfg->sabotage_reward, cfg->time_step_reward,
                               cfg->game_end_reward, dead, cfg->vote_reward};
    e->float_exact = true;
    for (int k = 0; k < 7; k++) {
        c.dr[k] = rewards[k];
        c.fr[k] = (float)rewards[k];
        // integers below 2^20: every sum the step forms is exact in float32, signed zeros included
        if (!(rewards[k] == (double)(long long)rewards[k] && rewards[k] > -1048576.0 && rewards[k] < 1048576.0)) e->float_exact = false;
    }
    for (int t = 0; t < 3; t++) // reward table: [win none/crew/imposter][index < n_imposters][dead][assignment code]
        for (int neg = 0; neg < 2; neg++)
            for (int dd = 0; dd < 2; dd++)
                for (int code = 0; code < 4; code++) {
                    double r = 0.0;                                    // base.py:369
                    if (code == RC_KILL) r = rewards[0];               // base.py:514-515
                    else if (code == RC_FIX) r = rewards[1];           // base.py:523
                    else if (code == RC_SAB) r = -1.0 * rewards[2];    // base.py:532
                    const double team = t == 0 ? 0.0 : (t == 1 ? rewards[4] : -1.0 * rewards[4]); // base.py:435,446
                    r += team;                                         // base.py:557
                    if (neg) r *= -1.0;                                // base.py:559
                    if (dd) r = dead;                                  // base.py:562
                    if (r == 0.0) r = rewards[3];                      // base.py:389-390
                    c.rew_tab[t * 16 + neg * 8 + dd * 4 + code] = (float)r;
                }
    if (cfg->variant == SUSNET_VARIANT_TAGGING) {
        // tagging.py:162-213 adds a per-step team reward that the table above does not span: the byte-parallel step
        // (susnet_swar.h) looks up only what an assignment left on an agent and does the rest in registers
        c.rew_tab[0] = (float)(1.0 * rewards[3]);  // tagging.py:162: ones * time_step_reward
        c.rew_tab[RC_KILL] = (float)rewards[0];
        c.rew_tab[RC_FIX] = (float)rewards[1];
        c.rew_tab[RC_SAB] = (float)(-1.0 * rewards[2]);
    }
    c.seed = cfg->seed; c.env_id_base = cfg->env_id_base;
    { // 1v1 ImposterTrainingGround on a grid without walls, every reachable reward an integer in [-127, 127]: susnet_duel.h
        bool ok = e->float_exact && A == 2 && J == 0 && cfg->variant == SUSNET_VARIANT_ITG && !c.shuffle_imp;
        const bool walls = c.n_valid != N * N; // a wall map: the same kernels with the bounds test replaced by the cells' blocked-move bits (DuelWallTable)
        c.duel_lut[0] = c.duel_lut[1] = 0;
        for (int idx = 0; idx < 8 && ok; idx++) {
            const int hit = idx & 1, dead0 = (idx >> 1) & 1, dead1 = (idx >> 2) & 1;
            const int t = dead1 ? 2 : 0; // no jobs: only the imposter can win, when no crew member is alive (pred_prey.py:94-97)
            const float r[2] = {c.rew_tab[t * 16 + 8 + dead0 * 4 + (hit ? (int)RC_KILL : 0)], c.rew_tab[t * 16 + 0 + dead1 * 4 + (hit ? (int)RC_KILL : 0)]};
            for (int a = 0; a < 2; a++) {
                if (!(r[a] >= -127.0f && r[a] <= 127.0f && r[a] == (float)(int)r[a]) || (r[a] == 0.0f && std::signbit(r[a]))) ok = false;
                else c.duel_lut[a] |= (uint64_t)(uint8_t)(int8_t)(int)r[a] << (8 * idx);
            }
        }
        c.duel_fast = ok && !walls ? 1 : 0;
        c.duel_walls = ok && walls ? 1 : 0;
    }

    // state blob layout (every array 256-byte aligned; row stride Bp)
    uint64_t off = 0;
    const uint64_t Bp = (uint64_t)c.Bp;
    auto take = [&](uint64_t bytes) { uint64_t o = off; off = up(off + bytes, 256); return o; };
    e->off_err = take(256);
    e->off_agent = take(2 * Bp * A);
    e->off_job = take(Bp * (J > 0 ? J : 1));
    e->off_jobdone = take(2 * Bp);
    e->off_t = take(2 * Bp);
    e->off_timer = take(2 * Bp);
    e->off_flags = take(Bp);
    e->off_rng = take(8 * Bp);
    e->off_msteps = take(4 * Bp);
    e->off_mfix = take(4 * Bp);
    e->off_msab = take(4 * Bp);
    e->off_mkv = take(4 * Bp);
    e->off_life = take(4 * Bp * SUSNET_N_LIFETIME);
    e->off_tickw = take(8 * Bp);
    e->off_ep = take(4 * Bp);
    susnet_layout &L = e->layout;
    L.state_bytes = off;
    L.state_align = 256;
    L.batch_padded = c.Bp;
    L.n_agents = A;
    const int tag_extra = cfg->variant == SUSNET_VARIANT_TAGGING ? A - 1 : 0;
    L.n_actions_imposter = c.nr_imp + tag_extra;
    L.n_actions_crew = c.nr_crew + tag_extra;
    L.action_space_n = 8 + (cfg->variant == SUSNET_VARIANT_TAGGING ? A : 0);
    L.obs_raw_size = 3 * A + (cfg->variant == SUSNET_VARIANT_TAGGING ? 3 * J + 2 * A + 1 : (J > 0 ? 3 * J : 0));
    L.envs_per_wave = c.epw;
    L.test_overrides = overrides;
    e->spec = pick_spec(c, e->float_exact, e->force_generic);
    *out = e;
    return SUSNET_OK;
}

extern "C" void susnet_destroy(susnet_env *env) { delete env; }

extern "C" int susnet_get_layout(const susnet_env *env, susnet_layout *out) {
    if (!env || !out) return fail(SUSNET_E_INVALID, "null argument");
    *out = env->layout;
    return SUSNET_OK;
}

extern "C" int susnet_bind_state(susnet_env *env, void *blob, uint64_t bytes, void *stream) {
    if (!env || !blob) return fail(SUSNET_E_INVALID, "null argument");
    if (bytes < env->layout.state_bytes) return fail(SUSNET_E_INVALID, "state blob too small");
    if ((uintptr_t)blob % 256) return fail(SUSNET_E_INVALID, "state blob must be 256-byte aligned");
    char *p = static_cast<char *>(blob);
    State &s = env->s;
    const uint32_t *tape = s.tape;
    int64_t tape_len = s.tape_len;
    s.err = reinterpret_cast<uint32_t *>(p + env->off_err);
    s.agent = reinterpret_cast<uint16_t *>(p + env->off_agent);
    s.job = reinterpret_cast<uint8_t *>(p + env->off_job);
    s.jobdone = reinterpret_cast<uint16_t *>(p + env->off_jobdone);
    s.t = reinterpret_cast<uint16_t *>(p + env->off_t);
    s.timer = reinterpret_cast<uint16_t *>(p + env->off_timer);
    s.flags = reinterpret_cast<uint8_t *>(p + env->off_flags);
    s.rng = reinterpret_cast<uint64_t *>(p + env->off_rng);
    s.m_steps = reinterpret_cast<uint32_t *>(p + env->off_msteps);
    s.m_fix = reinterpret_cast<uint32_t *>(p + env->off_mfix);
    s.m_sab = reinterpret_cast<uint32_t *>(p + env->off_msab);
    s.m_kv = reinterpret_cast<uint32_t *>(p + env->off_mkv);
    s.life = reinterpret_cast<uint32_t *>(p + env->off_life);
    s.tickw = reinterpret_cast<uint64_t *>(p + env->off_tickw);
    s.ep = reinterpret_cast<uint32_t *>(p + env->off_ep);
    s.tape = tape;
    s.tape_len = tape_len;
    HIP_TRY(hipMemsetAsync(blob, 0, env->layout.state_bytes, static_cast<hipStream_t>(stream)));
    env->bound = true;
    return SUSNET_OK;
}

extern "C" int susnet_bind_tape(susnet_env *env, const uint32_t *tape, int64_t words_per_env) {
    if (!env) return fail(SUSNET_E_INVALID, "null argument");
    if (env->cfg.rng_mode != SUSNET_RNG_TAPE) return fail(SUSNET_E_INVALID, "handle is not in TAPE mode");
    env->s.tape = tape;
    env->s.tape_len = tape ? words_per_env : 0;
    return SUSNET_OK;
}

static int check_bound(const susnet_env *env) {
    if (!env) return fail(SUSNET_E_INVALID, "null handle");
    if (!env->bound) return fail(SUSNET_E_STATE, "state blob not bound (susnet_bind_state)");
    return SUSNET_OK;
}
static inline dim3 grid_for(const susnet_env *env) { return dim3((unsigned)((env->c.B + kBlock - 1) / kBlock)); }

extern "C" int susnet_seed(susnet_env *env, uint64_t seed, uint64_t cursor, void *stream) {
    if (int rc = check_bound(env)) return rc;
    env->c.seed = seed;
    env->cfg.seed = seed;
    env->ticks = 0;
    if (env->c.dev_tick) HIP_TRY(hipMemsetAsync(env->s.tickw, 0, sizeof(uint64_t) * (size_t)env->c.Bp, static_cast<hipStream_t>(stream)));
    hipLaunchKernelGGL(k_fill_cursor, dim3((unsigned)((env->c.B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), env->c,
                       env->s, cursor);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

// observation plumbing -----------------------------------------------------------------------------
// packed LDS image of `elements` elements of `bits` bits each, in words, rounded to 16 bytes
static inline int img_words(int elements, int bits) { return (int)(((size_t)elements * bits + 127) / 128 * 4); }

static int build_obs(const susnet_env *env, const susnet_obs_spec *spec, ObsArgs &o, int64_t ticks_batch) {
    std::memset(&o, 0, sizeof(o));
    if (!spec || spec->mode == SUSNET_OBS_NONE) return SUSNET_OK;
    const Consts &c = env->c;
    if (spec->dtype != SUSNET_F32 && spec->dtype != SUSNET_U8) return fail(SUSNET_E_INVALID, "obs dtype must be F32 or U8");
    o.mode = spec->mode;
    o.dtype = spec->dtype;
    o.out = spec->out;
    o.out2 = spec->out2;
    if (spec->mode == SUSNET_OBS_RAW) {
        o.F = env->layout.obs_raw_size;
        o.words1 = img_words(kBlock * o.F, 8);
    } else if (spec->mode == SUSNET_OBS_FLAT) {
        if (spec->n_components < 1 || spec->n_components > 16) return fail(SUSNET_E_INVALID, "1..16 flat components");
        o.ncomp = spec->n_components;
        int F = 0;
        for (int i = 0; i < o.ncomp; i++) {
            int comp = spec->components[i];
            int sz = flat_component_size(comp, c.A, c.N, c.n_crew);
            if (sz < 0) return fail(SUSNET_E_INVALID, "unknown flat component");
            if ((comp == SUSNET_F_L1_CREW || comp == SUSNET_F_CLOSEST_CREW) && c.A - 1 != c.n_crew)
                return fail(SUSNET_E_INVALID, "l1_crew/closest_crew need exactly one imposter (component.py:442-446)");
            if (comp == SUSNET_F_ROOM_LOC && c.N != 9) return fail(SUSNET_E_INVALID, "room_loc is defined on the 9x9 grid only");
            o.comp[i] = comp;
            F += sz;
        }
        o.F = F;
        o.words1 = img_words(kBlock * F, 8);
        if (spec->dtype == SUSNET_F32) { // the compiled-in layouts (susnet_flat.h): kernels that have the writer use it
            if (c.A == 2 && c.N == 9 && o.ncomp == 1 && o.comp[0] == SUSNET_F_ONEHOT_POS) o.flat_feat = FEAT_ONEHOT;
            if (c.A == 3 && c.N == 14 && o.ncomp == 3 && o.comp[0] == SUSNET_F_ONEHOT_POS && o.comp[1] == SUSNET_F_ALIVE_CREW &&
                o.comp[2] == SUSNET_F_CLOSEST_CREW)
                o.flat_feat = FEAT_ONEHOT_ALIVE_CLOSEST;
        }
    } else if (spec->mode == SUSNET_OBS_PLANES || spec->mode == SUSNET_OBS_PERSP) {
        o.F = (c.A + 2) * c.N * c.N;
        o.F2 = c.A + c.J + (c.variant == SUSNET_VARIANT_TAGGING ? c.A : 0);
        o.words1 = img_words(kBlock * o.F, 1);
        o.words2 = img_words(kBlock * o.F2, 8);
        o.Fi = o.F;
        o.F2i = o.F2;
        if (spec->mode == SUSNET_OBS_PERSP) { // every agent's rotated copy of the same images
            o.F *= c.A;
            o.F2 *= c.A;
        }
    } else {
        return fail(SUSNET_E_INVALID, "unknown obs mode");
    }
    if (o.mode != SUSNET_OBS_PLANES && o.mode != SUSNET_OBS_PERSP) { o.Fi = o.F; o.F2i = o.F2; }
    if (!o.out) return fail(SUSNET_E_INVALID, "obs.out is null");
    if ((uintptr_t)o.out % 16 || (o.out2 && (uintptr_t)o.out2 % 16)) return fail(SUSNET_E_INVALID, "obs buffers must be 16-byte aligned");
    o.tick_stride = ticks_batch * o.F;
    o.tick_stride2 = ticks_batch * o.F2;
    return SUSNET_OK;
}

// spec: what pick_spec() returned for the launch (0 = the generic kernels, the LDS-column store; 3 / 4 / 6 = the byte-parallel
// configurations, whose rollouts stage the action stream in LDS: HasGroupWords in susnet_device.h)
// rollout: the launch is a fused rollout (the byte-parallel ones also keep the cell -> job map there: susnet_swar.h JobMap)
static size_t lds_bytes(const susnet_env *env, const ObsArgs &o, bool may_reset, int spec = 0, bool rollout = false) {
    const Consts &c = env->c;
    const bool swar = is_swar_spec(spec);
    size_t core = (size_t)lds_core_words(c.A, c.J, spec == 0, swar, (swar && rollout) ? c.N : 0) * 4;
    size_t perm = (may_reset && env->cfg.rng_mode == SUSNET_RNG_TAPE) ? (size_t)c.n_valid * kBlock : 0;
    size_t stage = (size_t)(o.words1 + o.words2) * 4;
    if (spec == 2 && rollout && c.duel_walls) core += (size_t)kDuelWallWords * 4; // the cells' blocked-move bits (susnet_duel.h DuelWallTable)
    return core + (perm > stage ? perm : stage);
}

extern "C" int susnet_obs_size(const susnet_env *env, const susnet_obs_spec *obs, int32_t *size_out, int32_t *size2_out) {
    if (!env || !obs) return fail(SUSNET_E_INVALID, "null argument");
    susnet_obs_spec tmp = *obs;
    static char dummy[16] __attribute__((aligned(16)));
    tmp.out = dummy;
    tmp.out2 = nullptr;
    ObsArgs o;
    if (int rc = build_obs(env, &tmp, o, env->c.B)) return rc;
    if (size_out) *size_out = o.F;
    if (size2_out) *size2_out = o.F2;
    return SUSNET_OK;
}

#define CHECK_LDS(bytes) \
    if ((bytes) > 64 * 1024) return fail(SUSNET_E_INVALID, "observation too large for the LDS staging area")

extern "C" int susnet_reset(susnet_env *env, const uint8_t *mask, const susnet_obs_spec *obs, void *stream) {
    if (int rc = check_bound(env)) return rc;
    ObsArgs o;
    if (int rc = build_obs(env, obs, o, env->c.B)) return rc;
    size_t sh = lds_bytes(env, o, true);
    CHECK_LDS(sh);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (env->cfg.rng_mode == SUSNET_RNG_TAPE)
        hipLaunchKernelGGL(k_reset<TapeRng>, grid_for(env), dim3(kBlock), sh, st, env->c, env->s, mask, o);
    else
        hipLaunchKernelGGL(k_reset<PhiloxRng>, grid_for(env), dim3(kBlock), sh, st, env->c, env->s, mask, o);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

static int strides_for(const susnet_env *env, int layout, int64_t &sa, int64_t &sb) {
    if (layout == SUSNET_LAYOUT_AB) { sa = env->c.B; sb = 1; }
    else if (layout == SUSNET_LAYOUT_BA) { sa = 1; sb = env->c.A; }
    else return fail(SUSNET_E_INVALID, "unknown layout");
    return SUSNET_OK;
}

extern "C" int susnet_sample_actions(susnet_env *env, void *actions_out, int32_t dtype, int32_t layout, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!actions_out) return fail(SUSNET_E_INVALID, "null actions_out");
    if (dtype != SUSNET_U8 && dtype != SUSNET_I32 && dtype != SUSNET_I64) return fail(SUSNET_E_INVALID, "actions dtype must be U8/I32/I64");
    int64_t sa, sb;
    if (int rc = strides_for(env, layout, sa, sb)) return rc;
    ObsArgs o;
    std::memset(&o, 0, sizeof(o));
    size_t sh = lds_bytes(env, o, false);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (env->cfg.rng_mode == SUSNET_RNG_TAPE)
        hipLaunchKernelGGL(k_sample<TapeRng>, grid_for(env), dim3(kBlock), sh, st, env->c, env->s, actions_out, dtype, sa, sb, env->ticks);
    else
        hipLaunchKernelGGL(k_sample_philox, grid_for(env), dim3(kBlock), 0, st, env->c, env->s, actions_out, dtype, sa, sb, env->ticks);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

// epsilon / mask_dead of a susnet_policy_opts (NULL = greedy, dead agents act like everybody else)
static int policy_opts(const susnet_env *env, const susnet_policy_opts *opts, float &eps, int &mask_dead) {
    eps = 0.0f;
    mask_dead = 0;
    if (!opts) return SUSNET_OK;
    if (!(opts->epsilon >= 0.0f && opts->epsilon <= 1.0f)) return fail(SUSNET_E_INVALID, "susnet_policy_opts: epsilon must lie in [0, 1]");
    if (opts->epsilon > 0.0f && env->cfg.rng_mode != SUSNET_RNG_PHILOX)
        return fail(SUSNET_E_INVALID, "susnet_policy_opts: exploration draws come from the production stream: PHILOX handles only");
    eps = opts->epsilon;
    mask_dead = opts->mask_dead != 0;
    return SUSNET_OK;
}
// the crew's network inside the one-kernel tick (susnet_policy_opts.crew_*): everywhere else it must be absent
static int no_crew_network(const susnet_policy_opts *opts, const char *who) {
    if (opts && (opts->crew_packed || opts->crew_dims || opts->crew_q_out))
        return fail(SUSNET_E_INVALID, std::string(who) + ": susnet_policy_opts.crew_* belong to susnet_qnet_policy_step / susnet_qnet_policy_rollout (here the crew's Q rows are an argument)");
    return SUSNET_OK;
}

extern "C" int susnet_policy_actions(susnet_env *env, const float *q_imposter, const float *q_crew, const susnet_policy_opts *opts, void *actions_out,
                                     int32_t dtype, int32_t layout, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!q_imposter || !actions_out) return fail(SUSNET_E_INVALID, "susnet_policy_actions: null q_imposter / actions_out");
    if (dtype != SUSNET_U8 && dtype != SUSNET_I32 && dtype != SUSNET_I64) return fail(SUSNET_E_INVALID, "actions dtype must be U8/I32/I64");
    if (!q_crew && env->cfg.rng_mode != SUSNET_RNG_PHILOX)
        return fail(SUSNET_E_INVALID, "susnet_policy_actions: a random crew (q_crew = NULL) draws from the production stream: PHILOX handles only");
    float eps;
    int mask_dead;
    if (int rc = policy_opts(env, opts, eps, mask_dead)) return rc;
    if (int rc = no_crew_network(opts, "susnet_policy_actions")) return rc;
    int64_t sa, sb;
    if (int rc = strides_for(env, layout, sa, sb)) return rc;
    hipLaunchKernelGGL(k_policy_actions, grid_for(env), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->c, env->s, q_imposter, q_crew,
                       (int)env->layout.n_actions_imposter, (int)env->layout.n_actions_crew, actions_out, dtype, sa, sb, env->ticks, eps, mask_dead);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

// ---- the policy loop's Q-network (susnet_qnet.h) ----
// which compiled-in feature layout (susnet_flat.h) a component list on this handle is: 0 = none
static int qnet_feat(const susnet_env *env, const int32_t *comp, int32_t ncomp) {
    if (!env || !comp) return 0;
    const Consts &c = env->c;
    if (c.A == 2 && c.N == 9 && ncomp == 1 && comp[0] == SUSNET_F_ONEHOT_POS) return FEAT_ONEHOT;
    if (c.A == 2 && c.N == 9 && ncomp == 1 && comp[0] == SUSNET_F_COORD_POS) return FEAT_COORD;
    if (c.A == 3 && c.N == 14 && c.n_imp == 1 && ncomp == 3 && comp[0] == SUSNET_F_ONEHOT_POS && comp[1] == SUSNET_F_ALIVE_CREW &&
        comp[2] == SUSNET_F_CLOSEST_CREW)
        return FEAT_ONEHOT_ALIVE_CLOSEST;
    return 0;
}
template <class ROW>
static bool qnet_dims_ok(const int32_t *dims, int32_t n_dims) {
    using Q = QNet<ROW>;
    const int cap[6] = {Q::F, Q::H1, Q::H2, Q::H3, Q::H4, Q::NO};
    if (!dims || n_dims != 6 || dims[0] != Q::F) return false;
    for (int l = 1; l < 6; l++)
        if (dims[l] < 1 || dims[l] > cap[l]) return false;
    return true;
}
// torch Linear weight [dn][dk] -> 32 x 32 blocks in [kb][nb] order; inside a block lane l holds, as four float4, the 16 k values of row
// n = 32 nb + l % 32 in MFMA-step order: float4 q = columns 32 kb + 8 q + 4 (l / 32) + {0, 1, 2, 3}
static void qnet_pack_dense(const float *W, int dk, int dn, int KP, int NP, float *dst) {
    const int KB = KP / 32, NB = NP / 32;
    for (int kb = 0; kb < KB; kb++)
        for (int nb = 0; nb < NB; nb++)
            for (int q = 0; q < 4; q++)
                for (int l = 0; l < 64; l++)
                    for (int r = 0; r < 4; r++) {
                        const int n = 32 * nb + (l & 31), k = 32 * kb + 8 * q + 4 * (l >> 5) + r;
                        dst[((((size_t)kb * NB + nb) * 4 + q) * 64 + l) * 4 + r] = (n < dn && k < dk) ? W[(size_t)n * dk + k] : 0.0f;
                    }
}
template <class ROW>
static void qnet_pack(const int32_t *d, const float *const *W, const float *const *Bv, const float *slopes, float *out) {
    using Q = QNet<ROW>;
    std::fill(out, out + Q::kPacked, 0.0f);
    // layer 1, transposed: one row per position bit (row kZero stays zero) ...
    if constexpr (ROW::kDeadZero) {
        for (int f = 0; f < Q::kOneHot; f++)
            for (int n = 0; n < d[1]; n++) out[Q::oW1 + f * Q::kRowStride + n] = W[0][(size_t)n * Q::F + f];
    } else { // ... the coordinate layout: row (coordinate c, value k) = k x column c of W1 (one float32 rounding, as torch's product has)
        for (int cc = 0; cc < Q::F; cc++)
            for (int k = 0; k < ROW::N; k++)
                for (int n = 0; n < d[1]; n++) out[Q::oW1 + (cc * ROW::N + k) * Q::kRowStride + n] = (float)k * W[0][(size_t)n * Q::F + cc];
    }
    // ... and one per combination v of the bits behind the one-hots: b1 + the columns of v's set bits, lowest first (float32 sums)
    for (int v = 0; v < (1 << Q::kTailBits); v++)
        for (int n = 0; n < d[1]; n++) {
            float acc = Bv[0][n];
            for (int bit = 0; bit < Q::kTailBits; bit++)
                if ((v >> bit) & 1) acc += W[0][(size_t)n * Q::F + (Q::F - Q::kTailBits) + bit];
            out[Q::oW1 + (Q::kTail + v) * Q::kRowStride + n] = acc;
        }
    const int off_w[4] = {Q::oW2, Q::oW3, Q::oW4, Q::oW5}, off_b[4] = {Q::oB2, Q::oB3, Q::oB4, Q::oB5};
    const int pad[6] = {Q::F, Q::H1, Q::H2, Q::H3, Q::H4, Q::NO};
    for (int l = 1; l < 5; l++) {
        qnet_pack_dense(W[l], d[l], d[l + 1], pad[l], pad[l + 1], out + off_w[l - 1]);
        for (int n = 0; n < d[l + 1]; n++) out[off_b[l - 1] + n] = Bv[l][n];
    }
    for (int l = 0; l < 4; l++) out[Q::oSlope + l] = slopes[l];
}
// (the kernels live in translation units of their own: inst_qnet_*.hip)
namespace susnet {
SUSNET_QNET_FOR(extern, QRow1, QSpec2)
SUSNET_QNET_FOR(extern, QRow3, QSpec3)
SUSNET_QNET_FOR(extern, QRowC, QSpec2)
}

extern "C" int64_t susnet_qnet_packed_floats(const susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims,
                                             int32_t n_dims) {
    switch (qnet_feat(env, components, n_components)) {
    case FEAT_ONEHOT: if (qnet_dims_ok<QRow1>(dims, n_dims)) return QNet<QRow1>::kPacked; break;
    case FEAT_ONEHOT_ALIVE_CLOSEST: if (qnet_dims_ok<QRow3>(dims, n_dims)) return QNet<QRow3>::kPacked; break;
    case FEAT_COORD: if (qnet_dims_ok<QRowC>(dims, n_dims)) return QNet<QRowC>::kPacked; break;
    }
    return fail(SUSNET_E_INVALID, "susnet_qnet: served are five Linear layers [F, <=256, <=128, <=64, <=32, <=32] on the compiled-in feature "
                                  "layouts (onehot_pos or coord_pos on the 2-agent 9x9 game; onehot_pos + alive_crew + closest_crew on the 3-agent 14x14 game)");
}

extern "C" int susnet_qnet_pack(const susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                                const float *const *weights, const float *const *biases, const float *slopes, float *packed) {
    const int64_t n = susnet_qnet_packed_floats(env, components, n_components, dims, n_dims);
    if (n < 0) return (int)n;
    if (!weights || !biases || !slopes || !packed) return fail(SUSNET_E_INVALID, "susnet_qnet_pack: null weights / biases / slopes / packed");
    for (int l = 0; l < 5; l++)
        if (!weights[l] || !biases[l]) return fail(SUSNET_E_INVALID, "susnet_qnet_pack: null layer");
    switch (qnet_feat(env, components, n_components)) {
    case FEAT_ONEHOT: qnet_pack<QRow1>(dims, weights, biases, slopes, packed); break;
    case FEAT_COORD: qnet_pack<QRowC>(dims, weights, biases, slopes, packed); break;
    default: qnet_pack<QRow3>(dims, weights, biases, slopes, packed); break;
    }
    return SUSNET_OK;
}

template <class ROW>
static int qnet_launch(susnet_env *env, const float *packed, float *q_out, int n_out, hipStream_t st) {
    HIP_TRY((qnet_launch_k<ROW>(env->c, env->s, packed, q_out, n_out, st)));
    return SUSNET_OK;
}
extern "C" int susnet_qnet_forward(susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                                   const float *packed, float *q_out, void *stream) {
    if (int rc = check_bound(env)) return rc;
    const int64_t n = susnet_qnet_packed_floats(env, components, n_components, dims, n_dims);
    if (n < 0) return (int)n;
    if (!packed || !q_out || (reinterpret_cast<uintptr_t>(packed) & 15u)) return fail(SUSNET_E_INVALID, "susnet_qnet_forward: packed (16-byte aligned) / q_out");
    switch (qnet_feat(env, components, n_components)) {
    case FEAT_ONEHOT: return qnet_launch<QRow1>(env, packed, q_out, dims[5], static_cast<hipStream_t>(stream));
    case FEAT_COORD: return qnet_launch<QRowC>(env, packed, q_out, dims[5], static_cast<hipStream_t>(stream));
    default: return qnet_launch<QRow3>(env, packed, q_out, dims[5], static_cast<hipStream_t>(stream));
    }
}

// susnet_qnet_policy_step: the network whose kernel also steps (nullptr: a plain / policy step through k_step)
struct QnetFuse {
    int feat;
    const float *packed;
    float *q_out;
    int n_out;
    // the crew's network (susnet_policy_opts.crew_*), or NULL = a random crew
    const float *crew_packed;
    float *crew_q_out;
    int crew_n_out;
};
template <class ROW, class S>
static int qnet_step_launch(susnet_env *env, bool tape, const QnetFuse &f, const StepArgs &a, const ObsArgs &o, size_t step_lds, hipStream_t st, int n_ticks, const TickStrides &ts) {
    using Q = QNet<ROW>;
    // [table image of the step, shared by the four waves][network image][one region per wave: the rest of a k_step workgroup's LDS]
    // per wave: the teams' greedy actions, then the step's region -- without the group-words area a k_step workgroup of the byte-parallel
    // configurations reserves (step_wave<.., TABLES = false> carves none: nothing stages the action stream here)
    const size_t group_words = HasGroupWords<S>::value ? (size_t)kGroupWords * 4 : 0;
    const size_t rest = step_lds - (size_t)kTableWords * 4 - group_words + (size_t)kStashWords * 4;
    const bool two = tape || f.crew_packed != nullptr; // (both teams: the second network's biases stay resident behind the image)
    const size_t sh = (size_t)kTableWords * 4 + (size_t)(two ? Q::kLdsBytesTwo : Q::kLdsBytes) + 4 * rest;
    if (sh > 160 * 1024) return fail(SUSNET_E_INVALID, "susnet_qnet_policy_step: observation too large for the LDS left beside the network image");
    QStepArgs ka{env->c, env->s, f.packed, f.q_out, f.n_out, a, o, (int)rest, n_ticks, ts, f.crew_packed, f.crew_q_out, f.crew_n_out, 0};
    if (tape) HIP_TRY((qnet_step_launch_k<ROW, S, TapeRng, true>(ka, sh, st))); // (TAPE: both networks, checked by the caller)
    else if (f.crew_packed) HIP_TRY((qnet_step_launch_k<ROW, S, PhiloxRng, true>(ka, sh, st)));
    else HIP_TRY((qnet_step_launch_k<ROW, S, PhiloxRng, false>(ka, sh, st)));
    return SUSNET_OK;
}

// n_ticks > 1 (fuse only: susnet_qnet_policy_rollout): the one-kernel tick repeated inside ONE launch, outputs tick-strided (ts)
static int step_impl(susnet_env *env, const susnet_step_io *io, const float *q_imp, const float *q_crew, void *stream, const QnetFuse *fuse = nullptr,
                     const susnet_policy_opts *opts = nullptr, int n_ticks = 1, const TickStrides *ts = nullptr) {
    if (int rc = check_bound(env)) return rc;
    if (!io || (!io->actions && !q_imp && !fuse)) return fail(SUSNET_E_INVALID, "null actions");
    StepArgs a;
    std::memset(&a, 0, sizeof(a));
    a.actions = io->actions;
    {
        float eps;
        int mask_dead;
        if (int rc = policy_opts(env, opts, eps, mask_dead)) return rc;
        a.epsilon = eps;
        a.mask_dead = mask_dead;
    }
    if (q_imp) {
        a.q_imp = q_imp;
        a.q_crew = q_crew;
        a.n_qi = env->layout.n_actions_imposter;
        a.n_qc = env->layout.n_actions_crew;
        if (a.n_qi > kMaxPolicyActions || a.n_qc > kMaxPolicyActions)
            return fail(SUSNET_E_INVALID, "susnet_policy_step: at most 16 actions per team (use susnet_policy_actions + susnet_step)");
        if (!q_crew && env->cfg.rng_mode != SUSNET_RNG_PHILOX)
            return fail(SUSNET_E_INVALID, "susnet_policy_step: a random crew (q_crew = NULL) draws from the production stream: PHILOX handles only");
    }
    if (io->actions) {
        a.act_dtype = io->actions_dtype;
        if (a.act_dtype != SUSNET_U8 && a.act_dtype != SUSNET_I32 && a.act_dtype != SUSNET_I64)
            return fail(SUSNET_E_INVALID, "actions dtype must be U8/I32/I64");
        if (int rc = strides_for(env, io->actions_layout, a.act_sa, a.act_sb)) return rc;
    } else { // the policy forms with actions == NULL ("not kept"): a zero-initialised susnet_step_io is valid
        a.act_dtype = SUSNET_U8;
        a.act_sa = env->c.B;
        a.act_sb = 1;
    }
    a.rewards.ptr = io->rewards;
    if (io->rewards) {
        if (io->rewards_dtype != SUSNET_F32 && io->rewards_dtype != SUSNET_F64) return fail(SUSNET_E_INVALID, "rewards dtype must be F32/F64");
        a.rewards.f64 = io->rewards_dtype == SUSNET_F64;
        if (int rc = strides_for(env, io->rewards_layout, a.rewards.sa, a.rewards.sb)) return rc;
    }
    a.done = io->done;
    a.trunc = io->truncated;
    a.term_obs = io->term_obs;
    a.roles = io->roles;
    a.raw_F = env->layout.obs_raw_size;
    a.tick = env->ticks;
    ObsArgs o;
    if (int rc = build_obs(env, io->obs, o, env->c.B)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the compiled-in kernels also serve TAPE handles (numpy parity mode): the reference's golden traces run through the
    // same code the production stream uses
    const bool tape = env->cfg.rng_mode == SUSNET_RNG_TAPE;
    const int spec = env->spec;
    size_t sh = lds_bytes(env, o, env->c.auto_reset != 0, spec);
    CHECK_LDS(sh);
    const dim3 g = grid_for(env), blk(kBlock);
    if (fuse) { // the one-kernel policy tick: the compiled-in games whose feature layout the network kernel knows
        if (tape && (!fuse->crew_packed || a.epsilon > 0.0f))
            return fail(SUSNET_E_INVALID, "susnet_qnet_policy_step: a random crew and exploration draw from the production stream: PHILOX handles only "
                                          "(a numpy-tape handle is served with BOTH teams' networks and epsilon = 0: nothing is drawn)");
        if (o.flat_feat != 0) { // the compiled-in feature writer stages 64 bit masks, not the generic writer's byte image
            ObsArgs small = o;
            small.words1 = 64 * 4;
            small.words2 = 0;
            sh = lds_bytes(env, small, env->c.auto_reset != 0, spec);
        }
        sh = (sh + 15) & ~(size_t)15;
        int rc;
        const TickStrides none = {};
        if (spec == 3 && fuse->feat == FEAT_ONEHOT_ALIVE_CLOSEST) rc = qnet_step_launch<QRow3, SpecCfg3>(env, tape, *fuse, a, o, sh, st, n_ticks, ts ? *ts : none);
        else if (spec == 2 && fuse->feat == FEAT_ONEHOT) rc = qnet_step_launch<QRow1, SpecCfg2>(env, tape, *fuse, a, o, sh, st, n_ticks, ts ? *ts : none);
        else if (spec == 2 && fuse->feat == FEAT_COORD) rc = qnet_step_launch<QRowC, SpecCfg2>(env, tape, *fuse, a, o, sh, st, n_ticks, ts ? *ts : none);
        else return fail(env, SUSNET_E_INVALID, "susnet_qnet_policy_step: served are the two compiled-in games (1v1 9x9 ITG, 1v2 14x14 with 4 jobs)");
        if (rc) return rc;
        env->ticks += (uint64_t)n_ticks;
        return SUSNET_OK;
    }
    if (is_family(spec)) kFamily[spec - kFamilySpecBase].step(tape, g, blk, sh, st, env->c, env->s, a, o);
    else switch (spec) {
    case 2: launch_step<SpecCfg2>(tape, g, blk, sh, st, env->c, env->s, a, o); break;
    case 3: launch_step<SpecCfg3>(tape, g, blk, sh, st, env->c, env->s, a, o); break;
    case 4: launch_step<SpecCfg4>(tape, g, blk, sh, st, env->c, env->s, a, o); break;
    case 6: launch_step<SpecTag5>(tape, g, blk, sh, st, env->c, env->s, a, o); break;
    case 12: launch_step<SpecA<2>>(tape, g, blk, sh, st, env->c, env->s, a, o); break;
    default: launch_step<GenericSpec>(tape, g, blk, sh, st, env->c, env->s, a, o); break;
    }
    HIP_TRY(hipGetLastError());
    env->ticks += 1;
    return SUSNET_OK;
}
extern "C" int susnet_step(susnet_env *env, const susnet_step_io *io, void *stream) { return step_impl(env, io, nullptr, nullptr, stream); }
extern "C" int susnet_policy_step(susnet_env *env, const float *q_imposter, const float *q_crew, const susnet_policy_opts *opts, const susnet_step_io *io,
                                  void *stream) {
    if (!q_imposter) return fail(SUSNET_E_INVALID, "susnet_policy_step: null q_imposter");
    if (int rc = no_crew_network(opts, "susnet_policy_step")) return rc;
    return step_impl(env, io, q_imposter, q_crew, stream, nullptr, opts);
}

// the crew's network of susnet_policy_opts (same components as the imposters'): checked, then into the fuse record
static int qnet_crew(const susnet_env *env, const int32_t *components, int32_t n_components, const susnet_policy_opts *opts, QnetFuse &f) {
    f.crew_packed = nullptr;
    f.crew_q_out = nullptr;
    f.crew_n_out = 0;
    if (!opts || !opts->crew_packed) {
        if (opts && (opts->crew_dims || opts->crew_q_out)) return fail(SUSNET_E_INVALID, "susnet_policy_opts: crew_dims / crew_q_out without crew_packed");
        return SUSNET_OK;
    }
    const int64_t n = susnet_qnet_packed_floats(env, components, n_components, opts->crew_dims, opts->crew_n_dims);
    if (n < 0) return (int)n;
    if (reinterpret_cast<uintptr_t>(opts->crew_packed) & 15u) return fail(SUSNET_E_INVALID, "susnet_policy_opts: crew_packed must be 16-byte aligned");
    if (opts->crew_dims[5] != env->layout.n_actions_crew)
        return fail(SUSNET_E_INVALID, "susnet_policy_opts: the crew network's output width must be the crew's action count");
    f.crew_packed = opts->crew_packed;
    f.crew_q_out = opts->crew_q_out;
    f.crew_n_out = opts->crew_dims[5];
    return SUSNET_OK;
}

extern "C" int susnet_qnet_policy_step(susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                                       const float *packed, float *q_out, const susnet_policy_opts *opts, const susnet_step_io *io, void *stream) {
    if (int rc = check_bound(env)) return rc;
    const int64_t n = susnet_qnet_packed_floats(env, components, n_components, dims, n_dims);
    if (n < 0) return (int)n;
    if (!packed || (reinterpret_cast<uintptr_t>(packed) & 15u)) return fail(SUSNET_E_INVALID, "susnet_qnet_policy_step: packed (16-byte aligned)");
    if (dims[5] != env->layout.n_actions_imposter)
        return fail(SUSNET_E_INVALID, "susnet_qnet_policy_step: the network's output width must be the imposters' action count");
    QnetFuse f = {qnet_feat(env, components, n_components), packed, q_out, dims[5], nullptr, nullptr, 0};
    if (int rc = qnet_crew(env, components, n_components, opts, f)) return rc;
    return step_impl(env, io, nullptr, nullptr, stream, &f, opts);
}

extern "C" int susnet_qnet_policy_rollout(susnet_env *env, const int32_t *components, int32_t n_components, const int32_t *dims, int32_t n_dims,
                                          const float *packed, const susnet_policy_opts *opts, const susnet_feed_io *feed, int32_t n_ticks, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!feed || n_ticks < 1) return fail(SUSNET_E_INVALID, "susnet_qnet_policy_rollout: feed is null / n_ticks < 1");
    const int64_t n = susnet_qnet_packed_floats(env, components, n_components, dims, n_dims);
    if (n < 0) return (int)n;
    if (!packed || (reinterpret_cast<uintptr_t>(packed) & 15u)) return fail(SUSNET_E_INVALID, "susnet_qnet_policy_rollout: packed (16-byte aligned)");
    if (dims[5] != env->layout.n_actions_imposter)
        return fail(SUSNET_E_INVALID, "susnet_qnet_policy_rollout: the network's output width must be the imposters' action count");
    if (!env->c.auto_reset) return fail(SUSNET_E_INVALID, "susnet_qnet_policy_rollout: the handle must auto-reset (episodes end inside the launch)");
    const int64_t B = env->c.B, A = env->c.A, S = env->layout.obs_raw_size;
    // (the raw observation is stored in 16-byte pieces: every tick's slot must start on a 16-byte boundary)
    if (feed->obs && n_ticks > 1 && (B * S) % 16 != 0)
        return fail(env, SUSNET_E_INVALID, "susnet_qnet_policy_rollout: feed->obs holds one [B][obs_raw_size] slot per tick, each 16-byte aligned: batch x "
                                           "obs_raw_size must be a multiple of 16 (or pass obs = NULL / one tick per call)");
    susnet_obs_spec obs;
    std::memset(&obs, 0, sizeof(obs));
    obs.mode = SUSNET_OBS_RAW;
    obs.dtype = SUSNET_U8;
    obs.out = feed->obs;
    susnet_step_io io;
    std::memset(&io, 0, sizeof(io));
    io.actions = feed->actions;
    io.actions_dtype = SUSNET_U8;
    io.actions_layout = SUSNET_LAYOUT_BA;
    io.rewards = feed->rewards;
    io.rewards_dtype = SUSNET_F32;
    io.rewards_layout = SUSNET_LAYOUT_BA;
    io.done = feed->done;
    io.truncated = feed->truncated;
    io.obs = feed->obs ? &obs : nullptr;
    io.term_obs = feed->term_obs;
    io.roles = feed->roles;
    QnetFuse f = {qnet_feat(env, components, n_components), packed, feed->q, dims[5], nullptr, nullptr, 0};
    if (int rc = qnet_crew(env, components, n_components, opts, f)) return rc;
    const TickStrides ts = {B * A, B * A * 4, B, B, B * S, B * 2, B * (int64_t)dims[5] * 4, B * (int64_t)f.crew_n_out * 4};
    return step_impl(env, &io, nullptr, nullptr, stream, &f, opts, n_ticks, &ts);
}

extern "C" int susnet_record_layout_of(const susnet_env *env, int32_t record_format, susnet_record_layout_t *out) {
    if (record_format == SUSNET_RECORD_DEFAULT) return susnet_record_layout(env, out);
    if (!env || !out) return fail(SUSNET_E_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    if (record_format != SUSNET_RECORD_COMPACT) return fail(SUSNET_E_INVALID, "susnet_record_layout_of: unknown record format");
    if (env->spec != 2 || !(env->c.duel_fast || env->c.duel_walls)) return SUSNET_OK; // record_bytes = 0: only the 1v1 kernels have the compact record
    out->record_bytes = 16;
    out->off_rewards = 0;
    out->off_obs = 8;
    out->off_actions = out->off_done = out->off_truncated = 14;
    out->flags_packed = 1;
    out->n_obs_segments = 1;
    out->obs_segments[0].off = 8;
    out->obs_segments[0].len = 6;
    return SUSNET_OK;
}

extern "C" int susnet_record_layout(const susnet_env *env, susnet_record_layout_t *out) {
    if (!env || !out) return fail(SUSNET_E_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    const int spec = env->spec; // (TAPE handles too: the reference's numpy streams run through the same kernels)
    const int A = env->c.A, F = env->layout.obs_raw_size;
    if (is_family(spec)) { // FamRecord (susnet_kernels.h): a layout independent of the job count, the observation in segments
        const FamilyEntry &fe = kFamily[spec - kFamilySpecBase];
        const int J = env->c.J;
        out->record_bytes = 4 * fe.record_dwords;
        out->off_rewards = 0;
        out->off_actions = 4 * A;
        const int head_bytes = A + 3 * A + (fe.tagging ? 2 * A + 1 : 0) + 2;
        out->off_done = 4 * A + head_bytes - 2;
        out->off_truncated = out->off_done + 1;
        out->off_obs = 5 * A;
        const int off_cells = 4 * (A + fe.head_dwords);
        int n = 0;
        out->obs_segments[n].off = 5 * A; out->obs_segments[n++].len = 3 * A;            // cells, alive
        if (J > 0 || fe.tagging) {
            out->obs_segments[n].off = off_cells; out->obs_segments[n++].len = 2 * J;    // job cells
            out->obs_segments[n].off = off_cells + 16; out->obs_segments[n++].len = J;   // job status
        }
        if (fe.tagging) { out->obs_segments[n].off = 8 * A; out->obs_segments[n++].len = 2 * A + 1; } // used, counts, steps until the vote
        out->n_obs_segments = n;
        out->planar = 1;
        return SUSNET_OK;
    }
    if (spec != 2 && spec != 3 && spec != 4 && spec != 6) return SUSNET_OK; // record_bytes = 0: no packed mode
    out->off_rewards = 0;
    out->off_actions = 4 * A;
    out->off_done = 5 * A;
    out->off_truncated = 5 * A + 1;
    out->off_obs = 5 * A + 2;
    out->record_bytes = 4 * A + (A + 2 + F + 3) / 4 * 4;
    if (spec != 2) {
        // the byte-parallel kernels (susnet_swar.h / susnet_swar2.h) put the raw row right behind the actions and done / truncated
        // behind the row: rewards | actions | obs | done | truncated | 0-padding.  The row's job cells then start at byte 8A, so
        // the words that hold them are stored as they are; with two lanes per environment (cfg4 at 32 envs per wave) every lane's
        // part of the row is whole dwords
        out->off_obs = 5 * A;
        out->off_done = 5 * A + F;
        out->off_truncated = 5 * A + F + 1;
        out->planar = 1; // stored as planes of 16-byte pieces (susnet_kernels.h store_record_planar)
    }
    out->n_obs_segments = 1;
    out->obs_segments[0].off = out->off_obs;
    out->obs_segments[0].len = F;
    return SUSNET_OK;
}

extern "C" int susnet_set_launch_limit(susnet_env *env, uint64_t bytes) {
    if (!env) return fail(SUSNET_E_INVALID, "null handle");
    if (bytes > (1ull << 31) - 1u) return fail(SUSNET_E_INVALID, "susnet_set_launch_limit: at most 2^31 - 1 bytes (32-bit buffer offsets)");
    env->launch_limit = bytes ? bytes : env->launch_limit_default;
    return SUSNET_OK;
}

extern "C" int susnet_rollout(susnet_env *env, const susnet_rollout_io *io, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!io || io->n_ticks < 1) return fail(SUSNET_E_INVALID, "n_ticks must be >= 1");
    const bool tape = env->cfg.rng_mode == SUSNET_RNG_TAPE;
    RolloutArgs a;
    a.n_ticks = io->n_ticks;
    a.tick_base = env->ticks;
    a.actions = io->actions;
    a.rewards = io->rewards;
    a.done = io->done;
    a.trunc = io->truncated;
    a.record = static_cast<uint8_t *>(io->record);
    a.record_bytes = 0;
    a.term_obs = io->term_obs;
    a.roles = io->roles;
    ObsArgs o;
    if (int rc = build_obs(env, io->obs, o, env->c.B)) return rc;
    const int spec = env->spec;
    if (a.record) {
        if (a.actions || a.rewards || a.done || a.trunc || o.mode != SUSNET_OBS_NONE)
            return fail(env, SUSNET_E_INVALID, "susnet_rollout: record is an alternative to the separate outputs, not an addition");
        susnet_record_layout_t lay;
        if (int rc = susnet_record_layout_of(env, io->record_format, &lay)) return rc;
        if (lay.record_bytes == 0) return fail(env, SUSNET_E_INVALID, "susnet_rollout: this configuration has no packed record of the requested format");
        if ((uintptr_t)a.record % 16) return fail(SUSNET_E_INVALID, "record buffer must be 16-byte aligned");
        a.record_bytes = lay.record_bytes;
    }
    size_t sh = lds_bytes(env, o, true, spec, true);
    CHECK_LDS(sh);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((env->c.B + env->c.epw - 1) / env->c.epw)), blk(kBlock);
    const bool all_traj = a.actions && a.rewards && a.done && a.trunc;
    const bool none_traj = !a.actions && !a.rewards && !a.done && !a.trunc;
    // trajectory mode addresses every output through a buffer descriptor with 32-bit offsets: a launch covers at most
    // as many ticks as keep every output array below 2 GiB; longer requests run as consecutive launches
    const bool traj = all_traj && o.mode == SUSNET_OBS_RAW && o.dtype == SUSNET_U8;
    const bool traj_noobs = all_traj && o.mode == SUSNET_OBS_NONE;
    const uint64_t AB = (uint64_t)env->c.A * (uint64_t)env->c.B;
    // the compiled-in float32 FlatFeaturizer layouts (susnet_flat.h): `onehot_pos` on the 1v1 9x9 game without walls, `onehot_pos +
    // alive_crew + closest_crew` on the 1v2 14x14 game, next to the full trajectory, on the production stream
    const bool flat1 = spec == 2 && (env->c.duel_fast || env->c.duel_walls) && env->c.N == 9 && o.ncomp == 1 && o.comp[0] == SUSNET_F_ONEHOT_POS;
    const bool flat3 = spec == 3 && env->c.N == 14 && o.ncomp == 3 && o.comp[0] == SUSNET_F_ONEHOT_POS && o.comp[1] == SUSNET_F_ALIVE_CREW &&
                       o.comp[2] == SUSNET_F_CLOSEST_CREW;
    const bool traj_flat = all_traj && !tape && o.mode == SUSNET_OBS_FLAT && o.dtype == SUSNET_F32 && (flat1 || flat3) && !a.term_obs && !a.roles;
    const uint64_t obs_tick_bytes = (uint64_t)o.tick_stride * (o.dtype == SUSNET_F32 ? 4u : 1u);
    const uint64_t tick_bytes = a.record ? (uint64_t)env->c.B * (uint64_t)a.record_bytes : std::max<uint64_t>(4u * AB, traj_flat ? obs_tick_bytes : (uint64_t)o.tick_stride);
    // (susnet_set_launch_limit; tests exercise the chunking on small batches).  Records are addressed slab by slab (record_slab): unless a limit
    // was set, a record launch takes any number of ticks
    const uint64_t limit = (a.record && env->launch_limit == env->launch_limit_default && !(env->layout.test_overrides & SUSNET_OVERRIDE_TRAJ_MAX_BYTES))
                               ? ~0ull : env->launch_limit;
    const uint64_t fit = limit / tick_bytes;
    if (a.record && (fit < 1 || tick_bytes > (1ull << 31) - 1u)) return fail(env, SUSNET_E_INVALID, "susnet_rollout: one tick of records exceeds 2 GiB");
    const int out = a.record                                ? (io->record_format == SUSNET_RECORD_COMPACT ? OUT_RECORD16 : OUT_RECORD)
                    : (none_traj && o.mode == SUSNET_OBS_NONE) ? OUT_NONE
                    : (traj && fit >= 1)                    ? OUT_TRAJ_RAW8
                    : (traj_noobs && fit >= 1)              ? OUT_TRAJ
                    : (traj_flat && fit >= 1)               ? OUT_TRAJ_FLAT
                                                            : OUT_ANY;
    const bool rec_term = a.record && a.term_obs && !a.roles && spec == 2 && (env->c.duel_fast || env->c.duel_walls); // whole records of the 1v1 kernel + the terminal states
    if ((a.term_obs || a.roles) && !rec_term && !(all_traj && o.mode == SUSNET_OBS_RAW && o.dtype == SUSNET_U8))
        return fail(env, SUSNET_E_INVALID, "susnet_rollout: term_obs / roles go with the full trajectory and the raw uint8 observation (term_obs also "
                                           "with the records of the 1v1 no-walls kernel)");
    if (tape && out != OUT_TRAJ_RAW8 && out != OUT_RECORD && out != OUT_RECORD16)
        return fail(env, SUSNET_E_INVALID, "susnet_rollout on a TAPE handle stores the full trajectory with the raw uint8 observation, as separate "
                                      "tensors or as packed records (nothing else)");
    const int chunk = (out == OUT_TRAJ_RAW8 || out == OUT_TRAJ || out == OUT_RECORD || out == OUT_RECORD16 || out == OUT_TRAJ_FLAT) ? (int)std::min<uint64_t>((uint64_t)io->n_ticks, fit) : io->n_ticks;
    // the family's 9 .. 12-agent instantiations serve the trajectory / record / state-only modes; anything else (float observations, partial
    // outputs: OUT_ANY) runs the generic kernel on the same state blob and the same streams
    const int run_spec = (is_family(spec) && env->c.A > 8 && out == OUT_ANY) ? 0 : spec;
    if (run_spec != spec) {
        sh = lds_bytes(env, o, true, 0, true);
        CHECK_LDS(sh);
    }
    for (int t0 = 0; t0 < io->n_ticks; t0 += chunk) {
        a.n_ticks = std::min(chunk, io->n_ticks - t0);
        a.tick_base = env->ticks + (uint64_t)t0;
        if (t0 > 0 && a.record) {
            a.record += (uint64_t)chunk * tick_bytes;
            if (a.term_obs) a.term_obs += (uint64_t)chunk * (uint64_t)env->c.B * (uint64_t)env->layout.obs_raw_size;
        }
        if (t0 > 0 && !a.record) { // only reached in the trajectory modes: the four trajectory outputs are bound, [T][B][...] slabs
            a.actions += (uint64_t)chunk * AB;
            a.rewards += (uint64_t)chunk * AB;
            a.done += (uint64_t)chunk * (uint64_t)env->c.B;
            a.trunc += (uint64_t)chunk * (uint64_t)env->c.B;
            if (o.out) o.out = static_cast<uint8_t *>(o.out) + (uint64_t)chunk * (out == OUT_TRAJ_FLAT ? obs_tick_bytes : (uint64_t)o.tick_stride);
            if (a.term_obs) a.term_obs += (uint64_t)chunk * (uint64_t)o.tick_stride;
            if (a.roles) a.roles += (uint64_t)chunk * (uint64_t)env->c.B;
        }
        if (is_family(run_spec)) kFamily[run_spec - kFamilySpecBase].rollout(tape, out, g, blk, sh, st, env->c, env->s, a, o);
        else switch (run_spec) {
        case 2: launch_rollout<SpecCfg2>(tape, out, g, blk, sh, st, env->c, env->s, a, o); break;
        case 3: launch_rollout<SpecCfg3>(tape, out, g, blk, sh, st, env->c, env->s, a, o); break;
        case 4: launch_rollout<SpecCfg4>(tape, out, g, blk, sh, st, env->c, env->s, a, o); break;
        case 6: launch_rollout<SpecTag5>(tape, out, g, blk, sh, st, env->c, env->s, a, o); break;
        case 12: launch_rollout<SpecA<2>>(tape, out, g, blk, sh, st, env->c, env->s, a, o); break;
        default: launch_rollout<GenericSpec>(tape, out, g, blk, sh, st, env->c, env->s, a, o); break;
        }
        HIP_TRY(hipGetLastError());
    }
    env->ticks += (uint64_t)io->n_ticks;
    return SUSNET_OK;
}

// ImposterScentFeaturizer (src/features/component.py:336-380), the one real-valued component: for every living agent other
// than agent 0 ("the imposter", component.py:350), (n - dx) / n and (n - dy) / n -- Python floats, i.e. float64, rounded to
// float32 when added to the float32 accumulator -- go to one of four float32 sums [x > 0, x <= 0, y > 0, y <= 0], in agent order.
__global__ __launch_bounds__(64) void k_scent(Consts c, const void *rows, int dtype, int64_t n_rows, int S_row, float *out) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= n_rows) return;
    const int A = c.A;
    const int64_t base = b * S_row;
    const double n = (double)c.N;
    const int ix = row_value(rows, dtype, base), iy = row_value(rows, dtype, base + 1);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 1; i < A; i++) {
        if (row_value(rows, dtype, base + 2 * A + i) == 0) continue; // alive_agents[i]
        const int dx = row_value(rows, dtype, base + 2 * i) - ix, dy = row_value(rows, dtype, base + 2 * i + 1) - iy;
        const float xs = (float)((n - (double)dx) / n), ys = (float)((n - (double)dy) / n);
        if (xs > 0.0f) acc[0] = __fadd_rn(acc[0], xs);
        else acc[1] = __fadd_rn(acc[1], xs);
        if (ys > 0.0f) acc[2] = __fadd_rn(acc[2], ys);
        else acc[3] = __fadd_rn(acc[3], ys);
    }
    *reinterpret_cast<float4 *>(out + 4 * b) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

extern "C" int susnet_scent(susnet_env *env, const void *rows, int32_t rows_dtype, int64_t n_rows, float *out, void *stream) {
    if (!env || !rows || !out || n_rows < 0) return fail(SUSNET_E_INVALID, "susnet_scent: null argument / negative n_rows");
    if (rows_dtype != SUSNET_U8 && rows_dtype != SUSNET_I32 && rows_dtype != SUSNET_I64 && rows_dtype != SUSNET_F32 && rows_dtype != SUSNET_F64)
        return fail(SUSNET_E_INVALID, "rows dtype must be U8 / I32 / I64 / F32 / F64");
    if ((uintptr_t)out % 16) return fail(SUSNET_E_INVALID, "susnet_scent: out must be 16-byte aligned");
    if (n_rows == 0) return SUSNET_OK;
    hipLaunchKernelGGL(k_scent, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), env->c, rows, (int)rows_dtype,
                       n_rows, (int)env->layout.obs_raw_size, out);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

// ---------------------------------------------------------------------------------------------------
// replay ring (susnet_ring_append)
// ---------------------------------------------------------------------------------------------------
struct RingArgs {
    susnet_ring_io io;
    int64_t B, n0, n1; // envs; first / one-past-last transition (n = tick * B + env) this launch writes
    int32_t A, S, n_imp;
    int32_t rows_per_wave; // 64, or fewer when 64 rows of 2 x trajectory_size x S bytes would not fit the LDS images
    int32_t tile_log2e;    // k_ring_append_tile: log2 of the tile's environments (3: 8 ticks x 8 envs)
    int32_t n0_b;            // n0 = n0_t * B + n0_b; pos_n0 = ring position of transition n0
    int64_t n0_t, pos_n0;
    // the trajectory as PACKED RECORDS (io.record; whole records only: the 1v1 kernels), wave-uniform: record size (0 = separate
    // tensors) and field offsets; rec_packed: actions and flags share one byte (SUSNET_RECORD_COMPACT)
    int32_t rec_bytes, rec_obs, rec_act, rec_rew, rec_done, rec_trunc, rec_packed;
};
// the trajectory's fields at (tick u, env b), from the separate tensors or from the records -- a compile-time choice (REC): as a run-time
// one every access carried a uniform branch and both address forms, and the append of the small 1v1 rows ran at 81 us instead of 56
__device__ __forceinline__ const uint8_t *ring_rec(const RingArgs &r, int64_t u, int64_t b) { return r.io.record + ((size_t)u * r.B + b) * (size_t)r.rec_bytes; }
template <bool REC>
__device__ __forceinline__ uint32_t ring_done(const RingArgs &r, int64_t u, int64_t b) {
    if constexpr (REC) return r.rec_packed ? (ring_rec(r, u, b)[r.rec_done] >> 6) & 1u : (uint32_t)ring_rec(r, u, b)[r.rec_done];
    else return r.io.done[u * r.B + b];
}
template <bool REC>
__device__ __forceinline__ uint32_t ring_trunc(const RingArgs &r, int64_t u, int64_t b) {
    if constexpr (REC) return r.rec_packed ? (uint32_t)(ring_rec(r, u, b)[r.rec_trunc] >> 7) : (uint32_t)ring_rec(r, u, b)[r.rec_trunc];
    else return r.io.truncated[u * r.B + b];
}
template <bool REC>
__device__ __forceinline__ uint32_t ring_action(const RingArgs &r, int64_t t, int64_t b, int i) {
    if constexpr (REC) return r.rec_packed ? (ring_rec(r, t, b)[r.rec_act] >> (3 * i)) & 7u : (uint32_t)ring_rec(r, t, b)[r.rec_act + i];
    else return r.io.actions[((size_t)t * r.B + b) * r.A + i];
}
template <bool REC>
__device__ __forceinline__ float ring_reward(const RingArgs &r, int64_t t, int64_t b, int i) {
    if constexpr (REC) return reinterpret_cast<const float *>(ring_rec(r, t, b) + r.rec_rew)[i];
    else return r.io.rewards[((size_t)t * r.B + b) * r.A + i];
}
// the flattened state an env's window holds at virtual tick u (= the state after tick u; u < 0: the carried-in window)
template <bool REC>
__device__ __forceinline__ const uint8_t *ring_state(const RingArgs &r, int64_t u, int64_t b) {
    const int Tw = r.io.trajectory_size;
    if (u < 0) return r.io.window + ((size_t)b * Tw + (size_t)(Tw + u < 0 ? 0 : Tw + u)) * r.S; // window[Tw - 1] = state before tick 0
    if constexpr (REC) return ring_rec(r, u, b) + r.rec_obs;
    else return r.io.obs + ((size_t)u * r.B + b) * r.S;
}
// One wave per 64 consecutive transitions (32 / 16 / 8 for long windows: RingArgs::rows_per_wave).
// Lane r gathers what its row needs into flat images in LDS, laid out exactly as the wave's 64 rows lie in each ring tensor (row-major;
// `states` and `next_states`: Tw * S bytes per row, the Tw - 1 shared states written to both; actions, rewards, done, imposters
// likewise), and the wave then writes every tensor as ONE linear range: 16 bytes per lane and step, no index arithmetic.  (One image of
// Tw + 1 states per row, read at offsets 0 and S, needs a quarter less LDS but its reads are unaligned dwords: measured 2.1 / 3.4 TB/s
// against 3.0 / 4.2 on the 1v1 and 1v2 shapes.)
// Two things made the first version slow (3.1-3.5 TB/s, 82 % of the wave cycles waiting): every element index was divided by Tw * S to
// find its row, and each lane stored its row's small tensors between its loads -- stores the loads behind them had to wait for
// (may-alias), one memory round trip per element.  Now a lane only LOADS in the gather phase (flags first, unrolled without an early
// exit; then rows, actions, rewards, roles) and all global stores happen after it.
constexpr int kRingFlagsUnroll = 8;
constexpr int kRingGroup = 3, kRingChunk = 8; // source states per load group; dwords of a state per load group
template <bool REC>
__global__ __launch_bounds__(64) void k_ring_append(RingArgs r) {
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x, Tw = r.io.trajectory_size, S = r.S, A = r.A, NI = r.n_imp;
    const int R = r.rows_per_wave;
    const int TS = Tw * S;
    const int img = (R * TS + 15) & ~15; // bytes of one state image (padded: the vector loops read up to 3 bytes past the last row)
    uint8_t *st_img = reinterpret_cast<uint8_t *>(smem), *nx_img = st_img + img;
    float *rew_img = reinterpret_cast<float *>(nx_img + img);        // [R][A]
    uint8_t *act_img = reinterpret_cast<uint8_t *>(rew_img + R * A); // [R][A] (+ pad)
    uint8_t *done_img = act_img + ((R * A + 15) & ~15);              // [R]
    int16_t *imp_img = reinterpret_cast<int16_t *>(done_img + 64);   // [R][NI]
    // (tick, env) of the lane's transition n = n0 + rel + lane without a 64-bit division per lane: the host supplies n0's, the rest is
    // 32-bit (rel + B < 2^32: checked there)
    const uint32_t rel = (uint32_t)blockIdx.x * (uint32_t)R;
    const int64_t n_first = r.n0 + (int64_t)rel;
    const int rows = (int)((r.n1 - n_first) < R ? (r.n1 - n_first) : R);
    if (lane < rows) {
        const uint32_t x = rel + (uint32_t)r.n0_b + (uint32_t)lane, tq = x / (uint32_t)r.B;
        const int64_t t = r.n0_t + (int64_t)tq, b = (int64_t)(x - tq * (uint32_t)r.B);
        // most recent episode boundary before tick t within the window's reach (the episode's first state is obs[e]); all flag
        // loads are independent of each other
        int64_t e = -(1ll << 62);
        if (Tw <= kRingFlagsUnroll) {
            uint32_t fd[kRingFlagsUnroll], ft[kRingFlagsUnroll];
#pragma unroll
            for (int k = 1; k <= kRingFlagsUnroll; k++) { // every lane loads (tick clamped): no per-lane branch, no wait between the loads
                const int64_t u = t - k < 0 ? 0 : t - k;
                const bool want = k <= Tw; // (wave-uniform)
                fd[k - 1] = want ? ring_done<REC>(r, u, b) : 0u;
                ft[k - 1] = want ? ring_trunc<REC>(r, u, b) : 0u;
            }
#pragma unroll
            for (int k = kRingFlagsUnroll; k >= 1; k--)
                if ((fd[k - 1] | ft[k - 1]) != 0u && t - k >= 0) e = t - k; // (descending k: the most recent boundary wins)
        } else {
            for (int64_t u = t - 1; u >= 0 && u > t - 1 - Tw; u--)
                if (ring_done<REC>(r, u, b) | ring_trunc<REC>(r, u, b)) { e = u; break; }
        }
        const uint32_t dn = ring_done<REC>(r, t, b), tr = ring_trunc<REC>(r, t, b);
        const uint32_t role_bits = r.io.roles ? (uint32_t)r.io.roles[t * r.B + b] : ((1u << NI) - 1u);
        uint8_t *my_st = st_img + (size_t)lane * TS, *my_nx = nx_img + (size_t)lane * TS;
        // The row needs Tw + 1 source states (replay_memory.py:108-113, 122-127): the window's Tw states -> states[k], and shifted by one
        // -> next_states[k - 1]; the state after the tick (the terminal observation where the episode ended) -> next_states[Tw - 1].  A
        // state = S consecutive bytes at an arbitrary address, fetched as UNALIGNED dwords (gfx950 serves them, global and LDS alike) + a
        // byte tail.  All loads of a group of kRingGroup states are issued before the first LDS store: one memory round trip per group,
        // not one per dword (the rolled load -> store loop this replaces made 18 dependent round trips per row and left the kernel
        // latency-bound at 3.7 TB/s).
        const uint8_t *nxt = (dn | tr) ? r.io.term_obs + ((size_t)t * r.B + b) * S : ring_state<REC>(r, t, b);
        auto source = [&](int k) -> const uint8_t * {
            if (k >= Tw) return nxt;
            int64_t u = t - Tw + k;
            if (u < e) u = e;
            return ring_state<REC>(r, u, b);
        };
        for (int k0 = 0; k0 <= Tw; k0 += kRingGroup) {
            for (int c0 = 0; c0 < S; c0 += 4 * kRingChunk) {
                uint32_t v[kRingGroup][kRingChunk];
                uint8_t tail[kRingGroup][3];
#pragma unroll
                for (int g = 0; g < kRingGroup; g++) {
                    if (k0 + g > Tw) break; // (wave-uniform)
                    const uint8_t *src = source(k0 + g) + c0;
#pragma unroll
                    for (int q = 0; q < kRingChunk; q++)
                        if (c0 + 4 * q + 4 <= S) __builtin_memcpy(&v[g][q], src + 4 * q, 4);
                    if (S - c0 < 4 * kRingChunk) { // the row ends in this chunk: its last S % 4 bytes
                        const int f0 = (S - c0) & ~3;
#pragma unroll
                        for (int q = 0; q < 3; q++)
                            if (f0 + q < S - c0) tail[g][q] = src[f0 + q];
                    }
                }
#pragma unroll
                for (int g = 0; g < kRingGroup; g++) {
                    const int k = k0 + g;
                    if (k > Tw) break;
                    uint8_t *d0 = k < Tw ? my_st + k * S + c0 : nullptr, *d1 = k > 0 ? my_nx + (k - 1) * S + c0 : nullptr;
#pragma unroll
                    for (int q = 0; q < kRingChunk; q++)
                        if (c0 + 4 * q + 4 <= S) {
                            if (d0) __builtin_memcpy(d0 + 4 * q, &v[g][q], 4);
                            if (d1) __builtin_memcpy(d1 + 4 * q, &v[g][q], 4);
                        }
                    if (S - c0 < 4 * kRingChunk) {
                        const int f0 = (S - c0) & ~3;
#pragma unroll
                        for (int q = 0; q < 3; q++)
                            if (f0 + q < S - c0) {
                                if (d0) d0[f0 + q] = tail[g][q];
                                if (d1) d1[f0 + q] = tail[g][q];
                            }
                    }
                }
            }
        }
        for (int i0 = 0; i0 < A; i0 += 8) { // (loads of eight agents in flight, then their LDS stores)
            uint8_t av[8];
            float rv[8];
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (i0 + q < A) {
                    av[q] = (uint8_t)ring_action<REC>(r, t, b, i0 + q);
                    rv[q] = ring_reward<REC>(r, t, b, i0 + q);
                }
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (i0 + q < A) {
                    act_img[lane * A + i0 + q] = av[q];
                    rew_img[lane * A + i0 + q] = rv[q];
                }
        }
        done_img[lane] = dn ? 1 : 0; // replay_memory.py:131: done, not truncation
        uint32_t m = role_bits;
        for (int k = 0; k < NI; k++) { // ascending agent indices
            const int i = __ffs((int)m) - 1;
            imp_img[lane * NI + k] = (int16_t)(i < 0 ? 0 : i);
            m &= m - 1u;
        }
    }
    wave_lds_fence();
    // ring position of row 0 of this wave; rows are consecutive positions modulo max_size
    int64_t pos0 = r.pos_n0 + (int64_t)rel; // (pos_n0 = (idx + n0) % max_size from the host; rel < max_size)
    if (pos0 >= r.io.max_size) pos0 -= r.io.max_size;
    const int total = rows * TS;
    if (__builtin_expect(pos0 + rows <= r.io.max_size, 1)) { // no wrap inside the wave: every output is ONE contiguous range
        float *out_s = r.io.states + (size_t)pos0 * TS, *out_n = r.io.next_states + (size_t)pos0 * TS;
        if ((((size_t)pos0 * TS) & 3u) == 0) { // 16-byte aligned ranges: four elements per lane and step
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(st_img), *n4 = reinterpret_cast<const uint32_t *>(nx_img);
            for (int g = 4 * lane; g < total; g += 256) {
                const uint32_t a = s4[g >> 2], c = n4[g >> 2];
                const float4 fa = make_float4((float)(a & 0xffu), (float)((a >> 8) & 0xffu), (float)((a >> 16) & 0xffu), (float)(a >> 24));
                const float4 fc = make_float4((float)(c & 0xffu), (float)((c >> 8) & 0xffu), (float)((c >> 16) & 0xffu), (float)(c >> 24));
                if (g + 4 <= total) {
                    *reinterpret_cast<float4 *>(out_s + g) = fa;
                    *reinterpret_cast<float4 *>(out_n + g) = fc;
                } else { // the range's last, partial group
                    const float va[4] = {fa.x, fa.y, fa.z, fa.w}, vc[4] = {fc.x, fc.y, fc.z, fc.w};
                    for (int q = 0; q < total - g; q++) { out_s[g + q] = va[q]; out_n[g + q] = vc[q]; }
                }
            }
        } else {
            for (int g = lane; g < total; g += 64) {
                out_s[g] = (float)st_img[g];
                out_n[g] = (float)nx_img[g];
            }
        }
        int64_t *out_a = r.io.ring_actions + (size_t)pos0 * A;
        float *out_r = r.io.ring_rewards + (size_t)pos0 * A;
        for (int g = lane; g < rows * A; g += 64) {
            out_a[g] = (int64_t)act_img[g];
            out_r[g] = rew_img[g];
        }
        if (lane < rows) r.io.ring_dones[pos0 + lane] = done_img[lane];
        for (int g = lane; g < rows * NI; g += 64) r.io.ring_imposters[(size_t)pos0 * NI + g] = imp_img[g];
    } else { // the ring wraps inside this wave's rows (once per trip round the ring): element by element
        for (int g = lane; g < total; g += 64) {
            const int row = g / TS, k = g - row * TS;
            int64_t p = pos0 + row;
            if (p >= r.io.max_size) p -= r.io.max_size;
            r.io.states[(size_t)p * TS + k] = (float)st_img[g];
            r.io.next_states[(size_t)p * TS + k] = (float)nx_img[g];
        }
        if (lane < rows) {
            int64_t p = pos0 + lane;
            if (p >= r.io.max_size) p -= r.io.max_size;
            for (int i = 0; i < A; i++) {
                r.io.ring_actions[p * A + i] = (int64_t)act_img[lane * A + i];
                r.io.ring_rewards[p * A + i] = rew_img[lane * A + i];
            }
            r.io.ring_dones[p] = done_img[lane];
            for (int k = 0; k < NI; k++) r.io.ring_imposters[p * NI + k] = imp_img[lane * NI + k];
        }
    }
}
// The same rows from a TILE per wave: TT consecutive ticks x TE consecutive environments (TT * TE = 64; lane = dt * TE + db).  A row needs
// the Tw + 1 states around its tick (replay_memory.py:108-113, 122-127) and in the kernel above every lane fetches all of them itself:
// each state of the trajectory is read by Tw + 1 waves.  Here the wave fetches the (TT + Tw) x TE states its tile touches ONCE into an
// LDS image (one state per lane + Tw * TE states ahead of the tile + the terminal states where an episode ended) and every lane then
// assembles its row from that image: (TT + Tw) / TT reads per state instead of Tw + 1.  The rows of one tick are TE consecutive ring
// positions, so the wave writes TT contiguous runs per tensor; run bases live in a small LDS table and the store loop walks all runs as
// one flattened index space (run = index / groups-per-run by a multiply), 16 bytes per lane and step as above.
struct RingRun {
    int64_t pos;      // ring position of the run's first row
    int32_t first, n; // first lane-row of the run in the images (dt * TE + lo); rows (0: nothing to write; < 0: -n rows, the ring wraps inside)
};
__device__ __forceinline__ void ring_copy_state(uint8_t *d0, uint8_t *d1, const uint8_t *src, int S) {
    int c = 0;
    for (; c + 16 <= S; c += 16) {
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_memcpy(&v[q], src + c + 4 * q, 4);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (d0) __builtin_memcpy(d0 + c + 4 * q, &v[q], 4);
            if (d1) __builtin_memcpy(d1 + c + 4 * q, &v[q], 4);
        }
    }
    for (; c + 4 <= S; c += 4) {
        uint32_t v;
        __builtin_memcpy(&v, src + c, 4);
        if (d0) __builtin_memcpy(d0 + c, &v, 4);
        if (d1) __builtin_memcpy(d1 + c, &v, 4);
    }
    for (; c < S; c++) {
        const uint8_t v = src[c];
        if (d0) d0[c] = v;
        if (d1) d1[c] = v;
    }
}
template <bool REC>
__global__ __launch_bounds__(64) void k_ring_append_tile(RingArgs r) {
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x, Tw = r.io.trajectory_size, S = r.S, A = r.A, NI = r.n_imp;
    const int le = r.tile_log2e, TE = 1 << le, TT = 64 >> le;
    const int64_t tiles_b = (r.B + TE - 1) >> le;
    const int64_t tile_t = (int64_t)blockIdx.x / tiles_b, tile_b = (int64_t)blockIdx.x - tile_t * tiles_b;
    const int64_t t_first = r.n0 / r.B; // first tick with a row to write
    const int64_t t0 = t_first + tile_t * TT, b0 = tile_b << le;
    const int dt = lane >> le, db = lane & (TE - 1);
    const int64_t t = t0 + dt, b = b0 + db;
    const bool live = t < r.io.n_ticks && b < r.B; // the lane's (tick, env) exists (its row is written only if t * B + b >= n0)
    const int TS = Tw * S;
    // LDS: [source states (TT + Tw) x TE][their episode-end flags][states image 64 x TS][next_states image][rewards][actions][done][imposters][runs]
    const int n_src = (TT + Tw) * TE;
    uint8_t *src_img = reinterpret_cast<uint8_t *>(smem);
    uint8_t *flg_img = src_img + ((n_src * S + 15) & ~15);
    uint8_t *st_img = flg_img + ((n_src + 15) & ~15);
    const int img = (64 * TS + 15) & ~15;
    uint8_t *nx_img = st_img + img;
    float *rew_img = reinterpret_cast<float *>(nx_img + img);
    uint8_t *act_img = reinterpret_cast<uint8_t *>(rew_img + 64 * A);
    uint8_t *done_img = act_img + ((64 * A + 15) & ~15);
    int16_t *imp_img = reinterpret_cast<int16_t *>(done_img + 64);
    RingRun *runs = reinterpret_cast<RingRun *>(reinterpret_cast<uint8_t *>(imp_img) + ((64 * NI * 2 + 15) & ~15));
    uint8_t *my_st = st_img + (size_t)lane * TS, *my_nx = nx_img + (size_t)lane * TS;

    // ---- phase A: every global load of the tile, then the LDS stores
    uint32_t dn = 0, tr = 0;
    if (live) { dn = ring_done<REC>(r, t, b); tr = ring_trunc<REC>(r, t, b); }
    const bool ended = (dn | tr) != 0u;
    // the Tw ticks ahead of the tile: lane = du * TE + db' for du < Tw (Tw * TE <= 64: checked on the host)
    const int du = lane >> le;
    const int64_t u_pre = t0 - Tw + du;
    const bool pre = du < Tw && b < r.B;
    uint32_t pre_flag = 0;
    if (pre && u_pre >= 0) pre_flag = ring_done<REC>(r, u_pre, b) | ring_trunc<REC>(r, u_pre, b);
    const uint8_t *sp[3] = {live ? ring_state<REC>(r, t, b) : nullptr, pre ? ring_state<REC>(r, u_pre, b) : nullptr,
                            live && ended ? r.io.term_obs + ((size_t)t * r.B + b) * S : nullptr};
    uint8_t *own_slot = src_img + (size_t)((dt + Tw) * TE + db) * S, *pre_slot = src_img + (size_t)(du * TE + db) * S, *last = my_nx + (size_t)(Tw - 1) * S;
    for (int c0 = 0; c0 < S; c0 += 4 * kRingChunk) {
        uint32_t v[3][kRingChunk];
        uint8_t tail[3][3];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            if (sp[g] == nullptr) continue;
            const uint8_t *src = sp[g] + c0;
#pragma unroll
            for (int q = 0; q < kRingChunk; q++)
                if (c0 + 4 * q + 4 <= S) __builtin_memcpy(&v[g][q], src + 4 * q, 4);
            if (S - c0 < 4 * kRingChunk) {
                const int f0 = (S - c0) & ~3;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    if (f0 + q < S - c0) tail[g][q] = src[f0 + q];
            }
        }
#pragma unroll
        for (int g = 0; g < 3; g++) {
            if (sp[g] == nullptr) continue;
            // own state -> its source slot, and the row's last next-state unless the episode ended (then the terminal state is)
            uint8_t *d0 = g == 0 ? own_slot + c0 : g == 1 ? pre_slot + c0 : last + c0;
            uint8_t *d1 = g == 0 && !ended ? last + c0 : nullptr;
#pragma unroll
            for (int q = 0; q < kRingChunk; q++)
                if (c0 + 4 * q + 4 <= S) {
                    __builtin_memcpy(d0 + 4 * q, &v[g][q], 4);
                    if (d1) __builtin_memcpy(d1 + 4 * q, &v[g][q], 4);
                }
            if (S - c0 < 4 * kRingChunk) {
                const int f0 = (S - c0) & ~3;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    if (f0 + q < S - c0) {
                        d0[f0 + q] = tail[g][q];
                        if (d1) d1[f0 + q] = tail[g][q];
                    }
            }
        }
    }
    if (live) {
        const uint32_t role_bits = r.io.roles ? (uint32_t)r.io.roles[t * r.B + b] : ((1u << NI) - 1u);
        for (int i0 = 0; i0 < A; i0 += 8) {
            uint8_t av[8];
            float rv[8];
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (i0 + q < A) {
                    av[q] = (uint8_t)ring_action<REC>(r, t, b, i0 + q);
                    rv[q] = ring_reward<REC>(r, t, b, i0 + q);
                }
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (i0 + q < A) {
                    act_img[lane * A + i0 + q] = av[q];
                    rew_img[lane * A + i0 + q] = rv[q];
                }
        }
        done_img[lane] = dn ? 1 : 0; // replay_memory.py:131: done, not truncation
        uint32_t m = role_bits;
        for (int k = 0; k < NI; k++) { // ascending agent indices
            const int i = __ffs((int)m) - 1;
            imp_img[lane * NI + k] = (int16_t)(i < 0 ? 0 : i);
            m &= m - 1u;
        }
    }
    flg_img[(dt + Tw) * TE + db] = ended ? 1 : 0;
    if (du < Tw) flg_img[du * TE + db] = pre_flag ? 1 : 0;
    if (lane < TT) { // the runs: tick t0 + lane, rows [lo, hi) of the tile's TE environments
        const int64_t tt = t0 + lane;
        RingRun run = {0, 0, 0};
        if (tt < r.io.n_ticks) {
            const int64_t nb = tt * r.B + b0; // transition index of the run's first environment
            const int lo = nb >= r.n0 ? 0 : (r.n0 - nb >= TE ? TE : (int)(r.n0 - nb));
            const int hi = r.B - b0 >= TE ? TE : (int)(r.B - b0);
            if (hi > lo) {
                run.pos = (r.io.idx + nb + lo) % r.io.max_size;
                run.first = lane * TE + lo;
                run.n = run.pos + (hi - lo) <= r.io.max_size ? hi - lo : -(hi - lo);
            }
        }
        runs[lane] = run;
    }
    wave_lds_fence();

    // ---- phase B: the row's window from the source image (states[k] and, shifted by one, next_states[k - 1])
    if (live) {
        int64_t e = -(1ll << 62); // most recent episode boundary before tick t within the window's reach
        for (int k = Tw; k >= 1; k--)
            if (flg_img[(dt + Tw - k) * TE + db] != 0 && t - k >= 0) e = t - k;
        for (int k = 0; k < Tw; k++) {
            int64_t u = t - Tw + k;
            if (u < e) u = e;
            const uint8_t *src = src_img + (size_t)((int)(u - t0 + Tw) * TE + db) * S;
            ring_copy_state(my_st + (size_t)k * S, k > 0 ? my_nx + (size_t)(k - 1) * S : nullptr, src, S);
        }
    }
    wave_lds_fence();

    // ---- phase C: TT contiguous runs per tensor
    const uint32_t run_elems = (uint32_t)(TE * TS); // floats of a full run
    bool fast = true;                              // every run: no wrap inside, 16-byte aligned start, whole groups of four floats
#pragma unroll 1
    for (int q = 0; q < TT; q++) {
        const RingRun run = runs[q];
        if (run.n < 0 || ((((size_t)run.pos * TS) | (size_t)((run.first & (TE - 1)) * TS) | (size_t)(run.n > 0 ? run.n * TS : 0)) & 3u) != 0) fast = false;
    }
    if (__builtin_expect(fast, 1)) {
        const uint32_t gpr = run_elems >> 2; // float4 groups of a full run (a shorter run: the groups past its end are skipped)
        const uint32_t magic = 0xffffffffu / gpr + 1u;
        const uint32_t total = gpr * (uint32_t)TT;
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(st_img), *n4 = reinterpret_cast<const uint32_t *>(nx_img);
        for (uint32_t g = lane; g < total; g += 64) {
            const uint32_t q = __umulhi(g, magic), w = g - q * gpr;
            const RingRun run = runs[q];
            if ((int)(4 * w) >= run.n * TS) continue;
            const uint32_t at = (uint32_t)run.first * (uint32_t)TS + 4 * w; // byte offset into the images (a multiple of 4: checked above)
            const uint32_t a = s4[at >> 2], c = n4[at >> 2];
            const float4 fa = make_float4((float)(a & 0xffu), (float)((a >> 8) & 0xffu), (float)((a >> 16) & 0xffu), (float)(a >> 24));
            const float4 fc = make_float4((float)(c & 0xffu), (float)((c >> 8) & 0xffu), (float)((c >> 16) & 0xffu), (float)(c >> 24));
            const size_t o = (size_t)run.pos * TS + 4 * w;
            *reinterpret_cast<float4 *>(r.io.states + o) = fa;
            *reinterpret_cast<float4 *>(r.io.next_states + o) = fc;
        }
    } else { // a run that wraps round the ring's end or starts off a 16-byte boundary: element by element
        for (int q = 0; q < TT; q++) {
            const RingRun run = runs[q];
            const int n = run.n < 0 ? -run.n : run.n;
            for (int g = lane; g < n * TS; g += 64) {
                const int row = g / TS, k = g - row * TS;
                int64_t p = run.pos + row;
                if (p >= r.io.max_size) p -= r.io.max_size;
                r.io.states[(size_t)p * TS + k] = (float)st_img[(size_t)run.first * TS + g];
                r.io.next_states[(size_t)p * TS + k] = (float)nx_img[(size_t)run.first * TS + g];
            }
        }
    }
    // the small tensors: lane-row l of the images is row l - run.first of run l / TE
    {
        const RingRun run = runs[dt];
        const int n = run.n < 0 ? -run.n : run.n;
        const int row = lane - run.first;
        if (row >= 0 && row < n) {
            int64_t p = run.pos + row;
            if (p >= r.io.max_size) p -= r.io.max_size;
            r.io.ring_dones[p] = done_img[lane];
            for (int k = 0; k < NI; k++) r.io.ring_imposters[p * NI + k] = imp_img[lane * NI + k];
        }
        const uint32_t magic_a = 0xffffffffu / (uint32_t)A + 1u;
        for (uint32_t g = lane; g < 64u * (uint32_t)A; g += 64) { // consecutive lanes: consecutive elements of a run
            const uint32_t l = __umulhi(g, magic_a), i = g - l * (uint32_t)A;
            const RingRun rl = runs[l >> le];
            const int nl = rl.n < 0 ? -rl.n : rl.n, rw = (int)l - rl.first;
            if (rw < 0 || rw >= nl) continue;
            int64_t p = rl.pos + rw;
            if (p >= r.io.max_size) p -= r.io.max_size;
            r.io.ring_actions[p * A + i] = (int64_t)act_img[g];
            r.io.ring_rewards[p * A + i] = rew_img[g];
        }
    }
}
// the carried window of every env after the launch: the window before the tick that follows the last one
template <bool REC>
__global__ __launch_bounds__(64) void k_ring_window(RingArgs r) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= r.B) return;
    const int Tw = r.io.trajectory_size, S = r.S;
    const int64_t t = r.io.n_ticks;
    int64_t e = -(1ll << 62);
    for (int64_t u = t - 1; u >= 0 && u > t - 1 - Tw; u--)
        if (ring_done<REC>(r, u, b) | ring_trunc<REC>(r, u, b)) { e = u; break; }
    // in place, slots ascending: slot k of the new window comes from obs, or (launches shorter than the window) from slot
    // k + t > k of the old one, which has not been overwritten yet
    // (a group's loads are all issued before its first store, as in k_ring_append: a slot read from the old window lies above every slot
    // written so far)
    uint8_t *dst = r.io.window + (size_t)b * Tw * S;
    for (int k0 = 0; k0 < Tw; k0 += kRingGroup) {
        for (int c0 = 0; c0 < S; c0 += 4 * kRingChunk) {
            uint32_t v[kRingGroup][kRingChunk];
            uint8_t tail[kRingGroup][3];
#pragma unroll
            for (int g = 0; g < kRingGroup; g++) {
                if (k0 + g >= Tw) break;
                int64_t u = t - Tw + k0 + g;
                if (u < e) u = e;
                const uint8_t *src = ring_state<REC>(r, u, b) + c0;
#pragma unroll
                for (int q = 0; q < kRingChunk; q++)
                    if (c0 + 4 * q + 4 <= S) __builtin_memcpy(&v[g][q], src + 4 * q, 4);
                if (S - c0 < 4 * kRingChunk) {
                    const int f0 = (S - c0) & ~3;
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        if (f0 + q < S - c0) tail[g][q] = src[f0 + q];
                }
            }
#pragma unroll
            for (int g = 0; g < kRingGroup; g++) {
                if (k0 + g >= Tw) break;
                uint8_t *d = dst + (k0 + g) * S + c0;
#pragma unroll
                for (int q = 0; q < kRingChunk; q++)
                    if (c0 + 4 * q + 4 <= S) __builtin_memcpy(d + 4 * q, &v[g][q], 4);
                if (S - c0 < 4 * kRingChunk) {
                    const int f0 = (S - c0) & ~3;
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        if (f0 + q < S - c0) d[f0 + q] = tail[g][q];
                }
            }
        }
    }
}

extern "C" int susnet_ring_append(susnet_env *env, const susnet_ring_io *io, void *stream) {
    if (!env || !io) return fail(SUSNET_E_INVALID, "null argument");
    if (io->n_ticks < 1 || io->trajectory_size < 1 || io->max_size < 1 || io->idx < 0 || io->idx >= io->max_size)
        return fail(SUSNET_E_INVALID, "susnet_ring_append: n_ticks, trajectory_size, max_size must be positive and 0 <= idx < max_size");
    const bool from_records = io->record != nullptr;
    if (from_records && (io->actions || io->rewards || io->done || io->truncated || io->obs))
        return fail(SUSNET_E_INVALID, "susnet_ring_append: record is an alternative to the separate trajectory tensors, not an addition");
    if ((!from_records && (!io->actions || !io->rewards || !io->done || !io->truncated || !io->obs)) || !io->term_obs || !io->window || !io->states ||
        !io->next_states || !io->ring_actions || !io->ring_rewards || !io->ring_dones || !io->ring_imposters)
        return fail(SUSNET_E_INVALID, "susnet_ring_append: null buffer");
    if (!io->roles && env->c.shuffle_imp) return fail(SUSNET_E_INVALID, "susnet_ring_append: roles are drawn per episode here (shuffle_imposter_index): pass roles");
    RingArgs r;
    r.io = *io;
    r.B = env->c.B;
    r.A = env->c.A;
    r.S = env->layout.obs_raw_size;
    r.n_imp = env->c.n_imp;
    r.rec_bytes = 0;
    if (from_records) { // the trajectory as the packed records a fused rollout wrote (whole records: the 1v1 kernels)
        susnet_record_layout_t lay;
        if (int rc = susnet_record_layout_of(env, io->record_format, &lay)) return rc;
        if (lay.record_bytes == 0 || lay.planar || lay.n_obs_segments != 1)
            return fail(env, SUSNET_E_INVALID, "susnet_ring_append: reads packed records where the handle stores them whole (the 1v1 kernels); the "
                                               "multi-agent kernels' planar records go through the separate trajectory tensors");
        r.rec_bytes = lay.record_bytes; r.rec_obs = lay.off_obs; r.rec_act = lay.off_actions; r.rec_rew = lay.off_rewards;
        r.rec_done = lay.off_done; r.rec_trunc = lay.off_truncated; r.rec_packed = lay.flags_packed;
    }
    const int64_t total = (int64_t)io->n_ticks * r.B;
    r.n0 = total > io->max_size ? total - io->max_size : 0; // (earlier rows would be overwritten by later ones of this same launch)
    r.n1 = total;
    r.n0_t = r.n0 / r.B;
    r.n0_b = (int32_t)(r.n0 % r.B);
    r.pos_n0 = (io->idx + r.n0) % io->max_size;
    if (r.n1 - r.n0 + r.B + 64 >= (1ll << 32)) return fail(SUSNET_E_INVALID, "susnet_ring_append: more than 2^32 rows in one launch");
    // the row images of k_ring_append: states, next_states (bytes), rewards (f32), actions (bytes), done, imposters (i16); a wave
    // takes 64 rows, or 32 / 16 / 8 when the window is long (trajectory_size x S bytes per row, twice): the reference's
    // ReplayBuffer accepts any trajectory_size (replay_memory.py:33-44)
    auto images = [&](size_t R) {
        return 2 * ((R * (size_t)io->trajectory_size * (size_t)r.S + 15) & ~(size_t)15) + R * r.A * 4 + ((R * r.A + 15) & ~(size_t)15) + 64 +
               R * r.n_imp * 2 + 16;
    };
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a (ticks x envs) tile per wave where the window is short enough for the tile's lanes to fetch the Tw ticks ahead of it and the
    // images fit; else 64 (32 / 16 / 8) consecutive rows per wave
    r.rows_per_wave = 64;
    r.tile_log2e = 0;
    const int te = env->ring_tile;
    const size_t Tw = (size_t)io->trajectory_size;
    const size_t tile_lds = te ? ((((size_t)(64 / te) + Tw) * te * r.S + 15) & ~(size_t)15) + ((((size_t)(64 / te) + Tw) * te + 15) & ~(size_t)15) +
                                     2 * ((64 * Tw * r.S + 15) & ~(size_t)15) + 64 * (size_t)r.A * 4 + ((64 * (size_t)r.A + 15) & ~(size_t)15) + 64 +
                                     ((64 * (size_t)r.n_imp * 2 + 15) & ~(size_t)15) + (64 / te) * sizeof(RingRun)
                                : 0;
    if (te && Tw * te <= 64 && tile_lds <= 32 * 1024 && r.A >= 2) {
        r.tile_log2e = te == 8 ? 3 : te == 16 ? 4 : 5;
        const int64_t t_first = r.n0 / r.B, tiles_t = (io->n_ticks - t_first + 64 / te - 1) / (64 / te), tiles_b = (r.B + te - 1) / te;
        if (tiles_t * tiles_b > 0x7fffffffll) return fail(SUSNET_E_INVALID, "susnet_ring_append: too many tiles for one launch");
        if (from_records) hipLaunchKernelGGL(k_ring_append_tile<true>, dim3((unsigned)(tiles_t * tiles_b)), dim3(64), tile_lds, st, r);
        else hipLaunchKernelGGL(k_ring_append_tile<false>, dim3((unsigned)(tiles_t * tiles_b)), dim3(64), tile_lds, st, r);
    } else {
        while (r.rows_per_wave > 8 && images((size_t)r.rows_per_wave) > 64 * 1024) r.rows_per_wave /= 2;
        const size_t sh = images((size_t)r.rows_per_wave);
        if (sh > 64 * 1024) return fail(SUSNET_E_INVALID, "susnet_ring_append: trajectory_size x state size too large (8 rows of the window exceed 64 KiB)");
        const int64_t waves = (r.n1 - r.n0 + r.rows_per_wave - 1) / r.rows_per_wave;
        if (from_records) hipLaunchKernelGGL(k_ring_append<true>, dim3((unsigned)waves), dim3(64), sh, st, r);
        else hipLaunchKernelGGL(k_ring_append<false>, dim3((unsigned)waves), dim3(64), sh, st, r);
    }
    if (from_records) hipLaunchKernelGGL(k_ring_window<true>, dim3((unsigned)((r.B + 63) / 64)), dim3(64), 0, st, r);
    else hipLaunchKernelGGL(k_ring_window<false>, dim3((unsigned)((r.B + 63) / 64)), dim3(64), 0, st, r);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

extern "C" int susnet_observe(susnet_env *env, const susnet_obs_spec *obs, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!obs || obs->mode == SUSNET_OBS_NONE) return fail(SUSNET_E_INVALID, "nothing to observe");
    ObsArgs o;
    if (int rc = build_obs(env, obs, o, env->c.B)) return rc;
    size_t sh = lds_bytes(env, o, false);
    CHECK_LDS(sh);
    hipLaunchKernelGGL(k_observe, grid_for(env), dim3(kBlock), sh, static_cast<hipStream_t>(stream), env->c, env->s, o);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

extern "C" int susnet_featurize(susnet_env *env, const void *rows, int32_t rows_dtype, int64_t n_rows, const susnet_obs_spec *obs,
                                void *stream) {
    if (int rc = check_bound(env)) return rc; // the device error word lives in the bound state blob
    if (!rows || n_rows < 0) return fail(SUSNET_E_INVALID, "rows is null / n_rows negative");
    if (!obs || (obs->mode != SUSNET_OBS_FLAT && obs->mode != SUSNET_OBS_PLANES && obs->mode != SUSNET_OBS_PERSP))
        return fail(SUSNET_E_INVALID, "susnet_featurize: obs mode must be FLAT, PLANES or PERSP");
    if (rows_dtype != SUSNET_U8 && rows_dtype != SUSNET_I32 && rows_dtype != SUSNET_I64 && rows_dtype != SUSNET_F32 &&
        rows_dtype != SUSNET_F64)
        return fail(SUSNET_E_INVALID, "rows dtype must be U8 / I32 / I64 / F32 / F64");
    if (n_rows == 0) return SUSNET_OK;
    ObsArgs o;
    if (int rc = build_obs(env, obs, o, n_rows)) return rc;
    const dim3 g((unsigned)((n_rows + kBlock - 1) / kBlock));
    const Consts &c = env->c;
    if (o.mode == SUSNET_OBS_FLAT && o.dtype == SUSNET_F32 && !env->force_generic) { // the compiled-in layouts (susnet_flat.h)
        using Row1 = FlatRow<FEAT_ONEHOT, 2, 9>;
        using Row3 = FlatRow<FEAT_ONEHOT_ALIVE_CLOSEST, 3, 14>;
        if (c.A == 2 && c.N == 9 && o.ncomp == 1 && o.comp[0] == SUSNET_F_ONEHOT_POS) {
            hipLaunchKernelGGL(k_featurize_flat<Row1>, g, dim3(kBlock), 64 * Row1::MW * 4, static_cast<hipStream_t>(stream), c, rows, (int)rows_dtype,
                               n_rows, (int)env->layout.obs_raw_size, env->s.err, static_cast<float *>(o.out));
            HIP_TRY(hipGetLastError());
            return SUSNET_OK;
        }
        if (c.A == 3 && c.N == 14 && o.ncomp == 3 && o.comp[0] == SUSNET_F_ONEHOT_POS && o.comp[1] == SUSNET_F_ALIVE_CREW && o.comp[2] == SUSNET_F_CLOSEST_CREW) {
            hipLaunchKernelGGL(k_featurize_flat<Row3>, g, dim3(kBlock), 64 * Row3::MW * 4, static_cast<hipStream_t>(stream), c, rows, (int)rows_dtype,
                               n_rows, (int)env->layout.obs_raw_size, env->s.err, static_cast<float *>(o.out));
            HIP_TRY(hipGetLastError());
            return SUSNET_OK;
        }
    }
    size_t sh = lds_bytes(env, o, false);
    CHECK_LDS(sh);
    hipLaunchKernelGGL(k_featurize, g, dim3(kBlock), sh, static_cast<hipStream_t>(stream), env->c, rows, (int)rows_dtype, n_rows,
                       (int)env->layout.obs_raw_size, env->s.err, o);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

extern "C" int susnet_export_state(susnet_env *env, const susnet_state_view *view, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!view) return fail(SUSNET_E_INVALID, "null view");
    hipLaunchKernelGGL(k_export, grid_for(env), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->c, env->s, *view);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

extern "C" int susnet_import_state(susnet_env *env, const susnet_state_view *view, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!view) return fail(SUSNET_E_INVALID, "null view");
    hipLaunchKernelGGL(k_import, grid_for(env), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->c, env->s, *view);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

static void fill_tick(susnet_env *env, uint64_t tick, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_tick, dim3((unsigned)((env->c.Bp + 255) / 256)), dim3(256), 0, st, env->c, env->s, tick);
}

extern "C" int susnet_tick(susnet_env *env, const uint64_t *set, uint64_t *get, void *stream) {
    if (!env) return fail(SUSNET_E_INVALID, "null handle");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (set) {
        env->ticks = *set;
        if (env->c.dev_tick) {
            if (int rc = check_bound(env)) return rc;
            fill_tick(env, env->ticks, st);
            HIP_TRY(hipGetLastError());
        }
    }
    if (get) {
        if (env->c.dev_tick) { // ordered on the caller's stream like every other call of the handle; synchronises it
            if (int rc = check_bound(env)) return rc;
            HIP_TRY(hipMemcpyAsync(&env->ticks, env->s.tickw, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        *get = env->ticks;
    }
    return SUSNET_OK;
}

extern "C" int susnet_device_tick(susnet_env *env, int32_t enable, void *stream) {
    if (int rc = check_bound(env)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (enable && !env->c.dev_tick) {
        fill_tick(env, env->ticks, st);
        HIP_TRY(hipGetLastError());
        env->c.dev_tick = 1;
    } else if (!enable && env->c.dev_tick) {
        HIP_TRY(hipMemcpyAsync(&env->ticks, env->s.tickw, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        env->c.dev_tick = 0;
    }
    return SUSNET_OK;
}

extern "C" int susnet_reduce_lifetime(susnet_env *env, int64_t *out_device, void *stream) {
    if (int rc = check_bound(env)) return rc;
    if (!out_device) return fail(SUSNET_E_INVALID, "null output");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(out_device, 0, sizeof(int64_t) * SUSNET_N_LIFETIME, st));
    const unsigned chunks = (unsigned)((env->c.B + 4095) / 4096);
    hipLaunchKernelGGL(k_reduce_lifetime, dim3(chunks < 64 ? chunks : 64, SUSNET_N_LIFETIME), dim3(256), 0, st, env->c, env->s, out_device);
    HIP_TRY(hipGetLastError());
    return SUSNET_OK;
}

extern "C" int susnet_poll_errors(susnet_env *env, uint32_t *bits_out, void *stream) {
    if (int rc = check_bound(env)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, env->s.err, sizeof(bits), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bits) HIP_TRY(hipMemsetAsync(env->s.err, 0, sizeof(uint32_t), st));
    if (bits_out) *bits_out = bits;
    if (bits & SUSNET_ERRBIT_ASSERT) return fail(SUSNET_E_ACTION_ASSERT, "Invalid action(s): action >= action_space.n");
    if (bits & SUSNET_ERRBIT_INDEX) return fail(SUSNET_E_ACTION_INDEX, "role-invalid action index");
    if (bits & SUSNET_ERRBIT_TAPE) return fail(SUSNET_E_TAPE, "random tape exhausted");
    if (bits & SUSNET_ERRBIT_ROW) return fail(SUSNET_E_ROW, "susnet_featurize: a state row holds a coordinate outside the grid");
    return SUSNET_OK;
}
