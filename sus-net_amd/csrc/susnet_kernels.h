// susnet_kernels.h -- the stepping kernels (k_step, k_rollout) as templates over the compiled-in configuration, plus the
// launchers the host calls.  Every compiled-in configuration is instantiated in a translation unit of its own
// (inst_*.hip: SUSNET_INSTANTIATE), so the library builds in parallel; susnet_capi.hip only declares them.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "susnet_device.h"
#include "susnet_obs.h"
#include "susnet_swar.h"
#include "susnet_swar2.h"
#include "susnet_duel.h"
#include "susnet_flat.h"

namespace susnet {

struct StepArgs {
    const void *actions;
    int32_t act_dtype;
    int64_t act_sa, act_sb; // element strides (agent, env)
    RewardSink rewards;
    uint8_t *done, *trunc;
    uint64_t tick; // steps taken so far (index of the action stream: the shuffled order of this step comes from it)
    // susnet_policy_step: the actions are not an input but the greedy choice of the teams' Q rows (q_imp [B][n_qi], q_crew [B][n_qc] or
    // NULL = the crew's draws from the action stream), made by the stepping lane itself and written to `actions` (NULL: not kept)
    const float *q_imp, *q_crew;
    int32_t n_qi, n_qc;
    float epsilon;     // > 0: every agent explores (takes its uniformly random role-valid draw instead) with this probability
    int32_t mask_dead; // != 0: dead agents are given index 0 (train.py:351-381 sets only the living agents' actions)
    // replay feed of the tick (susnet_step_io): the terminal state's raw row where the episode ended, the acting episode's roles
    uint8_t *term_obs; // [B][F] or NULL
    uint16_t *roles;   // [B] or NULL
    int32_t raw_F;     // flattened_state_size
};
constexpr int kMaxPolicyActions = 16; // Q row lengths susnet_policy_step serves

// what sample_actions_env writes through in a policy step: agent i takes its team's argmax (the crew's draw when it has no network),
// into the per-lane store the step reads and into the caller's action buffer
template <class Store>
struct PolicyStepSink {
    Store &st;
    void *out;
    int32_t dtype;
    int64_t sa, k0;
    uint32_t roles, a_imp, a_crew; // a_crew = ~0u: keep the sampled index
    // epsilon-greedy (train.py:355-381): explore[i] = the exploration stream's word for (tick, agent i) as a uniform number <= epsilon
    float eps;
    uint32_t alive, mask_dead;
    const PhiloxRng *rng;
    ActionStream *xs;
    uint64_t xbase; // tick * A
    __device__ __forceinline__ void set_act(int i, uint32_t sampled) const {
        uint32_t a = ((roles >> i) & 1u) ? a_imp : (a_crew != ~0u ? a_crew : sampled);
        if (eps > 0.0f) {
            const float u = (float)(xs->word(*rng, xbase + (uint64_t)i) >> 8) * 5.9604644775390625e-08f; // [0, 1): 24 bits
            a = u <= eps ? sampled : a;
        }
        if (mask_dead && !((alive >> i) & 1u)) a = 0u;
        st.set_act(i, a);
        if (out) {
            if (dtype == SUSNET_U8) reinterpret_cast<uint8_t *>(out)[(int64_t)i * sa + k0] = (uint8_t)a;
            else if (dtype == SUSNET_I32) reinterpret_cast<int32_t *>(out)[(int64_t)i * sa + k0] = (int32_t)a;
            else reinterpret_cast<int64_t *>(out)[(int64_t)i * sa + k0] = (int64_t)a;
        }
    }
};

struct RolloutArgs {
    int32_t n_ticks;
    uint64_t tick_base; // steps this handle has taken before the launch (index of the action stream)
    uint8_t *actions; // [T][B][A]
    float *rewards;   // [T][B][A]
    uint8_t *done, *trunc; // [T][B]
    uint8_t *record;       // packed mode: [T][B][record_bytes]
    int32_t record_bytes;
    // replay feed (susnet_ring_append): what the fused trajectory does not hold on its own
    uint8_t *term_obs;     // [T][B][F] raw uint8 rows, written ONLY where an episode ended: the true post-step (terminal) state
    uint16_t *roles;       // [T][B] imposter bitmask of the episode that acted at the tick
};

// the raw uint8 observation row of one env at (tick, b) of a [T][B][F] array, byte by byte (rare path: episode ends)
template <int F>
__device__ __forceinline__ void store_row_bytes(uint8_t *base, int64_t tick, int64_t B, int64_t b, const uint8_t *row) {
    uint8_t *p = base + (tick * B + b) * F;
#pragma unroll
    for (int f = 0; f < F; f++) p[f] = row[f];
}

// f(std::integral_constant<int, I>{}) for I = BEGIN .. END - 1
template <int BEGIN, int END, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

template <class RNG>
__device__ __forceinline__ RNG make_rng(const Consts &c, const State &s, int64_t b);

template <>
__device__ __forceinline__ PhiloxRng make_rng<PhiloxRng>(const Consts &c, const State &s, int64_t b) {
    PhiloxRng r;
    r.init(c.seed, c.env_id_base + (uint64_t)b, s.rng[b]);
    return r;
}
template <>
__device__ __forceinline__ TapeRng make_rng<TapeRng>(const Consts &c, const State &s, int64_t b) {
    TapeRng r;
    r.init(s.tape ? s.tape + b * s.tape_len : nullptr, s.tape ? s.tape_len : 0, s.rng[b]);
    return r;
}

__device__ __forceinline__ uint32_t load_action(const void *p, int dtype, int64_t k) {
    int64_t v;
    if (dtype == SUSNET_U8) v = reinterpret_cast<const uint8_t *>(p)[k];
    else if (dtype == SUSNET_I32) v = reinterpret_cast<const int32_t *>(p)[k];
    else v = reinterpret_cast<const int64_t *>(p)[k];
    if (v < 0) v = -1;
    if (v > 0x7ffffff0ll) v = 0x7ffffff0ll;
    return (uint32_t)(int32_t)v;
}

__device__ __forceinline__ void store_action(void *p, int dtype, int64_t k, uint32_t a) {
    if (dtype == SUSNET_U8) reinterpret_cast<uint8_t *>(p)[k] = (uint8_t)a;
    else if (dtype == SUSNET_I32) reinterpret_cast<int32_t *>(p)[k] = (int32_t)a;
    else reinterpret_cast<int64_t *>(p)[k] = (int64_t)a;
}

// All A actions of one env: the dtype test is wave-uniform and made ONCE, so the loads of a branch are issued back to back
// (a per-agent test put a wait between every two of them).
template <class S, class Store>
__device__ __forceinline__ void load_actions(const Consts &c, const StepArgs &a, int64_t b, Store &st) {
    const int A = S::A(c);
    const int64_t k0 = b * a.act_sb;
    auto clamp = [](int64_t v) -> uint32_t {
        if (v < 0) v = -1;
        if (v > 0x7ffffff0ll) v = 0x7ffffff0ll;
        return (uint32_t)(int32_t)v;
    };
    if (a.act_dtype == SUSNET_I64) {
#pragma unroll
        for (int i = 0; i < A; i++) st.set_act(i, clamp(reinterpret_cast<const int64_t *>(a.actions)[(int64_t)i * a.act_sa + k0]));
    } else if (a.act_dtype == SUSNET_I32) {
#pragma unroll
        for (int i = 0; i < A; i++) st.set_act(i, clamp(reinterpret_cast<const int32_t *>(a.actions)[(int64_t)i * a.act_sa + k0]));
    } else {
#pragma unroll
        for (int i = 0; i < A; i++) st.set_act(i, clamp(reinterpret_cast<const uint8_t *>(a.actions)[(int64_t)i * a.act_sa + k0]));
    }
}

template <class RNG>
__device__ __forceinline__ void finish_rng(const State &s, int64_t b, const RNG &rng) {
    s.rng[b] = rng.cur;
    if (rng.ovf) atomicOr(s.err, SUSNET_ERRBIT_TAPE);
}

// One step of the wave's 64 environments b0 .. b0 + 63 (lane tid = environment b0 + tid): the body of k_step, also the tail of the
// one-kernel policy tick (susnet_qnet.h k_qnet_step).  smem: the table image at LDS address 0 (every wave of the workgroup writes the
// same words into it); rest: the wave's own LDS region (lds_bytes() of the host minus the table image).  pre_imp >= 0: the imposters'
// greedy action, already chosen by the caller (the crew then draws from the action stream).
// obs_tick: which [B][F] slot of the observation output this step writes (o.tick_stride elements apart: the multi-tick policy kernel)
// TABLES = false: the caller has filled the table image already (the multi-tick policy kernel: once per launch, before its barrier)
template <class RNG, class S, bool TABLES = true>
// pre_crew >= 0 (with pre_imp): the crew's greedy action too (both teams by their networks: nothing is drawn unless agents explore)
__device__ __forceinline__ void step_wave(const Consts &c, const State &s, const StepArgs &a, const ObsArgs &o, uint32_t *smem, uint32_t *rest, int tid, int64_t b0,
                                          int pre_imp, int64_t obs_tick = 0, int pre_crew = -1) {
    const int64_t b = b0 + tid;
    const bool active = b < c.B;
    typename StoreFor<S>::type st;
    Tables T = carve_lds<S, false, TABLES>(c, smem, rest, tid, st); // (TABLES = false: the policy kernel's wave regions, without a group-words area)
    // One step per launch is latency-bound: every load the step needs is issued here, back to back and without a branch
    // between them (state rows are padded to Bp, so lanes past B read their own padding; their action index is clamped),
    // and only then are the tables written to LDS -- one memory round trip instead of one per table, per agent, per field.
    TableLoad<true> tl;
    if constexpr (TABLES) tl.issue(c, o.comp, tid);
    Env e = {};
    load_env<S>(c, s, st, b, e);
    const bool policy = a.q_imp != nullptr || pre_imp >= 0; // (wave-uniform) susnet_policy_step: Q rows instead of actions
    float qi[kMaxPolicyActions], qc[kMaxPolicyActions];
    if (pre_imp >= 0) {
#pragma unroll
        for (int k = 0; k < kMaxPolicyActions; k++) qi[k] = qc[k] = 0.0f;
    } else if (policy) {
        const int64_t bq = active ? b : 0;
#pragma unroll
        for (int k = 0; k < kMaxPolicyActions; k++) {
            qi[k] = k < a.n_qi ? a.q_imp[bq * a.n_qi + k] : 0.0f;
            qc[k] = (a.q_crew != nullptr && k < a.n_qc) ? a.q_crew[bq * a.n_qc + k] : 0.0f;
        }
    } else {
        load_actions<S>(c, a, active ? b : 0, st);
    }
    RNG rng = make_rng<RNG>(c, s, b);
    // step counter of the action stream: a kernel argument, or (graph-replayable launches) the env's own device word,
    // identical in every env -- read here, advanced below by the same lane
    uint64_t tick_word = a.tick;
    if (c.dev_tick) tick_word = s.tickw[b];
    if constexpr (TABLES) {
        tl.commit(smem, tid, true);
        wave_lds_fence();
    }
    if (active) {
        const int A = S::A(c);
        bool done = false, trunc = false;
        uint32_t bits = 0;
        const uint64_t step_tick = uniform64(tick_word);
        if (policy) { // visualize.py:547-562: every agent takes its team's argmax (first maximum, like torch.argmax)
            const bool crew_net = a.q_crew != nullptr || pre_crew >= 0; // (wave-uniform) the crew acts by a network, not by draws
            uint32_t a_imp = 0, a_crew = crew_net ? 0u : ~0u;
            float hi = qi[0], hc = qc[0];
#pragma unroll
            for (int k = 1; k < kMaxPolicyActions; k++) {
                if (k < a.n_qi && qi[k] > hi) { hi = qi[k]; a_imp = (uint32_t)k; }
                if (a.q_crew != nullptr && k < a.n_qc && qc[k] > hc) { hc = qc[k]; a_crew = (uint32_t)k; }
            }
            if (pre_imp >= 0) a_imp = (uint32_t)pre_imp;
            if (pre_crew >= 0) a_crew = (uint32_t)pre_crew;
            if constexpr (!RNG::kNumpy) {
                ActionStream pas, xs;
                pas.init();
                xs.init(kExploreStreamTag);
                PolicyStepSink<typename StoreFor<S>::type> sink = {st, const_cast<void *>(a.actions), a.act_dtype, a.act_sa, b * a.act_sb, S::imp(c, e.imp),
                                                                   a_imp, a_crew, a.epsilon, e.alive, (uint32_t)a.mask_dead, &rng, &xs, step_tick * (uint64_t)A};
                // the stream's draws are needed by a random crew and by exploration; they are made for EVERY agent (a word's digits
                // depend on the draws before them)
                if (crew_net && !(a.epsilon > 0.0f)) {
                    for (int i = 0; i < A; i++) sink.set_act(i, 0u);
                } else {
                    sample_actions_env<S>(c, sink, e, rng, pas, step_tick);
                }
            } else { // numpy tapes: both teams by their networks, no exploration (the host refuses anything else)
                PolicyStepSink<typename StoreFor<S>::type> sink = {st, const_cast<void *>(a.actions), a.act_dtype, a.act_sa, b * a.act_sb, S::imp(c, e.imp),
                                                                   a_imp, a_crew, 0.0f, e.alive, (uint32_t)a.mask_dead, nullptr, nullptr, 0ull};
                for (int i = 0; i < A; i++) sink.set_act(i, 0u);
            }
        }
        constexpr bool kDuelSpec = !S::kGeneric && S::kA == 2 && S::kJ == 0 && S::kVar == SUSNET_VARIANT_ITG && S::kStaticRoles && S::kFixedOrder;
        bool stepped = false;
        if constexpr (kDuelSpec) {
            if (c.duel_fast) { // no walls, byte-sized rewards: the register-arithmetic step (susnet_duel.h; a wall map steps through the tables here)
                stepped = true;
                const int32_t a0 = (int32_t)st.act(0), a1 = (int32_t)st.act(1);
                if (a0 >= 8 || a1 >= 8) bits |= SUSNET_ERRBIT_ASSERT;            // base.py:360-362
                else if (a0 < 0 || a0 >= 6 || a1 < 0 || a1 >= 5) bits |= SUSNET_ERRBIT_INDEX; // base.py:379-382
                if (bits) {
                    a.rewards.put(0, b, 0.0f);
                    a.rewards.put(1, b, 0.0f);
                } else {
                    const DuelConsts k = make_duel_consts(c);
                    Duel d;
                    to_duel(st, e, d);
                    clear_info_if_fresh(e);
                    rng.align();
                    float r0, r1;
                    uint32_t dn, tr;
                    duel_step<RNG::kNumpy>(k, d, e, rng.cur, (uint32_t)a0, (uint32_t)a1, r0, r1, dn, tr);
                    from_duel(d, st, e);
                    a.rewards.put(0, b, r0);
                    a.rewards.put(1, b, r1);
                    done = dn != 0u;
                    trunc = tr != 0u;
                }
            }
        }
        if (stepped) {
        } else if constexpr (UseSwar<S>::value) {
            // the index-order byte-parallel step (susnet_swar.h): validate like base.py:357-362 / 379-382, then repack
            using W = Swar<S>;
            const uint32_t space_n = 8u + (W::kTag ? (uint32_t)W::A : 0u); // base.py:360-362 / tagging.py:148-150
#pragma unroll
            for (int i = 0; i < A; i++) {
                const int32_t ai = (int32_t)st.act(i);
                if (ai >= (int32_t)space_n) bits |= SUSNET_ERRBIT_ASSERT;
                else if (ai < 0 || (uint32_t)ai >= n_actions<S>(c, (S::imp(c, e.imp) >> i) & 1u)) bits |= SUSNET_ERRBIT_INDEX;
            }
            if (bits) {
                for (int i = 0; i < A; i++) a.rewards.put(i, b, 0.0f);
            } else {
                W w;
                to_swar<S>(c, st, e, w);
                uint32_t act[W::NW], R[W::NW];
#pragma unroll
                for (int q = 0; q < W::NW; q++) act[q] = 0;
#pragma unroll
                for (int i = 0; i < A; i++) act[i / 4] |= (st.act(i) & 0xffu) << (8 * (i & 3));
                clear_info_if_fresh(e);
                identity_ranks<S>(R);
                if (S::order_random(c)) {
                    if constexpr (RNG::kNumpy) {
                        OrderOf<S> ord = (OrderOf<S>)0xFEDCBA9876543210ull;
                        rng.align();
                        shuffle_nibbles<false>(rng, ord, A); // base.py:372-374
                        ranks_from_order<S>(ord, R);
                    } else {
                        ActionStream as;
                        as.init();
                        ranks_from_stream<S>(c, S::imp(c, e.imp), rng, as, step_tick, false, R);
                    }
                }
                float rr[W::A];
                step_swar<S>(c, T, w, e, rng, act, R, rr, done, trunc);
                from_swar<S>(c, w, st, e);
#pragma unroll
                for (int i = 0; i < A; i++) a.rewards.put(i, b, rr[i]);
            }
        } else {
            OrderOf<S> ord = (OrderOf<S>)0xFEDCBA9876543210ull;
            if constexpr (!RNG::kNumpy) {
                if (S::order_random(c)) {
                    ActionStream as;
                    as.init();
                    uint32_t R[rank_words(S::kA)];
                    ranks_from_stream<S>(c, S::imp(c, e.imp), rng, as, step_tick, false, R);
                    order_from_ranks<S>(c, R, ord);
                }
            }
            bits = step_env<S, true, 0>(c, T, st, e, rng, a.rewards, b, done, trunc, nullptr, ord);
        }
        if (bits) atomicOr(s.err, bits);
        else atomicAdd(&s.life[(size_t)SUSNET_L_ENV_STEPS * c.Bp + b], 1u); // (fire and forget: nothing waits for it)
        if (c.dev_tick) s.tickw[b] = step_tick + 1ull;
        if (a.done) a.done[b] = done ? 1 : 0;
        if (a.trunc) a.trunc[b] = trunc ? 1 : 0;
        if (a.roles) a.roles[b] = (uint16_t)S::imp(c, e.imp); // the episode that acted (before an auto-reset draws new roles)
        if (__builtin_expect(a.term_obs != nullptr && (done || trunc), 0)) fill_raw<S>(c, st, e, a.term_obs + b * a.raw_F); // its true terminal state
        bool jobs_changed = false;
        if (__builtin_expect(c.auto_reset && (done || trunc), 0)) {
            accumulate_lifetime(c, s, b, e, trunc);
            reset_env<S>(c, T, st, tid, e, rng);
            e.flags |= FLAG_FRESH; // info counters stay readable until the next step
            jobs_changed = true;
        }
        store_env<S>(c, s, st, b, e, jobs_changed);
        finish_rng(s, b, rng);
    }
    int nrows = (int)((c.B - b0) < kBlock ? (c.B - b0) : kBlock);
    if constexpr (FlatFor<S>::kOk) { // the policy loop's observation: the compiled-in float32 feature row (susnet_flat.h)
        if (o.flat_feat == FlatFor<S>::kFeat) {
            using Row = typename FlatFor<S>::Row;
            Row row;
            row.clear();
            if (active) {
                uint32_t fx[Row::A], fy[Row::A], fal[Row::A];
#pragma unroll
                for (int i = 0; i < Row::A; i++) {
                    fx[i] = st.xy(i) & 15u;
                    fy[i] = st.xy(i) >> 4;
                    fal[i] = (e.alive >> i) & 1u;
                }
                row.build(fx, fy, fal);
            }
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(o.out) + obs_tick * o.tick_stride + b0 * Row::F, 0, nrows * Row::F * 4, 0x00020000);
            flat_store_wave(row, T.stage, tid, nrows, r, 0u);
            return;
        }
    }
    write_obs<S>(c, o, T, st, tid, e, active, b0, nrows, obs_tick);
}

template <class RNG, class S>
__global__ __launch_bounds__(kBlock) void k_step(Consts c, State s, StepArgs a, ObsArgs o) {
    extern __shared__ uint32_t smem[];
    warm_kernargs<sizeof(Consts) + sizeof(State) + sizeof(StepArgs) + sizeof(ObsArgs)>();
    step_wave<RNG, S>(c, s, a, o, smem, smem + kTableWords, (int)threadIdx.x, (int64_t)blockIdx.x * kBlock, -1);
}

// Fused random rollout: n_ticks x {sample_actions; step; auto-reset} with the state held on chip.
// OUT selects what a tick stores, at compile time (run-time optional outputs cost uniform branches on a path
// that is instruction-fetch bound): OUT_ANY = whatever pointers are non-null / any observation mode;
// OUT_NONE = nothing (state-only fast-forward); OUT_TRAJ_RAW8 = actions + rewards + done + truncated + the raw
// uint8 observation (the populate()-shaped record)
enum : int { OUT_ANY = 0, OUT_NONE = 1, OUT_TRAJ_RAW8 = 2, OUT_TRAJ = 3, OUT_RECORD = 4, OUT_TRAJ_FLAT = 5, OUT_RECORD16 = 6 };
// OUT_RECORD16 (the 1v1 no-walls kernel only): the COMPACT record -- 16 bytes, one full-line store per env-step: rewards f32[2] | x0 y0 x1 y1
// alive0 alive1 (the raw row, whole bytes) | one byte a0 | a1 << 3 | done << 6 | truncated << 7 | 0 (SUSNET_RECORD_COMPACT)
// OUT_TRAJ_FLAT = OUT_TRAJ + the float32 FlatFeaturizer row of the configuration's compiled-in layout (susnet_flat.h)
// the replay feed (term_obs, roles) only exists next to the full trajectory with the raw uint8 observation (susnet_rollout refuses
// it elsewhere): the other instantiations carry neither the two pointers nor their per-tick tests
__host__ __device__ constexpr bool kFeed(int out) { return out == OUT_TRAJ_RAW8; }
// OUT_TRAJ = OUT_TRAJ_RAW8 without an observation; OUT_RECORD = the OUT_TRAJ_RAW8 fields packed into ONE record per
// env-step (rewards f32[A] | actions u8[A] | done | truncated | raw obs u8[F], padded to a dword): a lane stores its
// record with one or two wide stores through a single buffer descriptor instead of six stores through five
// The record array of a launch is addressed TICK BY TICK: the buffer descriptor's base moves to the tick's slab (a 64-bit scalar add), its
// size is one slab, and every per-lane offset is an offset inside the slab -- so a launch is not limited to the 2 GiB a 32-bit offset from
// ONE base could span (tag5 at 65 536 envs used to need two launches for 512 ticks), and the 128-bit stores need no tick offset added
// on the vector unit (BufDst::st128).
__device__ __forceinline__ BufDst record_slab(uint8_t *record, uint32_t slab_bytes, int tick, uint32_t lane_off) {
    return make_buf_dst(record + (uint64_t)(uint32_t)tick * (uint64_t)slab_bytes, (uint64_t)slab_bytes, lane_off);
}
template <class S>
struct RecordLayout {
    static constexpr int kBytesTail = (S::kA > 0 ? S::kA : 0) + 2 + (S::kRawF > 0 ? S::kRawF : 0);
    static constexpr int kDwords = (S::kA > 0 ? S::kA : 0) + (kBytesTail + 3) / 4;
};
// The packed record of the byte-parallel FAMILY (job count read at run time: susnet_swar.h): a layout that does not depend on the job
// count -- rewards f32[A] | actions u8[A] | cells 2A | alive A | [tagging: used A | counts A | steps until the vote] | done | truncated |
// 0-padding to a dword | EIGHT job-cell slots (16 bytes) | EIGHT job-status slots (8 bytes) -- so that it is assembled from registers
// with static byte moves and stored as planes like the compiled-in records.  The raw observation (flatten_state order) is the
// concatenation of up to four SEGMENTS of it (susnet_record_layout_t::obs_segments): cells + alive, 2J job-cell bytes, J status bytes,
// the tagging tail.
template <class S>
struct FamRecord {
    static constexpr int A = S::kA > 0 ? S::kA : 1;
    static constexpr bool kTag = S::kVar == SUSNET_VARIANT_TAGGING;
    static constexpr int kHeadBytes = A + 3 * A + (kTag ? 2 * A + 1 : 0) + 2;
    static constexpr int kHeadDwords = (kHeadBytes + 3) / 4;
    static constexpr int kDwords = A + kHeadDwords + 4 + 2;
    static constexpr int kOffActions = 4 * A, kOffCells = 5 * A, kOffTag = 8 * A, kOffDone = 4 * A + kHeadBytes - 2,
                         kOffJobCells = 4 * (A + kHeadDwords), kOffJobStatus = kOffJobCells + 16;
};
template <class S, int OUT, class RNG = PhiloxRng>
__global__ __launch_bounds__(kBlock) void k_rollout(Consts c, State s, RolloutArgs a, ObsArgs o) {
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    // XCD-aware wave -> environment-block mapping: workgroup ids are dealt round-robin to the 8 XCDs, each with its
    // own L2.  Consecutive env blocks write adjacent pieces of the same 128-byte lines (a wave's 64 done flags are
    // half a line), so they are given to workgroups of the SAME XCD, dispatched back to back: the halves meet in one
    // L2 instead of leaving two XCDs as partial-line write-backs.
    const uint32_t nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t per = nblk >> 3, rem = nblk & 7u;
    const uint32_t wave_id = xcd * per + (xcd < rem ? xcd : rem) + slot;
    const int64_t b0 = (int64_t)wave_id * c.epw, b = b0 + tid;
    const bool active = tid < c.epw && b < c.B;
    const int nrows = (int)((c.B - b0) < c.epw ? (c.B - b0) : c.epw);
    const int A = S::A(c);
    typename StoreFor<S>::type st;
    Tables T = setup_lds<S>(c, smem, tid, st);
    if (tid < 16) T.comp[tid] = (uint32_t)o.comp[tid];
    wave_lds_fence();
    Env e = {};
    RNG rng = make_rng<RNG>(c, s, active ? b : 0);
    ActionStream as;
    as.init();
    if (active) load_env<S>(c, s, st, b, e);
    // trajectory mode with register-direct observation rows has no cooperative (all-lane) work past this point:
    // idle lanes of a ragged last wave leave, and the tick loop runs without per-block exec masking
    if (((OUT == OUT_TRAJ_RAW8 && S::kRawF > 0) || OUT == OUT_TRAJ || OUT == OUT_RECORD) && !active) return;
    // the step counter of the action stream: a kernel argument, or (graph-replayable launches) a device word that the
    // last workgroup to finish advances; either way wave-uniform, in scalar registers
    uint64_t tick_base = a.tick_base;
    if (c.dev_tick) tick_base = uniform64(s.tickw[active ? b : b0]); // (every env holds the same count)
    LifeAcc life;
    life.clear();
    const int64_t AB = (int64_t)A * c.B;
    // per-lane output cursors, bumped once per tick (no 64-bit index arithmetic per store)
    const int64_t bb = active ? b : 0;
    uint8_t *pa = a.actions ? a.actions + bb * A : nullptr; // env-major rows: one lane writes its A values with wide stores
    float *pr = a.rewards ? a.rewards + bb * A : nullptr;
    uint8_t *pd = a.done ? a.done + bb : nullptr;
    uint8_t *pt = a.trunc ? a.trunc + bb : nullptr;
    constexpr int kRawF = S::kRawF;
    // compiled-in configurations write their raw uint8 row straight from registers (a few 4-byte stores per lane:
    // the 64 rows of a wave are contiguous, so every touched line is fully written within the tick); measured
    // faster than the LDS image + 16-byte copy-out, also when that copy-out was software-pipelined across ticks
    constexpr bool kDirect = (OUT == OUT_TRAJ_RAW8) && S::kRawF > 0;
    // OUT_TRAJ_RAW8: every output goes through a buffer descriptor (wave-uniform base and size, hardware range
    // check) with a fixed per-lane byte offset and the tick's slab offset in a scalar register, so a store costs no
    // vector address arithmetic and a tick advances five scalar offsets.  The host only selects this mode when
    // every output array of the launch is below 2 GiB.
    constexpr bool kTraj = OUT == OUT_TRAJ_RAW8 || OUT == OUT_TRAJ;
    const uint64_t nt = (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    BufDst da = make_buf_dst(a.actions, nt * (uint64_t)AB, (uint32_t)(bb * A));
    BufDst dr = make_buf_dst(a.rewards, nt * (uint64_t)AB * 4u, (uint32_t)(bb * A) * 4u);
    BufDst dd = make_buf_dst(a.done, nt * (uint64_t)c.B, (uint32_t)bb);
    BufDst dt = make_buf_dst(a.trunc, nt * (uint64_t)c.B, (uint32_t)bb);
    BufDst dobs = make_buf_dst(o.out, nt * (uint64_t)o.tick_stride, (uint32_t)(bb * (kRawF > 0 ? kRawF : 0)));
    const uint32_t slab_a = (uint32_t)AB, slab_d = (uint32_t)c.B, slab_o = (uint32_t)o.tick_stride;
    constexpr bool kRec = OUT == OUT_RECORD; // (host: compiled-in configurations with a static raw row only)
    constexpr int kRecDwords = RecordLayout<S>::kDwords > 0 ? RecordLayout<S>::kDwords : 1;
    BufDst drec = make_buf_dst(a.record, nt * (uint64_t)c.B * (uint64_t)a.record_bytes, (uint32_t)(bb * a.record_bytes));
    const uint32_t slab_rec = (uint32_t)c.B * (uint32_t)a.record_bytes;
    if (kTraj) { pa = nullptr; pr = nullptr; pd = nullptr; pt = nullptr; }
#ifdef SUSNET_STAMPS // diagnostic build only (tools/stamps.py): cycle shares of the tick's segments, one wave
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    unsigned long long seg2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(k) do { unsigned long long tn = __builtin_readcyclecounter(); seg[k] += tn - tprev; tprev = tn; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    if (active && a.n_ticks > 0) clear_info_if_fresh(e); // once per launch instead of once per tick
    // one tick; PAR = compile-time position of the absolute tick in its 4-tick action-stream block, or -1 = run time
    auto tick_body = [&](int tick, auto par) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
#ifdef SUSNET_STAMPS
        tprev = __builtin_readcyclecounter();
#endif
        STAMP(0);
        // this tick's slabs: scalar offsets derived from the (wave-uniform) tick index -- nothing loop-carried (loop-carried
        // scalar offsets ended up in vector registers, every store inside a waterfall loop)
        if (kTraj) {
            const uint32_t t32 = (uint32_t)tick;
            da.so = t32 * slab_a; dr.so = t32 * (4u * slab_a); dd.so = t32 * slab_d; dt.so = t32 * slab_d; dobs.so = t32 * slab_o;
        }
        if (kRec) drec = record_slab(a.record, slab_rec, tick, (uint32_t)(bb * a.record_bytes));
        if (active) {
            sample_actions_env<S, PAR>(c, st, e, rng, as, tick_base + (uint64_t)tick);
            // shuffled order: the tick's shuffle draws follow its action draws in the stream
            OrderOf<S> ow = (OrderOf<S>)0xFEDCBA9876543210ull;
            if constexpr (!RNG::kNumpy) { // (numpy parity: step_env shuffles from the env's own words)
                if (S::order_random(c)) {
                    uint32_t R[rank_words(S::kA)];
                    ranks_from_stream<S, PAR>(c, S::imp(c, e.imp), rng, as, tick_base + (uint64_t)tick, true, R);
                    order_from_ranks<S>(c, R, ow);
                }
            }
            STAMP(1);
            uint32_t av[S::kA > 0 ? S::kA : 1];
            float rr[S::kA > 0 ? S::kA : 1];
            if (kRec) {
#pragma unroll
                for (int i = 0; i < (S::kA > 0 ? S::kA : 0); i++) av[i] = st.act(i);
            } else if (kTraj || (OUT == OUT_ANY && pa != nullptr)) {
                if (!S::kGeneric) {
#pragma unroll
                    for (int i = 0; i < A; i++) av[i] = st.act(i);
                    if (kTraj) store_row_u8<(S::kA > 0 ? S::kA : 1)>(da, av);
                    else store_row_u8<(S::kA > 0 ? S::kA : 1)>(PtrDst{pa}, av);
                } else if (kTraj) {
                    for (int i = 0; i < A; i++) da.st8((uint32_t)i, st.act(i));
                } else {
                    for (int i = 0; i < A; i++) pa[i] = (uint8_t)st.act(i);
                }
            }
            STAMP(2);
            RewardRowSink sink{{(OUT == OUT_ANY && pr) ? (void *)pr : nullptr, 1, 0, 0}, dr, rr};
            bool done, trunc;
#ifdef SUSNET_STAMPS
            if (kRec) step_env<S, false, 3, false>(c, T, st, e, rng, sink, 0, done, trunc, seg2, ow);
            else if (kTraj) step_env<S, false, 2, false>(c, T, st, e, rng, sink, 0, done, trunc, seg2, ow);
            else step_env<S, false, 0, false>(c, T, st, e, rng, sink, 0, done, trunc, seg2, ow);
#else
            if (kRec) step_env<S, false, 3, false>(c, T, st, e, rng, sink, 0, done, trunc, nullptr, ow);
            else if (kTraj) step_env<S, false, 2, false>(c, T, st, e, rng, sink, 0, done, trunc, nullptr, ow);
            else step_env<S, false, 0, false>(c, T, st, e, rng, sink, 0, done, trunc, nullptr, ow);
#endif
            STAMP(3);
            if (kTraj) {
                dd.st8(0u, done ? 1u : 0u);
                dt.st8(0u, trunc ? 1u : 0u);
            } else if (OUT == OUT_ANY) {
                if (pd != nullptr) *pd = done ? 1 : 0;
                if (pt != nullptr) *pt = trunc ? 1 : 0;
            }
            STAMP(4);
            if (kFeed(OUT) && a.roles != nullptr) a.roles[(int64_t)tick * c.B + b] = (uint16_t)S::imp(c, e.imp);
            if (__builtin_expect(done || trunc, 0)) {
                life.add_episode(e, trunc);
                if (kFeed(OUT) && a.term_obs != nullptr) { // the terminal state, before the in-launch reset replaces it
                    const int F = o.F;
                    uint8_t *p = a.term_obs + ((int64_t)tick * c.B + b) * F;
                    if constexpr (S::kRawF > 0) {
                        uint8_t row[S::kRawF + 4];
                        fill_raw<S>(c, st, e, row);
#pragma unroll
                        for (int f = 0; f < S::kRawF; f++) p[f] = row[f];
                    } else {
                        fill_raw<S>(c, st, e, p);
                    }
                }
                reset_env<S>(c, T, st, tid, e, rng);
                // info counters of a terminal step stay readable until the next step: only the launch's last
                // tick can be observed, every other episode end zeroes them right away (wave-uniform branch)
                if (tick == a.n_ticks - 1) e.flags |= FLAG_FRESH;
                else zero_metrics(e);
            }
            STAMP(5);
            if (kRec) { // one packed record per env-step: rewards | actions | done | truncated | raw observation
                constexpr int kA = S::kA > 0 ? S::kA : 0, kF = kRawF > 0 ? kRawF : 0, kNB = kA + 2 + kF;
                uint8_t row[kF + 4];
                fill_raw<S>(c, st, e, row);
                uint32_t by[(kNB + 3) / 4 * 4];
#pragma unroll
                for (int k = 0; k < kA; k++) by[k] = av[k] & 0xffu;
                by[kA] = done ? 1u : 0u;
                by[kA + 1] = trunc ? 1u : 0u;
#pragma unroll
                for (int f = 0; f < kF; f++) by[kA + 2 + f] = row[f];
#pragma unroll
                for (int k = kNB; k < (kNB + 3) / 4 * 4; k++) by[k] = 0u;
                uint32_t w[kRecDwords];
#pragma unroll
                for (int i = 0; i < kA; i++) w[i] = __float_as_uint(rr[i]);
#pragma unroll
                for (int k = 0; k < (kNB + 3) / 4; k++)
                    w[kA + k] = by[4 * k] | (by[4 * k + 1] << 8) | (by[4 * k + 2] << 16) | (by[4 * k + 3] << 24);
                store_dwords<kRecDwords>(drec, w);
            }
            if (OUT == OUT_ANY) {
                pa = pa ? pa + AB : pa;
                pr = pr ? pr + AB : pr;
                pd = pd ? pd + c.B : pd;
                pt = pt ? pt + c.B : pt;
            }
        }
        if (OUT == OUT_ANY || (OUT == OUT_TRAJ_RAW8 && !kDirect)) {
            int tid_o = tid; // (an opaque copy: see k_rollout_swar)
            asm volatile("" : "+v"(tid_o));
            if (OUT == OUT_ANY) write_obs<S>(c, o, T, st, tid_o, e, active, b0, nrows, tick);
            else write_obs_raw8<S>(c, o, T, st, tid_o, e, active, b0, nrows, tick);
        }
        if (kDirect) {
            if (active) {
                uint8_t row[(kRawF > 0 ? kRawF : 1) + 4];
                fill_raw<S>(c, st, e, row);
                // 4-byte stores (naturally aligned when F % 4 == 0, otherwise the hardware's unaligned
                // access splits them), then a 2-byte and a 1-byte tail
                uint32_t rv[(kRawF > 0 ? kRawF : 1)];
#pragma unroll
                for (int k = 0; k < kRawF; k++) rv[k] = row[k];
                store_row_u8<(kRawF > 0 ? kRawF : 1)>(dobs, rv);
            }
        }
        STAMP(6);
    };
    // The action stream is walked in GROUPS of ticks that start on a Philox block boundary: 4 ticks (4 * W words) when the
    // word layout is compiled in, 12 for the 1v1 game (three ticks per word).  Inside a group every block generation and
    // word selection is static; ticks before the first group boundary of a launch and after its last full group run
    // through the run-time flavour.
    // (only fully compiled-in configurations: SpecA<2> -- two agents, everything else read at run time -- unrolled twelve ticks
    // of its run-time-checked step into 35 000 instructions with 782 spilled SGPRs; it walks the stream tick by tick now)
    constexpr int kGroup = (S::kGeneric || OUT == OUT_ANY || RNG::kNumpy || !S::kStaticAw) ? 0 : (S::kA == 2 ? 4 * kDuelTicksPerWord : 4);
    int tick = 0;
    while (tick < a.n_ticks) {
        if (kGroup > 0 && tick + kGroup <= a.n_ticks && ((tick_base + (uint64_t)tick) % (uint64_t)(kGroup > 0 ? kGroup : 1)) == 0ull) {
            static_for<0, (kGroup > 0 ? kGroup : 1)>([&](auto pos) __attribute__((always_inline)) { tick_body(tick + decltype(pos)::value, pos); });
            tick += kGroup;
        } else {
            tick_body(tick, std::integral_constant<int, -1>{});
            tick++;
        }
    }
#ifdef SUSNET_STAMPS
    if (blockIdx.x == 100 && tid == 0)
        for (int k = 0; k < 8; k++) {
            atomicAdd(reinterpret_cast<unsigned long long *>(s.err) + 2 + k, seg[k]);
            atomicAdd(reinterpret_cast<unsigned long long *>(s.err) + 10 + k, seg2[k]);
        }
#endif
    if (active) {
        store_env<S>(c, s, st, b, e, true);
        finish_rng(s, b, rng);
        life.flush(c, s, b, (uint32_t)(a.n_ticks > 0 ? a.n_ticks : 0));
        if (c.dev_tick) s.tickw[b] = tick_base + (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    }
}

// N consecutive bytes (packed four to a word) at a row destination: widest naturally-splitting stores
template <int N, class D>
__device__ __forceinline__ void store_packed_bytes(const D &d, const uint32_t *w) {
    constexpr int kW = N / 4;
    if (kW >= 4) {
#pragma unroll
        for (int k = 0; k + 4 <= kW; k += 4) d.st128(4u * k, w[k], w[k + 1], w[k + 2], w[k + 3]);
    }
    constexpr int k4 = kW / 4 * 4;
    if (kW - k4 >= 2) d.st64(4u * k4, w[k4], w[k4 + 1]);
    if ((kW - k4) & 1) d.st32(4u * (kW - 1), w[kW - 1]);
    if ((N & 3) >= 2) d.st16(4u * kW, w[kW] & 0xffffu);
    if (N & 1) d.st8((uint32_t)(N - 1), (w[kW] >> (8 * ((N & 3) - 1))) & 0xffu);
}

// ---- the next episode, drawn ahead ------------------------------------------------------------------------------------------------
// An in-launch reset used to send the WHOLE wave through the reset path -- Philox blocks, placement draws with their rejection rules,
// the episode constants of the byte-parallel form -- whenever ONE of its environments finished: every tenth tick for cfg3, every third
// for the tagging game, a quarter of the rollouts' cycles with one or two lanes doing useful work.  On the production stream a reset's
// draws are a function of (seed, env, episode index) alone (susnet_device.h ResetStream), so a lane can hold its NEXT episode's
// initial state ready: a finishing lane just takes it (a dozen register moves), and the expensive pass runs only when a finishing
// lane has none -- and then draws for EVERY lane of the wave that has none, at full lane efficiency (about one pass per ten resets).
// Numpy tapes (parity mode) keep the inline reset: their draws depend on the words consumed so far.
template <class S, class Store>
__device__ __forceinline__ void draw_episode(const Consts &c, const Tables &T, int tid, const PhiloxRng &rng, uint32_t episode, Store &stn, Env &en) {
    en = Env{};
    en.ep = episode;
    reset_env<S>(c, T, stn, tid, en, const_cast<PhiloxRng &>(rng)); // (Philox flavour: reads the key and the env id only)
}

// ---- the packed record of the byte-parallel kernels, stored as PLANES --------------------------------------------------------------
// A record of R bytes per env-step stored as [tick][env][R] makes every store instruction of a wave touch 64 pieces R bytes apart:
// 40 partially written cache lines per instruction at R = 40, and the write path (TCP -> L2 requests) saturates at 4.4 TB/s of such
// stores with no arithmetic at all (tools/store_patterns.hip aos40: 301 us for cfg3's launch; aos80: 3.6 TB/s) -- which is where the
// rollouts had arrived once their instruction count was cut.  The same dwords stored as planes of 16-byte pieces --
// [tick][piece][env][16 bytes], the last piece 8 and / or 4 bytes wide when R is not a multiple of 16 -- make every store instruction
// write 1 KiB of consecutive bytes (soa40: 5.4 TB/s; cfg3 96 -> 108 G env-steps/s on the same box with nothing else changed).
// Byte o of env b's record of tick t lives at  t * B * R + B * P(o) + b * W(o) + (o - P(o)),  P(o) = start of o's piece, W(o) its
// width (susnet_record_layout_t::planar; record_piece() below is the same rule for the host and the consumers).
struct RecordPiece { uint32_t start, width; };
__host__ __device__ constexpr RecordPiece record_piece(uint32_t record_bytes, uint32_t o) {
    const uint32_t full = record_bytes / 16u * 16u;
    if (o < full) return RecordPiece{o / 16u * 16u, 16u};
    const uint32_t rem = record_bytes - full; // 4, 8 or 12 (= 8 + 4)
    if (rem >= 8u && o < full + 8u) return RecordPiece{full, 8u};
    return RecordPiece{rem >= 8u ? full + 8u : full, 4u};
}
// dwords [D0, D0 + N) of my environment's record (N = 1, 2 or 4; never across a piece): one store.  d.vo = 0, d.so = the tick's slab
template <int RECORD_DWORDS, int D0, int N>
__device__ __forceinline__ void store_record_piece(const BufDst &d, uint32_t B, uint32_t b, const uint32_t *v) {
    constexpr RecordPiece pc = record_piece(4u * RECORD_DWORDS, 4u * D0);
    static_assert(4u * D0 + 4u * N <= pc.start + pc.width, "a store stays inside one piece");
    const uint32_t at = B * pc.start + b * pc.width + (4u * D0 - pc.start);
    if (N == 4) d.st128(at, v[0], v[1], v[2], v[3]);
    else if (N == 2) d.st64(at, v[0], v[1]);
    else d.st32(at, v[0]);
}
// all RECORD_DWORDS dwords, piece by piece
template <int RECORD_DWORDS>
__device__ __forceinline__ void store_record_planar(const BufDst &d, uint32_t B, uint32_t b, const uint32_t *rec) {
    constexpr int k4 = RECORD_DWORDS / 4 * 4;
    static_for<0, RECORD_DWORDS / 4>([&](auto p) __attribute__((always_inline)) { store_record_piece<RECORD_DWORDS, 4 * decltype(p)::value, 4>(d, B, b, rec + 4 * decltype(p)::value); });
    if constexpr (RECORD_DWORDS - k4 >= 2) store_record_piece<RECORD_DWORDS, k4, 2>(d, B, b, rec + k4);
    if constexpr ((RECORD_DWORDS - k4) & 1) store_record_piece<RECORD_DWORDS, RECORD_DWORDS - 1, 1>(d, B, b, rec + RECORD_DWORDS - 1);
}

// Fused random rollout of the byte-parallel (SWAR) configurations: the same contract as k_rollout, the state held as
// packed bytes (susnet_swar.h) for the whole launch.  OUT_RECORD is not offered for these configurations.
template <class S, int OUT, class RNG = PhiloxRng>
__global__ __launch_bounds__(kBlock) void k_rollout_swar(Consts c, State s, RolloutArgs a, ObsArgs o) {
    using W = Swar<S>;
    constexpr int A = W::A, NW = W::NW;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const uint32_t nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; // XCD-aware mapping (see k_rollout)
    const uint32_t per = nblk >> 3, rem = nblk & 7u;
    const uint32_t wave_id = xcd * per + (xcd < rem ? xcd : rem) + slot;
    const int64_t b0 = (int64_t)wave_id * c.epw, b = b0 + tid;
    const bool active = tid < c.epw && b < c.B;
    const int nrows = (int)((c.B - b0) < c.epw ? (c.B - b0) : c.epw);
    typename StoreFor<S>::type st;
    Tables T = setup_lds<S, true, true>(c, smem, tid, st);
    if (tid < 16) T.comp[tid] = (uint32_t)o.comp[tid];
    if constexpr (!RNG::kNumpy && RankLut<S>::kOk) build_rank_lut<S>(smem, tid);
    JobMap::clear_all(kJobMapWord, c.N, tid); // the cell -> job map (susnet_swar.h): zeroed once, then kept by the lanes that own the columns
    wave_lds_fence();
    JobMap jm;
    jm.init(kJobMapWord, tid);
    Env e = {};
    RNG rng = make_rng<RNG>(c, s, active ? b : 0);
    if (active) {
        load_env<S>(c, s, st, b, e);
        jm.set_jobs(st, S::J(c));
    }
    constexpr bool kFlat = OUT == OUT_TRAJ_FLAT; // (cooperative feature stores: every lane stays)
    constexpr bool kTraj = OUT == OUT_TRAJ_RAW8 || OUT == OUT_TRAJ || kFlat;
    // the family's raw row has a run-time length (the job count): still written straight from registers, by the one case of a wave-uniform
    // switch over the job count that applies (round 4 sent it through the cooperative LDS writer: 37 G where the packed record reached 70)
    constexpr bool kFamRaw = OUT == OUT_TRAJ_RAW8 && S::kRawF < 0;
    if (((kTraj && !kFlat) || OUT == OUT_NONE || OUT == OUT_RECORD) && !active) return; // no cooperative work past this point in these modes
    W w;
    to_swar<S>(c, st, e, w);
    uint64_t tick_base = a.tick_base;
    if (c.dev_tick) tick_base = uniform64(s.tickw[active ? b : b0]); // (every env holds the same count)
    LifeAcc life;
    life.clear();
    const int64_t AB = (int64_t)A * c.B;
    const int64_t bb = active ? b : 0;
    // OUT_ANY: the four optional trajectory outputs, addressed from the tick index where they are stored (round 4 carried four 64-bit
    // per-lane cursors through the step: eight registers the 12-agent kernels do not have)
    auto any_ptr = [&](auto *base, int tick, int64_t per_tick, int64_t lane_off) __attribute__((always_inline)) {
        return base ? base + ((int64_t)tick * per_tick + lane_off) : base;
    };
    constexpr int kRawF = S::kRawF;
    const uint64_t nt = (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    BufDst da = make_buf_dst(a.actions, nt * (uint64_t)AB, (uint32_t)(bb * A));
    BufDst dr = make_buf_dst(a.rewards, nt * (uint64_t)AB * 4u, (uint32_t)(bb * A) * 4u);
    BufDst dd = make_buf_dst(a.done, nt * (uint64_t)c.B, (uint32_t)bb);
    BufDst dt = make_buf_dst(a.trunc, nt * (uint64_t)c.B, (uint32_t)bb);
    BufDst dobs = make_buf_dst(o.out, nt * (uint64_t)o.tick_stride, (uint32_t)bb * (uint32_t)(kRawF > 0 ? kRawF : o.F));
    const uint32_t slab_a = (uint32_t)AB, slab_d = (uint32_t)c.B, slab_o = (uint32_t)o.tick_stride;
    // OUT_TRAJ_FLAT: the wave's float32 feature rows [B][F], written cooperatively (susnet_flat.h); base = the wave's first row
    using FlatRowT = typename FlatFor<S>::Row;
    BufDst dflat = make_buf_dst(o.out, nt * (uint64_t)o.tick_stride * 4u, (uint32_t)b0 * (uint32_t)(FlatRowT::F * 4));
    FlatRowT frow;
    frow.clear();
    constexpr bool kRec = OUT == OUT_RECORD; // one packed record per env-step: rewards f32[A] | actions u8[A] | raw obs u8[F] | done | truncated | 0-padding
    constexpr bool kFam = S::kRawF < 0;      // run-time job count: the family's record layout (FamRecord)
    constexpr int kRecDwords = kFam ? FamRecord<S>::kDwords : RecordLayout<S>::kDwords;
    BufDst drec = make_buf_dst(a.record, nt * (uint64_t)c.B * (uint64_t)a.record_bytes, 0u); // (planes: store_record_planar adds the piece offsets)
    const uint32_t slab_rec = (uint32_t)c.B * (uint32_t)a.record_bytes;
    if (active && a.n_ticks > 0) {
        clear_info_if_fresh(e); // once per launch instead of once per tick
        rng.align();            // the first step aligns the event cursor; later steps find it aligned (step_swar: realign)
    }
    // production stream: the words of a group of 4 ticks (whole Philox blocks) are generated at the group's first tick and staged
    // in LDS (GroupWords); the tick loop is rolled -- one copy of the step and of the reset path
    static_assert(HasGroupWords<S>::value, "grouped action stream");
    // (a shuffled order comes from the rank tables where the tick's shuffle digits sit in the words the tables assume -- RankLut::kOk;
    // elsewhere from the insertion rule itself, ranks_from_stream)
    using GW = GroupWords<S::kAw.W, false>;
    GW gw;
    gw.init(kGroupWordsWord<S>(), tid);
    // SOFTWARE PIPELINE (production stream): a tick's actions and turn ranks depend on nothing but the tick index and the
    // episode's roles, so tick t + 1 is sampled while tick t steps, in three stages placed where other work covers their LDS
    // round trips (with the stages in sequence 28 % of the wave's cycles were parked on s_waitcnt):
    //   A  top of tick t          read the words of tick t + 1 (refill the group first if t + 1 starts one)
    //   B  middle of tick t       (after the kill section) action digits of t + 1 from those words, rank-table reads ISSUED
    //   C  top of tick t + 1      rank bytes from the table words that arrived long ago
    // A reset at tick t changes the roles: the lanes that reset redo B from the same words inside the (rare) reset branch.
#ifdef SUSNET_STAMPS
    unsigned long long wseg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wseg2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wprev = 0;
#define KSTAMP(k) do { unsigned long long tn = __builtin_readcyclecounter(); wseg[k] += tn - wprev; wprev = tn; } while (0)
#else
#define KSTAMP(k) do {} while (0)
#endif
    uint32_t act_n[NW], R_n[NW];
    RankRaw raw_n = {0u, 0u, 0u, 0u};
    TickWords<GW::W> tw_n;
    identity_ranks<S>(R_n);
#pragma unroll
    for (int q = 0; q < NW; q++) act_n[q] = 0u;
#pragma unroll
    for (int k = 0; k < GW::W; k++) tw_n.wd[k] = 0u;
    tw_n.rem = 0u;
    auto stage_a = [&](int tick, bool first) __attribute__((always_inline)) {
        if constexpr (!RNG::kNumpy) {
            const uint64_t gt = tick_base + (uint64_t)tick;
            const uint32_t pos = (uint32_t)gt & (uint32_t)(GW::G - 1);
            if (first || pos == 0u) { // (wave-uniform)
                gw.refill(rng, gt / (uint64_t)GW::G);
                wave_lds_publish();
            }
#pragma unroll
            for (int k = 0; k < GW::W; k++) tw_n.wd[k] = gw.read(pos, k);
        }
    };
    // (the step's `mid` interface has two parts -- the two-lane kernel puts each inside one of the step's gates, susnet_swar.h NoMid; here
    // part 0 is all of B and the step runs it between its kill and job sections: split and placed in the gates this kernel measured slower)
    auto stage_b = [&](int part) __attribute__((always_inline)) {
        if constexpr (!RNG::kNumpy) {
            if (part == 0) {
                TickWords<GW::W> tw = tw_n;
                sample_actions_swar<S, 0>(c, w, rng, tw, 0ull, act_n);
                if constexpr (RankLut<S>::kOk) raw_n = ranks_lut_issue<S, 0>(rng, tw, 0ull);
                else if constexpr (S::kOrd > 0) ranks_from_stream<S, 0>(c, 0u, rng, tw, 0ull, true, R_n);
            }
        }
    };
    MidParts<decltype(stage_b)> mid_b{stage_b, tw_n.wd[0], tw_n.wd[0]};
    if (!RNG::kNumpy && active && a.n_ticks > 0) {
        stage_a(0, true);
        stage_b(0);
    }
    // the next episode of my environment, drawn ahead (production stream; see draw_episode).  Not in OUT_ANY: the drawn-ahead episode is a
    // second copy of the whole byte-parallel state (47 registers with 8 agents and 2 imposters), and next to the general observation
    // writer that pushed the 5..8-agent kernels past 256 vector registers, i.e. into accumulator-register copies inside code that runs
    // under divergent EXEC by design (isa_checks.parked_under_divergence).  There a finishing lane draws its episode on the spot -- the
    // same RESET-stream words, so the same episode.
    // (nor with 9 .. 12 agents: three words per quantity, 60 registers per copy of the state)
    constexpr bool kAhead = !RNG::kNumpy && OUT != OUT_ANY && S::kA <= 8;
    W wn = w;
    typename StoreFor<S>::type stn = st;
    uint32_t impn = 0;
    bool have_next = false;
    auto tick_body = [&](int tick) __attribute__((always_inline)) {
        if (kTraj) { // this tick's slabs: scalar offsets derived from the (wave-uniform) tick index, nothing loop-carried
            const uint32_t t32 = (uint32_t)tick;
            da.so = t32 * slab_a; dr.so = t32 * (4u * slab_a); dd.so = t32 * slab_d; dt.so = t32 * slab_d; dobs.so = t32 * slab_o;
        }
        if (kRec) drec = record_slab(a.record, slab_rec, tick, 0u);
        if (active) {
            uint32_t act[NW], R[NW];
            if constexpr (RNG::kNumpy) { // numpy parity: base.py:326-330, then np.random.shuffle (base.py:372-374) from the env's own words
                sample_actions_swar<S>(c, w, rng, act);
                identity_ranks<S>(R);
                if (S::kOrd > 0) {
                    OrderOf<S> ord = (OrderOf<S>)0xFEDCBA9876543210ull;
                    if constexpr (A > 8) shuffle_nibbles_rolled(rng, ord, A);
                    else shuffle_nibbles<false>(rng, ord, A);
                    ranks_from_order<S>(ord, R);
                }
            } else {
#ifdef SUSNET_STAMPS
                wprev = __builtin_readcyclecounter();
#endif
#pragma unroll
                for (int q = 0; q < NW; q++) act[q] = act_n[q];
                if constexpr (RankLut<S>::kOk) ranks_lut_finish<S>(raw_n, R); // stage C
                else {
#pragma unroll
                    for (int q = 0; q < NW; q++) R[q] = R_n[q];
                }
                stage_a(tick + 1, false); // (also past the launch's last tick: nothing of it is kept)
                KSTAMP(0);
            }
            if (kTraj) store_packed_bytes<A>(da, act);
            else if (OUT == OUT_ANY && a.actions != nullptr) store_packed_bytes<A>(PtrDst{any_ptr(a.actions, tick, AB, bb * A)}, act);
            float rr[A];
            bool done, trunc;
            // (the win rules run unconditionally at the launch's first tick only: the state may come from outside; see step_swar)
#ifdef SUSNET_STAMPS
            step_swar<S>(c, T, w, e, rng, act, R, rr, done, trunc, wseg2, mid_b, jm, tick == 0, tick != a.n_ticks - 1);
#else
            step_swar<S>(c, T, w, e, rng, act, R, rr, done, trunc, nullptr, mid_b, jm, tick == 0, tick != a.n_ticks - 1);
#endif
            KSTAMP(1);
            if (kTraj) {
                store_row_f32<A>(dr, rr);
                dd.st8(0u, done ? 1u : 0u);
                dt.st8(0u, trunc ? 1u : 0u);
            } else if (OUT == OUT_ANY) {
                if (a.rewards != nullptr) store_row_f32<A>(PtrDst{reinterpret_cast<uint8_t *>(any_ptr(a.rewards, tick, AB, bb * A))}, rr);
                if (a.done != nullptr) *any_ptr(a.done, tick, (int64_t)c.B, bb) = done ? 1 : 0;
                if (a.trunc != nullptr) *any_ptr(a.trunc, tick, (int64_t)c.B, bb) = trunc ? 1 : 0;
            }
            if (kFeed(OUT) && a.roles != nullptr) a.roles[(int64_t)tick * c.B + b] = (uint16_t)swar_imp_bits(w);
            KSTAMP(2);
            const bool fin = done || trunc;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(fin) != 0ull, 0)) {
                if constexpr (kAhead) {
                    // a finishing lane without a drawn episode: draw now -- for every lane that has none (see draw_episode)
                    if (__builtin_amdgcn_ballot_w64(fin && !have_next) != 0ull) {
                        if (!have_next) {
                            Env en;
                            draw_episode<S>(c, T, tid, rng, e.ep, stn, en); // (the index of my next reset, whether or not I finish now)
                            to_swar<S>(c, stn, en, wn);
                            impn = en.imp;
                            have_next = true;
                        }
                    }
                }
                if (fin) {
                    life.add_episode(e, trunc);
                    if (kFeed(OUT) && a.term_obs != nullptr) { // the terminal state, before the in-launch reset replaces it
                        if constexpr (kFam) { // (run-time row length: byte by byte, from the store form -- rare path)
                            from_swar<S>(c, w, st, e);
                            fill_raw<S>(c, st, e, a.term_obs + ((int64_t)tick * c.B + b) * o.F);
                        } else {
                            uint32_t trow[(kRawF + 3) / 4];
                            raw_row_swar<S>(w, trow, (uint32_t)c.tag_interval);
                            store_packed_bytes<kRawF>(PtrDst{a.term_obs + ((int64_t)tick * c.B + b) * kRawF}, trow);
                        }
                    }
                    jm.clear_jobs(st, S::J(c)); // the finished episode's job cells leave the map, the new ones enter
                    if constexpr (RNG::kNumpy) {
                        reset_env<S>(c, T, st, tid, e, rng);
                        to_swar<S>(c, st, e, w);
                    } else if constexpr (!kAhead) { // base.py:251-324 on the RESET stream, drawn on the spot (see kAhead)
                        Env en;
                        draw_episode<S>(c, T, tid, rng, e.ep, st, en);
                        to_swar<S>(c, st, en, w);
                        e.imp = en.imp;
                        e.alive = (1u << A) - 1u;
                        e.jd = e.used = e.timer = e.t = 0u;
                        e.ep += 1u;
                    } else { // base.py:251-324, drawn ahead: take it
                        st = stn;
                        w = wn;
                        e.imp = impn;
                        e.alive = (1u << A) - 1u;
                        e.jd = e.used = e.timer = e.t = 0u;
                        e.ep += 1u;
                        have_next = false;
                    }
                    jm.set_jobs(st, S::J(c));
                    // new roles: the next tick's action digits again, from the words already fetched -- and its turn ranks, whose digits
                    // continue what the action draws (role-dependent ranges) left of their last word
                    stage_b(0);
                    // info counters of a terminal step stay readable until the next step: only the launch's last tick can be observed
                    if (tick == a.n_ticks - 1) e.flags |= FLAG_FRESH;
                    else zero_metrics(e);
                }
            }
            KSTAMP(3);
            if constexpr (OUT == OUT_TRAJ_RAW8 && !kFam) {
                uint32_t row[(kRawF + 3) / 4];
                raw_row_swar<S>(w, row, (uint32_t)c.tag_interval);
                store_packed_bytes<kRawF>(dobs, row);
            }
            if constexpr (kFamRaw) { // flatten_state with the handle's job count: one case runs (scalar compares)
                const int Jn = S::J(c);
                static_for<0, 9>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int JJ = decltype(jc)::value;
                    if (Jn == JJ) {
                        uint32_t row[RawRowOf<S, JJ>::kDwords];
                        raw_row_swar_n<S, JJ>(w, row, (uint32_t)c.tag_interval);
                        store_packed_bytes<RawRowOf<S, JJ>::F>(dobs, row);
                    }
                });
            }
            if constexpr (kFlat) { // the feature row of the state after the step (and after an in-launch reset)
                uint32_t fx[A], fy[A], fal[A];
#pragma unroll
                for (int i = 0; i < A; i++) {
                    const uint32_t cell = (w.xy[i / 4] >> (8 * (i & 3))) & 0xffu;
                    fx[i] = cell & 15u;
                    fy[i] = cell >> 4;
                    fal[i] = (w.al[i / 4] >> (8 * (i & 3))) & 1u;
                }
                frow.build(fx, fy, fal);
            }
            if constexpr (kRec && kFam) { // the family's record (FamRecord): nothing in it depends on the job count
                using FR = FamRecord<S>;
                uint8_t by[FR::kHeadDwords * 4];
                int k = 0;
#pragma unroll
                for (int i = 0; i < A; i++) by[k++] = (uint8_t)(act[i / 4] >> (8 * (i & 3)));
#pragma unroll
                for (int i = 0; i < A; i++) { // cells as flatten_state has them: x0 y0 x1 y1 ...
                    const uint32_t cell = w.xy[i / 4] >> (8 * (i & 3));
                    by[k++] = (uint8_t)(cell & 15u);
                    by[k++] = (uint8_t)((cell >> 4) & 15u);
                }
#pragma unroll
                for (int i = 0; i < A; i++) by[k++] = (uint8_t)((w.al[i / 4] >> (8 * (i & 3))) & 1u);
                if (W::kTag) { // tagging.py:220-230: used_tag_actions, tag_counts, steps until the vote
#pragma unroll
                    for (int i = 0; i < A; i++) by[k++] = (uint8_t)((w.used[i / 4] >> (8 * (i & 3))) & 1u);
#pragma unroll
                    for (int i = 0; i < A; i++) by[k++] = (uint8_t)(w.cnt[i / 4] >> (8 * (i & 3)));
                    by[k++] = (uint8_t)((uint32_t)c.tag_interval - w.timer);
                }
                by[k++] = done ? 1 : 0;
                by[k++] = trunc ? 1 : 0;
#pragma unroll
                for (; k < FR::kHeadDwords * 4; k++) by[k] = 0;
                uint32_t rec[kRecDwords];
#pragma unroll
                for (int i = 0; i < A; i++) rec[i] = __float_as_uint(rr[i]);
#pragma unroll
                for (int q = 0; q < FR::kHeadDwords; q++)
                    rec[A + q] = (uint32_t)by[4 * q] | ((uint32_t)by[4 * q + 1] << 8) | ((uint32_t)by[4 * q + 2] << 16) | ((uint32_t)by[4 * q + 3] << 24);
#pragma unroll
                for (int q = 0; q < 4; q++) rec[A + FR::kHeadDwords + q] = w.jobs_obs[q]; // the eight job-cell slots (x, y bytes; zeros past the job count)
                rec[A + FR::kHeadDwords + 4] = w.jd[0];                                    // the eight job-status slots
                rec[A + FR::kHeadDwords + 5] = w.jd[W::JW - 1];
                store_record_planar<kRecDwords>(drec, (uint32_t)c.B, (uint32_t)bb, rec);
            }
            if constexpr (kRec && !kFam) { // (byte moves between statically known positions: the compiler folds them into v_perm / v_alignbyte)
                uint32_t row[(kRawF + 3) / 4];
                raw_row_swar<S>(w, row, (uint32_t)c.tag_interval);
                constexpr int kNB = A + 2 + kRawF;
                uint8_t by[(kNB + 3) / 4 * 4];
#pragma unroll
                for (int i = 0; i < A; i++) by[i] = (uint8_t)(act[i / 4] >> (8 * (i & 3)));
                // the raw row right behind the actions: its job cells then start at byte 8A of the record, i.e. the words that hold
                // them (and, for an even job count, the status word) go out as they are; done | truncated follow the row
#pragma unroll
                for (int f = 0; f < kRawF; f++) by[A + f] = (uint8_t)(row[f / 4] >> (8 * (f & 3)));
                by[A + kRawF] = done ? 1 : 0;
                by[A + kRawF + 1] = trunc ? 1 : 0;
#pragma unroll
                for (int q = kNB; q < (kNB + 3) / 4 * 4; q++) by[q] = 0;
                uint32_t rec[kRecDwords];
#pragma unroll
                for (int i = 0; i < A; i++) rec[i] = __float_as_uint(rr[i]);
#pragma unroll
                for (int q = 0; q < (kNB + 3) / 4; q++)
                    rec[A + q] = (uint32_t)by[4 * q] | ((uint32_t)by[4 * q + 1] << 8) | ((uint32_t)by[4 * q + 2] << 16) | ((uint32_t)by[4 * q + 3] << 24);
                store_record_planar<kRecDwords>(drec, (uint32_t)c.B, (uint32_t)bb, rec);
            }
            KSTAMP(4);
        }
        if (OUT == OUT_ANY) { // any observation mode: through the cooperative writer, on the bitmask / packed-store form
            if (active) from_swar<S>(c, w, st, e);
            // (an opaque copy of the lane id: the writer's lane-dependent addresses -- row pointers, the copy-out's piece offsets -- are
            // invariant over the tick loop, and hoisted out of it they are held across the whole step: with them the 5..8-agent kernels
            // needed up to 304 vector registers)
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            write_obs<S>(c, o, T, st, tid_o, e, active, b0, nrows, tick);
        }

        if constexpr (kFlat) flat_store_wave(frow, T.stage, tid, nrows, dflat.r, dflat.vo + (uint32_t)tick * (slab_o * 4u));
    };
#pragma clang loop unroll(disable)
    for (int tick = 0; tick < a.n_ticks; tick++) tick_body(tick);
#ifdef SUSNET_STAMPS
    if (blockIdx.x == 100 && tid == 0)
        for (int k = 0; k < 8; k++) {
            atomicAdd(reinterpret_cast<unsigned long long *>(s.err) + 2 + k, wseg[k]);
            atomicAdd(reinterpret_cast<unsigned long long *>(s.err) + 10 + k, wseg2[k]);
        }
#endif
    if (active) {
        const State se = kernarg_reload<State>(kStateArgOffset); // (not `s`: see kernarg_reload)
        from_swar<S>(c, w, st, e);
        store_env<S>(c, se, st, b, e, true);
        finish_rng(se, b, rng);
        life.flush(c, se, b, (uint32_t)(a.n_ticks > 0 ? a.n_ticks : 0));
        if (c.dev_tick) se.tickw[b] = tick_base + (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    }
}

// Fused random rollout with TWO lanes per environment (susnet_swar2.h): 8-agent configurations at 32 envs per wave.  Same contract
// as k_rollout_swar for the modes OUT_NONE / OUT_TRAJ / OUT_TRAJ_RAW8 / OUT_RECORD; RNG = the production stream, or
// caller-supplied words (numpy parity: both lanes of a pair walk the environment's tape identically and keep their own half).
template <class S, int OUT, class RNG = PhiloxRng>
__global__ __launch_bounds__(kBlock) void k_rollout_swar2(Consts c, State s, RolloutArgs a, ObsArgs o) {
    using W = Swar2<S>;
    constexpr int A = W::A;
    static_assert(A == 8, "two words of four agents");
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const uint32_t nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; // XCD-aware mapping (see k_rollout)
    const uint32_t per = nblk >> 3, rem = nblk & 7u;
    const uint32_t wave_id = xcd * per + (xcd < rem ? xcd : rem) + slot;
    const uint32_t h = (uint32_t)tid >> 5;
    const int64_t b0 = (int64_t)wave_id * 32, b = b0 + (tid & 31);
    const bool active = b < c.B;
    typename StoreFor<S>::type st;
    Tables T = setup_lds<S, true, true>(c, smem, tid, st);
    if (RankLut<S>::kOk) build_rank_lut<S>(smem, tid);
    JobMap::clear_all(kJobMapWord, c.N, tid); // the cell -> job map (susnet_swar.h): one column per environment, shared by its two lanes
    wave_lds_fence();
    JobMap jm;
    jm.init(kJobMapWord, tid & 31);
    Env e = {};
    RNG rng = make_rng<RNG>(c, s, active ? b : 0);
    if (!active) return; // both lanes of a pair leave together: every exchange below is between two active lanes
    load_env<S>(c, s, st, b, e);
    jm.set_jobs(st, S::J(c)); // (both lanes of the pair write the same bytes)
    W w;
    to_swar2<S>(c, st, e, h, w);
    uint64_t tick_base = a.tick_base;
    if (c.dev_tick) tick_base = uniform64(s.tickw[b]);
    LifeAcc life;
    life.clear();
    const int64_t AB = (int64_t)A * c.B;
    constexpr int kRawF = S::kRawF;
    constexpr bool kTraj = OUT == OUT_TRAJ_RAW8 || OUT == OUT_TRAJ, kRec = OUT == OUT_RECORD;
    const uint64_t nt = (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    // every lane writes its own four agents' actions / rewards / cells / alive flags; what the environment has once (done,
    // truncated, job cells and status, the replay feed) is written by the lane of the low word
    BufDst da = make_buf_dst(a.actions, nt * (uint64_t)AB, (uint32_t)(b * A) + 4u * h);
    BufDst dr = make_buf_dst(a.rewards, nt * (uint64_t)AB * 4u, ((uint32_t)(b * A) + 4u * h) * 4u);
    BufDst dd = make_buf_dst(a.done, nt * (uint64_t)c.B, (uint32_t)b);
    BufDst dt = make_buf_dst(a.trunc, nt * (uint64_t)c.B, (uint32_t)b);
    BufDst dobs = make_buf_dst(o.out, nt * (uint64_t)o.tick_stride, (uint32_t)(b * kRawF));
    // the packed record as planes of 16-byte pieces (see store_record_planar): 20 dwords = rewards [0, 8) | actions [8, 10) | cells
    // [10, 14) | alive [14, 16) | job cells [16, 18) | job status 18 | done, truncated 19; lane (b, h) stores its own four of each
    BufDst drec = make_buf_dst(a.record, nt * (uint64_t)c.B * (uint64_t)a.record_bytes, 0u);
    const uint32_t rec_pb = (uint32_t)c.B * 16u, rec_b16 = (uint32_t)b * 16u;
    const uint32_t at_rew = h * rec_pb + rec_b16, at_act = 2u * rec_pb + rec_b16 + 4u * h, at_pos = (2u + h) * rec_pb + rec_b16 + (h ? 0u : 8u),
                   at_al = 3u * rec_pb + rec_b16 + 8u + 4u * h, at_tail = h ? 0xffffff00u : 4u * rec_pb + rec_b16; // (0xffffff00: past any slab a 32-bit record count can describe)
    const uint32_t slab_a = (uint32_t)AB, slab_d = (uint32_t)c.B, slab_o = (uint32_t)o.tick_stride;
    const uint32_t slab_rec = (uint32_t)c.B * (uint32_t)a.record_bytes;
    if (a.n_ticks > 0) {
        clear_info_if_fresh(e);
        rng.align(); // the first step aligns the event cursor; later steps find it aligned (step_swar2: realign)
    }
    // the action stream: groups of 8 ticks = whole PAIRS of Philox blocks, lane h generating the blocks 2k + h, staged in LDS
    // (GroupWords); the tick loop is rolled -- one copy of the step and of the reset path
    static_assert(HasGroupWords<S>::value && (RankLut<S>::kOk || S::kOrd == 0), "grouped action stream; a shuffled order comes from the rank tables");
    using GW = GroupWords<S::kAw.W, true>;
    GW gw;
    gw.init(kGroupWordsWord<S>(), tid);
    // SOFTWARE PIPELINE in three stages, each placed where its LDS round trip is covered by other work (with the stages in
    // sequence 29 % of the wave's cycles were parked on s_waitcnt):
    //   A  top of tick t          read the words of tick t + 1 (refill the group first if t + 1 starts one)
    //   B  middle of tick t       (after the kill section) action digits of t + 1 from those words, rank-table reads ISSUED
    //   C  top of tick t + 1      rank bytes from the table words that arrived long ago
    // A reset at tick t re-derives B for the new roles from the words already fetched.
    uint32_t act_n = 0u;
    RankRaw raw_n = {0u, 0u, 0u, 0u};
    TickWords<GW::W> tw_n;
#pragma unroll
    for (int k = 0; k < GW::W; k++) tw_n.wd[k] = 0u;
    tw_n.rem = 0u;
    auto stage_a = [&](int tick, bool first) __attribute__((always_inline)) {
        if constexpr (!RNG::kNumpy) {
            const uint64_t gt = tick_base + (uint64_t)tick;
            const uint32_t pos = (uint32_t)gt & (uint32_t)(GW::G - 1);
            if (first || pos == 0u) { // (wave-uniform)
                gw.refill(rng, gt / (uint64_t)GW::G);
                wave_lds_publish();
            }
#pragma unroll
            for (int k = 0; k < GW::W; k++) tw_n.wd[k] = gw.read(pos, k);
        }
    };
    TickWords<GW::W> tw_b = tw_n; // (B in two parts: see k_rollout_swar)
    auto stage_b = [&](int part) __attribute__((always_inline)) {
        if constexpr (!RNG::kNumpy) {
            if (part == 0) {
                tw_b = tw_n;
                act_n = sample_actions_pair<S, 0>(w, rng, tw_b, 0ull);
            } else {
                if constexpr (RankLut<S>::kOk) raw_n = ranks_lut_issue<S, 0>(rng, tw_b, 0ull);
            }
        }
    };
    MidParts<decltype(stage_b)> mid_b{stage_b, tw_n.wd[0], tw_b.rem};
    if (!RNG::kNumpy && a.n_ticks > 0) {
        stage_a(0, true);
        stage_b(0);
        stage_b(1);
    }
    // the next episode of my environment, drawn ahead (production stream; see draw_episode): both lanes of a pair hold their own half
    W wn = w;
    typename StoreFor<S>::type stn = st;
    uint32_t impn = 0;
    bool have_next = false;
    auto tick_body = [&](int tick) __attribute__((always_inline)) {
        if (kTraj) {
            const uint32_t t32 = (uint32_t)tick;
            da.so = t32 * slab_a; dr.so = t32 * (4u * slab_a); dd.so = t32 * slab_d; dt.so = t32 * slab_d; dobs.so = t32 * slab_o;
        }
        if (kRec) drec = record_slab(a.record, slab_rec, tick, 0u);
        uint32_t act, R;
        if constexpr (RNG::kNumpy) { // numpy parity: base.py:326-330, then np.random.shuffle (base.py:372-374), from the env's own words
            uint32_t a2[2] = {0u, 0u}, R2[2];
#pragma unroll
            for (int i = 0; i < A; i++) a2[i / 4] |= rng.bounded(S::nr_crew(c) + ((w.imp_bits >> i) & 1u)) << (8 * (i & 3));
            identity_ranks<S>(R2);
            if (S::kOrd > 0) {
                OrderOf<S> ord = (OrderOf<S>)0xFEDCBA9876543210ull;
                shuffle_nibbles<false>(rng, ord, A);
                ranks_from_order<S>(ord, R2);
            }
            act = h ? a2[1] : a2[0];
            R = h ? R2[1] : R2[0];
        } else {
            uint32_t R2[2];
            if constexpr (RankLut<S>::kOk) ranks_lut_finish<S>(raw_n, R2); // stage C
            else identity_ranks<S>(R2);
            act = act_n;
            R = h ? R2[1] : R2[0];
            stage_a(tick + 1, false); // (also past the launch's last tick: nothing of it is kept)
        }
        float rr[4];
        bool done, trunc;
        if constexpr (RNG::kNumpy) step_swar2<S>(c, w, e, rng, act, R, rr, done, trunc, jm, tick == 0, false);
        else step_swar2<S>(c, w, e, rng, act, R, rr, done, trunc, jm, tick == 0, tick != a.n_ticks - 1, mid_b);
        // (the rewards come out of LDS lookups issued at the very end of the step: they are stored LAST, behind everything else
        // the tick writes, so that nothing waits for them)
        if (kTraj) {
            da.st32(0u, act);
            if (h == 0u) {
                dd.st8(0u, done ? 1u : 0u);
                dt.st8(0u, trunc ? 1u : 0u);
            }
        }
        if (kRec) drec.st32(at_act, act); // my four action bytes
        if (kFeed(OUT) && a.roles != nullptr && h == 0u) a.roles[(int64_t)tick * c.B + b] = (uint16_t)w.imp_bits;
        const bool fin = done || trunc; // (both lanes of a pair: done / truncated are the environment's)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(fin) != 0ull, 0)) {
            if constexpr (!RNG::kNumpy) {
                // a finishing environment without a drawn episode: draw now -- for every environment of the wave that has none
                if (__builtin_amdgcn_ballot_w64(fin && !have_next) != 0ull) {
                    if (!have_next) {
                        Env en;
                        draw_episode<S>(c, T, tid, rng, e.ep, stn, en); // (the index of my next reset, whether or not I finish now)
                        to_swar2<S>(c, stn, en, h, wn);
                        impn = en.imp;
                        have_next = true;
                    }
                }
            }
            if (fin) {
                life.add_episode(e, trunc);
                if (kFeed(OUT) && a.term_obs != nullptr) { // the terminal state, before the in-launch reset replaces it
                    Swar<S> f;
                    gather_swar2<S>(w, f);
                    uint32_t trow[(kRawF + 3) / 4];
                    raw_row_swar<S>(f, trow);
                    if (h == 0u) store_packed_bytes<kRawF>(PtrDst{a.term_obs + ((int64_t)tick * c.B + b) * kRawF}, trow);
                }
                jm.clear_jobs(st, S::J(c)); // the finished episode's job cells leave the map, the new ones enter
                if constexpr (RNG::kNumpy) {
                    reset_env<S>(c, T, st, tid, e, rng);
                    to_swar2<S>(c, st, e, h, w);
                } else { // base.py:251-324, drawn ahead: take it
                    st = stn;
                    w = wn;
                    e.imp = impn;
                    e.alive = (1u << A) - 1u;
                    e.jd = e.used = e.timer = e.t = 0u;
                    e.ep += 1u;
                    have_next = false;
                }
                jm.set_jobs(st, S::J(c));
                // new roles: the next tick's action digits again, from the words already fetched -- and its turn ranks, whose digits
                // continue what the action draws (role-dependent ranges) left of their last word
                if constexpr (!RNG::kNumpy) {
                    stage_b(0);
                    stage_b(1);
                }
                if (tick == a.n_ticks - 1) e.flags |= FLAG_FRESH;
                else zero_metrics(e);
            }
        }
        if (OUT == OUT_TRAJ_RAW8) { // flatten_state (base.py:234-235): cells [0, 16) | alive [16, 24) | job cells [24, 32) | job status [32, 36)
            const uint32_t x = w.xy & 0x0f0f0f0fu, y = (w.xy >> 4) & 0x0f0f0f0fu;
            dobs.st64(8u * h, __builtin_amdgcn_perm(y, x, 0x05010400u), __builtin_amdgcn_perm(y, x, 0x07030602u));
            dobs.st32(16u + 4u * h, w.al & k01);
            if (h == 0u) {
                dobs.st64(24u, w.jobs_obs[0], w.jobs_obs[1]);
                dobs.st32(32u, w.jd);
            }
        }
        if (kRec) { // the raw row at [5A, 5A + F) -- every lane its own dwords of it -- then done | truncated | 0-padding (low lane)
            static_assert(kRawF == 36 && A == 8, "cells [0, 16) | alive [16, 24) | job cells [24, 32) | job status [32, 36)");
            const uint32_t x = w.xy & 0x0f0f0f0fu, y = (w.xy >> 4) & 0x0f0f0f0fu;
            drec.st64(at_pos, __builtin_amdgcn_perm(y, x, 0x05010400u), __builtin_amdgcn_perm(y, x, 0x07030602u));
            drec.st32(at_al, w.al & k01);
            // (the tail is the environment's: the lane of the high word stores it to an offset the tick slab's range check drops -- a divergent
            // `if (h == 0)` is a branch per tick, ~20 cycles at one wave per SIMD)
            drec.st128(at_tail, w.jobs_obs[0], w.jobs_obs[1], w.jd, (done ? 1u : 0u) | (trunc ? 0x100u : 0u));
            drec.st128(at_rew, __float_as_uint(rr[0]), __float_as_uint(rr[1]), __float_as_uint(rr[2]), __float_as_uint(rr[3]));
        }
        if (kTraj) store_row_f32<4>(dr, rr);
    };
#pragma clang loop unroll(disable)
    for (int tick = 0; tick < a.n_ticks; tick++) tick_body(tick);
    {
        Swar<S> f;
        gather_swar2<S>(w, f);
        from_swar<S>(c, f, st, e);
    }
    if (h == 0u) {
        const State se = kernarg_reload<State>(kStateArgOffset); // (not `s`: see kernarg_reload)
        store_env<S>(c, se, st, b, e, true);
        finish_rng(se, b, rng);
        life.flush(c, se, b, (uint32_t)(a.n_ticks > 0 ? a.n_ticks : 0));
        if (c.dev_tick) se.tickw[b] = tick_base + (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    }
}

// Fused random rollout of the 1v1 no-walls game (susnet_duel.h): the headline kernel.  Same contract as k_rollout for the
// modes OUT_NONE / OUT_TRAJ / OUT_TRAJ_RAW8; RNG = the production stream or caller-supplied words (numpy parity).
// WALLS: the wall-map flavour of the step (susnet_duel.h DuelWallTable; Consts::duel_walls)
template <class RNG, int OUT, bool WALLS = false>
__global__ __launch_bounds__(kBlock) void k_rollout_duel(Consts c, State s, RolloutArgs a, ObsArgs o) {
    using S = Spec<2, 0, SUSNET_VARIANT_ITG, 0, 0, 1>;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const uint32_t nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; // XCD-aware mapping (see k_rollout)
    const uint32_t per = nblk >> 3, rem = nblk & 7u;
    const uint32_t wave_id = xcd * per + (xcd < rem ? xcd : rem) + slot;
    const int64_t b0 = (int64_t)wave_id * c.epw, b = b0 + tid;
    const bool active = tid < c.epw && b < c.B;
    typename StoreFor<S>::type st;
    Tables T = setup_lds<S>(c, smem, tid, st);
    if constexpr (WALLS) { // the cells' blocked-move bits sit right behind the table image; what the reset / the feature rows stage moves up
        DuelWallTable::build(c.N, tid);
        wave_lds_publish();
        T.stage += kDuelWallWords;
        T.perm += 4u * kDuelWallWords;
    }
    Env e = {};
    // OUT_TRAJ_FLAT stores the feature rows cooperatively, so every lane stays: a lane without an environment (ragged last
    // wave, fewer than 64 environments per wave) mirrors environment 0 and its stores are dropped by the buffer range check
    constexpr bool kFlat = OUT == OUT_TRAJ_FLAT;
    if (!kFlat && !active) return; // no cooperative work past this point
    const int nrows = (int)((c.B - b0) < c.epw ? (c.B - b0) : c.epw);
    const int64_t bl = active ? b : 0;
    const uint32_t ghost = active ? 0u : 0x7fffff00u; // added to per-lane buffer offsets: past the end of every array
    load_env<S>(c, s, st, bl, e);
    RNG rng = make_rng<RNG>(c, s, bl);
    ActionStream as;
    as.init();
    Duel d;
    to_duel(st, e, d);
    if constexpr (WALLS) DuelWallTable::lookup(d);
    DuelConsts k = make_duel_consts(c);
    // the low halves of the two reward tables are v_perm's second source every tick and an instruction takes one scalar operand: as
    // scalars they are copied to a vector register tick after tick (the compiler re-materialises rather than keep them); pinned here
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(k.lut0_lo), "=v"(k.lut1_lo) : "s"(k.lut0_lo), "s"(k.lut1_lo));
    uint64_t tick_base = a.tick_base;
    if (c.dev_tick) tick_base = uniform64(s.tickw[bl]); // (every env holds the same count)
    LifeAcc life;
    life.clear();
    constexpr bool kTraj = OUT == OUT_TRAJ_RAW8 || OUT == OUT_TRAJ || kFlat;
    constexpr bool kRec = OUT == OUT_RECORD; // one 20-byte record per env-step: rewards | actions done truncated | x0 y0 x1 y1 | alive0 alive1 0 0
    constexpr bool kRec16 = OUT == OUT_RECORD16; // the compact 16-byte record (see OUT_RECORD16)
    constexpr uint32_t kRecBytes = kRec16 ? 16u : 20u;
    const uint64_t nt = (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
    const uint64_t B = (uint64_t)c.B;
    BufDst drec = make_buf_dst(a.record, nt * kRecBytes * B, (uint32_t)bl * kRecBytes);
    BufDst da = make_buf_dst(a.actions, nt * 2u * B, (uint32_t)bl * 2u + ghost);
    BufDst dr = make_buf_dst(a.rewards, nt * 8u * B, (uint32_t)bl * 8u + ghost);
    BufDst dd = make_buf_dst(a.done, nt * B, (uint32_t)bl + ghost);
    BufDst dt = make_buf_dst(a.trunc, nt * B, (uint32_t)bl + ghost);
    BufDst dobs = make_buf_dst(o.out, nt * (uint64_t)o.tick_stride, (uint32_t)bl * 6u);
    const uint32_t slab_d = (uint32_t)c.B, slab_o = (uint32_t)o.tick_stride;
    // OUT_TRAJ_FLAT: the wave's float32 feature rows [B][36] (susnet_flat.h); base = the wave's first row
    using FlatRowT = typename FlatFor<S>::Row;
    BufDst dflat = make_buf_dst(o.out, nt * (uint64_t)o.tick_stride * 4u, (uint32_t)b0 * (uint32_t)(FlatRowT::F * 4));
    if (a.n_ticks > 0) {
        clear_info_if_fresh(e); // once per launch instead of once per tick
        rng.align();            // the first step aligns the event-stream cursor; later steps find it aligned (see below)
    }
    // the next episode's spawn cells, drawn ahead (production stream; see draw_episode): one register
    uint32_t pqn = d.pq;
    bool have_next = false;
    auto tick_body = [&](int tick, auto par) __attribute__((always_inline)) {
        constexpr int POS = decltype(par)::value;
        if (kTraj) { // this tick's slabs: scalar offsets derived from the (wave-uniform) tick index, nothing loop-carried
            const uint32_t t32 = (uint32_t)tick;
            da.so = t32 * (2u * slab_d); dr.so = t32 * (8u * slab_d); dd.so = t32 * slab_d; dt.so = t32 * slab_d; dobs.so = t32 * slab_o;
        }
        if (kRec || kRec16) drec = record_slab(a.record, kRecBytes * slab_d, tick, (uint32_t)bl * kRecBytes);
        uint32_t a0, a1;
        if constexpr (RNG::kNumpy) { // base.py:326-330 with numpy's own words
            a0 = rng.bounded(6u);
            a1 = rng.bounded(5u);
        } else { // production stream: three ticks per word, 30 = 6 * 5 out of the word per tick (see sample_actions_env)
            const uint64_t gt = tick_base + (uint64_t)tick;
            uint32_t w;
            if (POS >= 0) {
                constexpr int q = POS >= 0 ? POS / kDuelTicksPerWord : 0, sl = POS >= 0 ? POS % kDuelTicksPerWord : 0;
                if (POS == 0) as.gen(rng, gt / (uint64_t)(4 * kDuelTicksPerWord));
                w = sl == 0 ? as.at(q) : as.rem;
            } else {
                const uint32_t sl = (uint32_t)(gt % (uint64_t)kDuelTicksPerWord);
                w = as.word(rng, gt / (uint64_t)kDuelTicksPerWord);
                w *= sl >= 1u ? kDuelRange : 1u;
                w *= sl >= 2u ? kDuelRange : 1u;
            }
            // hi32(w * 30) = 5 * a0 + a1 with a0 = hi32(w * 6), a1 = hi32(lo32(w * 6) * 5): the nested digits from ONE multiply
            const uint64_t p = (uint64_t)w * (uint64_t)kDuelRange;
            const uint32_t pair = (uint32_t)(p >> 32);
            as.rem = (uint32_t)p;
            a0 = (pair * 13u) >> 6; // pair / 5 for pair < 30
            a1 = pair - 5u * a0;
        }
        if (kTraj) da.st16(0u, a0 | (a1 << 8));
        float r0, r1;
        uint32_t done, trunc, hit;
        duel_step<RNG::kNumpy, WALLS>(k, d, e, rng.cur, a0, a1, r0, r1, done, trunc, &hit);
        if (kTraj) {
            dr.st64(0u, __float_as_uint(r0), __float_as_uint(r1));
            dd.st8(0u, done);
            dt.st8(0u, trunc);
        }
        if (kFeed(OUT) && a.roles != nullptr) a.roles[(int64_t)tick * c.B + b] = (uint16_t)1u; // the imposter is agent 0 (pred_prey.py:52-66, shuffle off)
        const bool fin = (done | trunc) != 0u;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(fin) != 0ull, 0)) {
            if constexpr (!RNG::kNumpy) {
                // a finishing lane without drawn spawn cells: draw now -- for every lane of the wave that has none (see draw_episode)
                if (__builtin_amdgcn_ballot_w64(fin && !have_next) != 0ull) {
                    if (!have_next) {
                        typename StoreFor<S>::type stn;
                        Env en;
                        draw_episode<S>(c, T, tid, rng, e.ep, stn, en); // (the index of my next reset, whether or not I finish now)
                        Duel dn;
                        to_duel(stn, en, dn);
                        pqn = dn.pq;
                        have_next = true;
                    }
                }
            }
            if (fin) {
                if (!RNG::kNumpy) rng.cur += (uint64_t)hit; // the landed kill's word of the event stream: a hit ends the episode (the crew has
                                                            // one member), so it is counted here instead of by a 64-bit add on every tick
                life.add_episode(e, trunc != 0u);
                if ((kFeed(OUT) || kRec || kRec16) && a.term_obs != nullptr) { // the terminal state, before the in-launch reset replaces it
                    PtrDst tp{a.term_obs + ((int64_t)tick * c.B + b) * 6};
                    tp.st32(0u, d.pq - k01);
                    tp.st16(4u, duel_alive_bytes(d));
                }
                if constexpr (RNG::kNumpy) {
                    reset_env<S>(c, T, st, tid, e, rng);
                    to_duel(st, e, d);
                } else { // base.py:251-324, drawn ahead: take it
                    d.pq = pqn;
                    d.lv = 0xffffffffu;
                    e.t = 0u;
                    e.ep += 1u;
                    have_next = false;
                }
                if constexpr (WALLS) DuelWallTable::lookup(d); // the spawn cells' bits (rare path: the wait sits in here)
                // info counters of a terminal step stay readable until the next step: only the launch's last tick can be observed.
                // The next step will align the event cursor: done right here unless this was the launch's last tick (then the
                // stored cursor is the one the last step left)
                if (tick == a.n_ticks - 1) e.flags |= FLAG_FRESH;
                else { zero_metrics(e); rng.align(); }
            }
        }
        if (OUT == OUT_TRAJ_RAW8) { // flatten_state: x0 y0 x1 y1 alive0 alive1
            dobs.st32(0u, d.pq - k01);
            dobs.st16(4u, duel_alive_bytes(d));
        }
        if (kRec) {
            drec.st128(0u, __float_as_uint(r0), __float_as_uint(r1), a0 | (a1 << 8) | (done << 16) | (trunc << 24), d.pq - k01);
            drec.st32(16u, duel_alive_bytes(d));
        }
        if (kRec16) // rewards | x0 y0 x1 y1 | alive0 alive1, a0 | a1 << 3 | done << 6 | truncated << 7, 0: ONE store, a whole 16-byte piece of a line
            drec.st128(0u, __float_as_uint(r0), __float_as_uint(r1), d.pq - k01, duel_alive_bytes(d) | ((a0 | (a1 << 3) | (done << 6) | (trunc << 7)) << 16));
        if constexpr (kFlat) { // onehot_pos of the state after the step (and after an in-launch reset)
            const uint32_t pos = d.pq - k01;
            const uint32_t fx[2] = {pos & 0xffu, (pos >> 16) & 0xffu}, fy[2] = {(pos >> 8) & 0xffu, pos >> 24}, fal[2] = {d.lv & 1u, d.lv >> 31};
            FlatRowT frow;
            frow.build(fx, fy, fal);
            flat_store_wave(frow, T.stage, tid, nrows, dflat.r, dflat.vo + (uint32_t)tick * (slab_o * 4u));
        }
    };
    constexpr int kGroup = RNG::kNumpy ? 0 : 4 * kDuelTicksPerWord;
    int tick = 0;
    while (tick < a.n_ticks) {
        if (kGroup > 0 && tick + kGroup <= a.n_ticks && ((tick_base + (uint64_t)tick) % (uint64_t)(kGroup > 0 ? kGroup : 1)) == 0ull) {
            static_for<0, (kGroup > 0 ? kGroup : 1)>([&](auto pos) __attribute__((always_inline)) { tick_body(tick + decltype(pos)::value, pos); });
            tick += kGroup;
        } else {
            tick_body(tick, std::integral_constant<int, -1>{});
            tick++;
        }
    }
    if (kFlat && !active) return;
    const State se = kernarg_reload<State>(kStateArgOffset); // (not `s`: see kernarg_reload)
    from_duel(d, st, e);
    store_env<S>(c, se, st, b, e, true);
    finish_rng(se, b, rng);
    life.flush(c, se, b, (uint32_t)(a.n_ticks > 0 ? a.n_ticks : 0));
    if (c.dev_tick) se.tickw[b] = tick_base + (uint64_t)(a.n_ticks > 0 ? a.n_ticks : 0);
}

// configurations compiled in (BASELINE.json configs 2, 3/5, 4); anything else runs the generic kernels
using SpecCfg2 = Spec<2, 0, SUSNET_VARIANT_ITG, 0, 0, 1>; // ImposterTrainingGround 1v1, no jobs, fixed order, imposter = agent 0 (any wall map)
using SpecCfg3 = Spec<3, 4, SUSNET_VARIANT_BASE, 1, -1, 1>; // FourRoomEnv 1v2, 4 jobs, random order (index-order SWAR step)
using SpecCfg4 = Spec<8, 4, SUSNET_VARIANT_BASE, 1, -1, 2>; // FourRoomEnv 2v6, 4 jobs, random order (index-order SWAR step)
using SpecTag5 = Spec<5, 5, SUSNET_VARIANT_TAGGING, 1, -1, 1>; // FourRoomEnvWithTagging 1v4, 5 jobs (notebooks/experiment.ipynb)
// agent count compiled in, everything else read at run time (any variant / order / up to 8 jobs): the packed-VGPR
// tables without a full specialisation
template <int A_> using SpecA = Spec<A_, -1, -1, -1>;


// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
// tape: the handle draws from caller-supplied words (numpy parity) instead of the production stream; offered for the
// populate()-shaped trajectory (OUT_TRAJ_RAW8) and for the packed record (OUT_RECORD) -- the modes bench.py times
template <bool WALLS>
inline void launch_duel(bool tape, int out, dim3 g, dim3 blk, size_t sh, hipStream_t st, const Consts &c, const State &s, const RolloutArgs &a, const ObsArgs &o) {
    if (out == OUT_TRAJ_FLAT) hipLaunchKernelGGL((k_rollout_duel<PhiloxRng, OUT_TRAJ_FLAT, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (tape && out == OUT_RECORD16) hipLaunchKernelGGL((k_rollout_duel<TapeRng, OUT_RECORD16, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (out == OUT_RECORD16) hipLaunchKernelGGL((k_rollout_duel<PhiloxRng, OUT_RECORD16, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (tape && out == OUT_RECORD) hipLaunchKernelGGL((k_rollout_duel<TapeRng, OUT_RECORD, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (tape) hipLaunchKernelGGL((k_rollout_duel<TapeRng, OUT_TRAJ_RAW8, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (out == OUT_RECORD) hipLaunchKernelGGL((k_rollout_duel<PhiloxRng, OUT_RECORD, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (out == OUT_NONE) hipLaunchKernelGGL((k_rollout_duel<PhiloxRng, OUT_NONE, WALLS>), g, blk, sh, st, c, s, a, o);
    else if (out == OUT_TRAJ) hipLaunchKernelGGL((k_rollout_duel<PhiloxRng, OUT_TRAJ, WALLS>), g, blk, sh, st, c, s, a, o);
    else hipLaunchKernelGGL((k_rollout_duel<PhiloxRng, OUT_TRAJ_RAW8, WALLS>), g, blk, sh, st, c, s, a, o);
}
template <class SPEC>
void launch_rollout(bool tape, int out, dim3 g, dim3 blk, size_t sh, hipStream_t st, const Consts &c, const State &s, const RolloutArgs &a, const ObsArgs &o) {
    constexpr bool kDuelSpec = !SPEC::kGeneric && SPEC::kA == 2 && SPEC::kJ == 0 && SPEC::kVar == SUSNET_VARIANT_ITG && SPEC::kStaticRoles && SPEC::kFixedOrder;
    if constexpr (kDuelSpec) {
        if ((c.duel_fast || c.duel_walls) && (out == OUT_NONE || out == OUT_TRAJ || out == OUT_TRAJ_RAW8 || out == OUT_RECORD || out == OUT_TRAJ_FLAT || out == OUT_RECORD16)) { // susnet_duel.h
            if (c.duel_walls) launch_duel<true>(tape, out, g, blk, sh, st, c, s, a, o);
            else launch_duel<false>(tape, out, g, blk, sh, st, c, s, a, o);
            return;
        }
    }
    if constexpr (UseSplit<SPEC>::value) {
        if (c.epw == 32 && out != OUT_ANY) { // half-filled waves: two lanes per environment (susnet_swar2.h)
            if (tape && out == OUT_RECORD) hipLaunchKernelGGL((k_rollout_swar2<SPEC, OUT_RECORD, TapeRng>), g, blk, sh, st, c, s, a, o);
            else if (tape) hipLaunchKernelGGL((k_rollout_swar2<SPEC, OUT_TRAJ_RAW8, TapeRng>), g, blk, sh, st, c, s, a, o);
            else if (out == OUT_NONE) hipLaunchKernelGGL((k_rollout_swar2<SPEC, OUT_NONE>), g, blk, sh, st, c, s, a, o);
            else if (out == OUT_TRAJ_RAW8) hipLaunchKernelGGL((k_rollout_swar2<SPEC, OUT_TRAJ_RAW8>), g, blk, sh, st, c, s, a, o);
            else if (out == OUT_TRAJ) hipLaunchKernelGGL((k_rollout_swar2<SPEC, OUT_TRAJ>), g, blk, sh, st, c, s, a, o);
            else hipLaunchKernelGGL((k_rollout_swar2<SPEC, OUT_RECORD>), g, blk, sh, st, c, s, a, o);
            return;
        }
    }
    if constexpr (UseSwar<SPEC>::value) {
        if (out == OUT_TRAJ_FLAT) {
            if constexpr (FlatFor<SPEC>::kOk) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_TRAJ_FLAT>), g, blk, sh, st, c, s, a, o);
        } else if (tape && out == OUT_RECORD) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_RECORD, TapeRng>), g, blk, sh, st, c, s, a, o);
        else if (tape) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_TRAJ_RAW8, TapeRng>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_NONE) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_NONE>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_TRAJ_RAW8) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_TRAJ_RAW8>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_TRAJ) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_TRAJ>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_RECORD) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_RECORD>), g, blk, sh, st, c, s, a, o);
        else if constexpr (SPEC::kA <= 8) hipLaunchKernelGGL((k_rollout_swar<SPEC, OUT_ANY>), g, blk, sh, st, c, s, a, o);
        // (9 .. 12 agents: OUT_ANY -- float observations, partial outputs -- runs the generic kernel, chosen by the host: the general
        // observation writer beside three words per quantity does not fit 256 vector registers)
    } else {
        if (tape && out == OUT_RECORD) { // (SpecCfg2 on a wall map: the table kernel)
            if constexpr (!SPEC::kGeneric && SPEC::kRawF > 0) hipLaunchKernelGGL((k_rollout<SPEC, OUT_RECORD, TapeRng>), g, blk, sh, st, c, s, a, o);
        } else if (tape) hipLaunchKernelGGL((k_rollout<SPEC, OUT_TRAJ_RAW8, TapeRng>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_NONE) hipLaunchKernelGGL((k_rollout<SPEC, OUT_NONE>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_TRAJ_RAW8) hipLaunchKernelGGL((k_rollout<SPEC, OUT_TRAJ_RAW8>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_TRAJ) hipLaunchKernelGGL((k_rollout<SPEC, OUT_TRAJ>), g, blk, sh, st, c, s, a, o);
        else if (out == OUT_RECORD) {
            // the packed record exists for configurations whose whole raw row is known at compile time
            if constexpr (!SPEC::kGeneric && SPEC::kRawF > 0) hipLaunchKernelGGL((k_rollout<SPEC, OUT_RECORD>), g, blk, sh, st, c, s, a, o);
        } else hipLaunchKernelGGL((k_rollout<SPEC, OUT_ANY>), g, blk, sh, st, c, s, a, o);
    }
}
// tape: the handle draws from caller-supplied words (numpy parity) instead of the production stream
template <class SPEC>
void launch_step(bool tape, dim3 g, dim3 blk, size_t sh, hipStream_t st, const Consts &c, const State &s, const StepArgs &a, const ObsArgs &o) {
    if (tape) hipLaunchKernelGGL((k_step<TapeRng, SPEC>), g, blk, sh, st, c, s, a, o);
    else hipLaunchKernelGGL((k_step<PhiloxRng, SPEC>), g, blk, sh, st, c, s, a, o);
}
// (variadic: a specialisation's name may contain commas)
#define SUSNET_DECLARE(...)                                                                                               \
    extern template void launch_rollout<__VA_ARGS__>(bool, int, dim3, dim3, size_t, hipStream_t, const Consts &, const State &, const RolloutArgs &, const ObsArgs &); \
    extern template void launch_step<__VA_ARGS__>(bool, dim3, dim3, size_t, hipStream_t, const Consts &, const State &, const StepArgs &, const ObsArgs &);
#define SUSNET_INSTANTIATE(...)                                                                                           \
    template void launch_rollout<__VA_ARGS__>(bool, int, dim3, dim3, size_t, hipStream_t, const Consts &, const State &, const RolloutArgs &, const ObsArgs &); \
    template void launch_step<__VA_ARGS__>(bool, dim3, dim3, size_t, hipStream_t, const Consts &, const State &, const StepArgs &, const ObsArgs &);

} // namespace susnet
