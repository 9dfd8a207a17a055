// susnet_device.h -- device-side environment logic for gfx950 (CDNA4), one wavefront lane per environment.
//
// Reference behaviour implemented here (paths relative to the reference repo root):
//   reset            src/environment/base.py:251-324, src/environment/tagging.py:62-101
//   sample_actions   src/environment/base.py:326-330
//   step             src/environment/base.py:332-407, src/environment/tagging.py:120-235
//   _agent_step      src/environment/base.py:462-533
//   win conditions   src/environment/base.py:409-460, src/environment/pred_prey.py:78-99
//   _merge_rewards   src/environment/base.py:553-563
//   info counters    src/metrics.py:35-64
//
// Execution model: a workgroup is ONE 64-lane wavefront (64 environments; 32 or 16 in the fused rollout when the
// batch would otherwise leave SIMDs without a wave).  Per-lane scalars (alive / imposter / tag-used / job-done
// bitmasks, t, counters) live in VGPRs.  The per-agent table (cell, tag count, action) and the job cells come in two
// storage flavours behind one interface:
//   LdsStore        [index][lane] columns in LDS: any agent/job count, a lane-varying index is one
//                   conflict-free ds_read_b32 (generic kernels);
//   RegStore<A, J>  byte lanes of packed VGPR words for configurations compiled in (Spec<...>): loops unroll, a
//                   data-dependent index is a shift (per-lane arrays indexed at run time would go to scratch).
// Wave-shared LDS tables hold what is looked up by data-dependent index and is constant: the wall map, the spawn
// cells, the (action, cell) -> next cell move table and the reward table.  No s_barrier anywhere: a lane only
// touches its own column, except the cooperative observation writer, and LDS operations of one wave complete in
// order.  Wave ballots gate the rare work (kill search / resolution, fix / sabotage) under uniform control flow.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/susnet.h"

namespace susnet {

constexpr int kWave = 64;
constexpr int kBlock = 64; // one wave per workgroup (see header comment)

// Action enum values (src/environment/base.py:46-58)
enum : int { ACT_STAY = 0, ACT_UP = 1, ACT_DOWN = 2, ACT_LEFT = 3, ACT_RIGHT = 4, ACT_KILL = 5, ACT_FIX = 6, ACT_SABOTAGE = 7 };
// reward codes: which ASSIGNMENT (base.py:514-515,523,532) an agent's reward slot last received
enum : uint32_t { RC_NONE = 0, RC_KILL = 1, RC_FIX = 2, RC_SAB = 3 };

// per-env flag byte
constexpr uint32_t FLAG_FRESH = 1;    // episode was auto-reset: zero the info counters at the next step
constexpr uint32_t FLAG_CREW_WON = 2; // SusMetrics.CREW_WON latched (metrics.update, base.py:433)
constexpr uint32_t FLAG_IMP_WON = 4;  // SusMetrics.IMPOSTER_WON latched (base.py:444)

// packed agent halfword in HBM: x[0:4) y[4:8) alive[8] imposter[9] tag_used[10] tag_count[11:15)
__device__ __forceinline__ uint32_t pack_agent(uint32_t xy, uint32_t alive, uint32_t imp, uint32_t used, uint32_t cnt) {
    return (xy & 0xffu) | (alive << 8) | (imp << 9) | (used << 10) | ((cnt & 15u) << 11);
}

// Immutable per-handle constants, passed BY VALUE as a kernel argument (lands in SGPRs / kernarg memory).
struct Consts {
    int32_t B, Bp;
    int32_t A, J, N, n_imp, n_crew, variant;
    int32_t max_t, order_random, shuffle_imp, tag_interval;
    int32_t n_valid, auto_reset, nr_imp, nr_crew; // nr_* = length of the role part of agent_action_map
    int32_t epw, dev_tick;                        // environments per wave in the fused rollout; dev_tick: the step counter
                                                  // is read from / advanced in device memory (graph-replayable launches)
    int32_t aw_W, aw_pad;                         // action-stream words a tick owns (see AwLayout)
    uint8_t aw_word[32];                          // draw d (A action draws, then the A - 1 shuffle draws) -> word of the tick
    uint32_t grid_rows[SUSNET_MAX_GRID];          // bit j of row i = grid[i][j]
    uint32_t valid_xy[SUSNET_MAX_GRID * SUSNET_MAX_GRID / 4]; // np.argwhere(grid) order; bytes x | y << 4
    // move_tab[a][cell] = cell after role-relative action a in {STAY, UP, DOWN, LEFT, RIGHT, other}: the whole of
    // move() + _is_valid_position() (base.py:69-79, 548-551, transposed wall lookup included) as one byte lookup
    uint32_t move_tab[6 * 64];
    double dr[8];  // kill, fix, sabotage, time_step, game_end, dead_penalty, vote (reference: Python numbers)
    float fr[8];   // the same as float32 (used by compiled-in kernels when every value is float-exact)
    // reward of one agent as a table over (win: none/crew/imposter, index < n_imposters, dead, assignment code):
    // assignments -> _merge_rewards -> zero fill (base.py:514-515,523,532,553-563,389-390) evaluated on the host
    float rew_tab[48];
    uint64_t seed, env_id_base;
    // 1v1 no-walls fast path (susnet_duel.h): reward bytes over {kill landed, imposter dead, crew dead} per agent
    uint64_t duel_lut[2];
    int32_t duel_fast, duel_walls; // which flavour of the duel kernels serves the handle's fused rollouts: no walls / a wall map (at most one is set)
};
enum : int { RW_KILL = 0, RW_FIX = 1, RW_SAB = 2, RW_TSR = 3, RW_END = 4, RW_DEAD = 5, RW_VOTE = 6 };
template <class RT> __device__ __forceinline__ RT rw(const Consts &c, int k);
template <> __device__ __forceinline__ double rw<double>(const Consts &c, int k) { return c.dr[k]; }
template <> __device__ __forceinline__ float rw<float>(const Consts &c, int k) { return c.fr[k]; }

// Kernel arguments are read with scalar loads where they are first needed; the argument block here is ~2.5 KB (Consts carries
// the tables), i.e. 40 cache lines of 64 B, and every first touch of a line is a miss that the wave waits for on its own
// (the compiler does not hoist the loads above branches).  A kernel that runs once per tick pays those misses back to back:
// measured ~1 us of a 4 us step.  warm_kernargs<BYTES>() touches every line that holds scalars (not the table image, which
// is read with vector loads) in ONE batch at the top of the kernel, so the later loads hit the scalar cache.
constexpr int kTableImageBegin = 0x70, kTableImageEnd = 0x70 + 4 * (16 + 64 + 384); // grid_rows .. move_tab (static_assert below)
template <int OFF, int END>
struct KernargToucher {
    static constexpr bool kInTables = OFF >= ((kTableImageBegin + 63) & ~63) && OFF + 64 <= (kTableImageEnd & ~63);
    template <class P>
    static __device__ __forceinline__ void issue(P p, uint32_t *sink) {
        if constexpr (OFF < END) {
            if constexpr (!kInTables) asm volatile("s_load_dword %0, %1, %2" : "=s"(sink[OFF / 64]) : "s"(p), "n"(OFF));
            KernargToucher<OFF + 64, END>::issue(p, sink);
        }
    }
    // (after the wait: keeps every destination register reserved until its load has landed)
    static __device__ __forceinline__ void retire(const uint32_t *sink) {
        if constexpr (OFF < END) {
            if constexpr (!kInTables) asm volatile("" ::"s"(sink[OFF / 64]));
            KernargToucher<OFF + 64, END>::retire(sink);
        }
    }
};
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    uint32_t sink[(BYTES + 63) / 64];
    KernargToucher<0, BYTES>::issue(__builtin_amdgcn_kernarg_segment_ptr(), sink);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KernargToucher<0, BYTES>::retire(sink);
}

// Device pointers into the caller's state blob (SoA, row stride Bp).
struct State {
    uint32_t *err;      // [1] device error word
    uint64_t *tickw;    // [Bp] device-resident step counter, ONE COPY PER ENVIRONMENT (used when Consts::dev_tick): a lane reads
                        // and advances its own word, so no launch ever reads a word another workgroup is writing
    uint16_t *agent;    // [A][Bp]
    uint8_t *job;       // [J][Bp]  x | y << 4 (constant within an episode)
    uint16_t *jobdone;  // [Bp] bitmask
    uint16_t *t;        // [Bp]
    uint16_t *timer;    // [Bp] tag_reset_timer
    uint8_t *flags;     // [Bp]
    uint64_t *rng;      // [Bp] words consumed
    uint32_t *ep;       // [Bp] resets drawn so far: the index of the env's next reset in the RESET stream (production protocol)
    uint32_t *m_steps;  // [Bp] TOTAL_TIME_STEPS
    uint32_t *m_fix;    // [Bp] COMPLETED_JOBS
    uint32_t *m_sab;    // [Bp] SABOTAGED_JOBS
    uint32_t *m_kv;     // [Bp] kills[0:16) imp_voted[16:24) crew_voted[24:32)
    uint32_t *life;     // [SUSNET_N_LIFETIME][Bp]
    const uint32_t *tape; // TAPE mode: [B][tape_len]
    int64_t tape_len;
};

// The stepping kernels take (Consts, State, <args>, ObsArgs) by value.  A fused rollout needs State's thirteen pointers twice: in
// its prologue (load the environments) and in its epilogue (store them); held in scalar registers across the tick loop they are
// 26 SGPRs that push other values into spill lanes.  The epilogue therefore reads the struct AGAIN from the kernel-argument
// segment, through a pointer the compiler cannot connect to the earlier loads (a handful of scalar loads, once per launch).
constexpr uint32_t kStateArgOffset = (uint32_t)((sizeof(Consts) + alignof(State) - 1) / alignof(State) * alignof(State));
template <class T>
__device__ __forceinline__ T kernarg_reload(uint32_t byte_offset) {
    typedef const __attribute__((address_space(4))) char *kernarg_ptr;
    kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    T out;
    __builtin_memcpy(&out, p + byte_offset, sizeof(T)); // (scalar loads: the address is wave-uniform, the segment constant)
    return out;
}

// Packing of a tick's bounded draws into 32-bit words of the ACTION stream (production protocol; restated in
// oracle/susnet_oracle.c so_action_layout).  The draws are the A action draws in agent order (base.py:326-330) and, with a
// shuffled action order (np.random.shuffle, base.py:372-374), the A - 1 draws k = 1 .. A-1 (range k + 1) that place agent k
// among agents 0 .. k (see ranks_from_stream: a uniform random order, built as turn RANKS).  Consecutive
// draws share a word by nested multiply-shift (digit = hi32(w * n), w = lo32(w * n): joint bias <= product of the ranges
// * 2^-32); a word is closed as soon as the product of the draws' LARGEST possible ranges would pass 2^16, so the packing
// is static.  A tick owns W consecutive words (tick t: words t * W ..), not rounded up to Philox blocks.  The 1v1 game
// (A == 2: ranges 6 and 5 whichever agent is the imposter) packs THREE ticks into one word (30^3 = 27000).
constexpr uint32_t kAwCap = 65536u;
struct AwLayout {
    int W;
    uint8_t word[32];
};
__host__ __device__ constexpr AwLayout make_aw_layout(int A, uint32_t max_range, bool shuffled) {
    AwLayout L = {1, {}};
    int n = 0, word = 0;
    uint32_t prod = 1;
    for (int i = 0; i < A; i++) {
        if (prod * max_range > kAwCap) { word++; prod = 1; }
        prod *= max_range;
        L.word[n++] = (uint8_t)word;
    }
    if (shuffled)
        for (int k = 1; k < A; k++) {
            const uint32_t radix = (uint32_t)k + 1u;
            if (prod * radix > kAwCap) { word++; prod = 1; }
            prod *= radix;
            L.word[n++] = (uint8_t)word;
        }
    L.W = A == 2 ? 1 : word + 1;
    return L;
}
constexpr int kDuelTicksPerWord = 3; // 1v1: ticks served by one action-stream word
constexpr uint32_t kDuelRange = 30u; // 6 * 5: what a 1v1 tick takes out of its word

// Compile-time specialisation of a configuration; -1 = read the value from Consts at run time.
template <int A_, int J_, int VAR_, int ORD_, int SHUF_ = -1, int NI_ = -1>
struct Spec {
    static constexpr bool kGeneric = A_ < 0;
    // action-stream layout known at compile time (agent count, variant and order compiled in)
    static constexpr bool kStaticAw = A_ > 0 && VAR_ >= 0 && ORD_ >= 0;
    static constexpr AwLayout kAw = make_aw_layout(A_ > 0 ? A_ : 1, (VAR_ == SUSNET_VARIANT_ITG ? 6u : 7u) + (VAR_ == SUSNET_VARIANT_TAGGING ? (uint32_t)(A_ > 0 ? A_ - 1 : 0) : 0u), ORD_ > 0);
    __device__ static __forceinline__ int aw_W(const Consts &c) { return kStaticAw ? kAw.W : c.aw_W; }
    __device__ static __forceinline__ int aw_word(const Consts &c, int d) { return kStaticAw ? (int)kAw.word[d] : (int)c.aw_word[d]; }
    static constexpr int kA = A_, kJ = J_, kVar = VAR_, kOrd = ORD_, kShuf = SHUF_, kNI = NI_;
    // flattened_state_size (base.py:230-232; tagging.py:42-60) when the configuration is compiled in
    static constexpr int kRawF = (A_ < 0 || J_ < 0 || VAR_ < 0) ? -1
                                 : 3 * A_ + ((J_ > 0 || VAR_ == SUSNET_VARIANT_TAGGING) ? 3 * (J_ > 0 ? J_ : 0) : 0) +
                                       (VAR_ == SUSNET_VARIANT_TAGGING ? 2 * A_ + 1 : 0);
    // reward arithmetic type: double reproduces the reference's float64 chain for ANY constants; the
    // compiled-in kernels are only selected when all constants are float-exact integers, so float is exact
    using RT = typename std::conditional<(A_ < 0), double, float>::type;
    __device__ static __forceinline__ int A(const Consts &c) { return A_ >= 0 ? A_ : c.A; }
    __device__ static __forceinline__ int J(const Consts &c) { return J_ >= 0 ? J_ : c.J; }
    __device__ static __forceinline__ int variant(const Consts &c) { return VAR_ >= 0 ? VAR_ : c.variant; }
    __device__ static __forceinline__ bool order_random(const Consts &c) { return ORD_ >= 0 ? (ORD_ != 0) : (c.order_random != 0); }
    __device__ static __forceinline__ bool tagging(const Consts &c) { return variant(c) == SUSNET_VARIANT_TAGGING; }
    __device__ static __forceinline__ uint32_t nr_imp(const Consts &c) { return VAR_ >= 0 ? (VAR_ == SUSNET_VARIANT_ITG ? 6u : 7u) : (uint32_t)c.nr_imp; }
    // roles: with shuffle_imposter_index off the imposters are always agents [0, n_imp) (base.py:278)
    __device__ static __forceinline__ int n_imp(const Consts &c) { return NI_ >= 0 ? NI_ : c.n_imp; }
    __device__ static __forceinline__ uint32_t imp(const Consts &, uint32_t env_mask) { return (SHUF_ == 0 && NI_ >= 0) ? ((1u << (NI_ >= 0 ? NI_ : 0)) - 1u) : env_mask; }
    static constexpr bool kStaticRoles = SHUF_ == 0 && NI_ >= 0;
    static constexpr bool kFixedOrder = ORD_ == 0;
    __device__ static __forceinline__ bool shuffle_imp(const Consts &c) { return SHUF_ >= 0 ? (SHUF_ != 0) : (c.shuffle_imp != 0); }
    __device__ static __forceinline__ uint32_t nr_crew(const Consts &c) { return VAR_ >= 0 ? (VAR_ == SUSNET_VARIANT_ITG ? 5u : 6u) : (uint32_t)c.nr_crew; }
};
using GenericSpec = Spec<-1, -1, -1, -1>;
// configurations whose fused rollouts stage the action stream in LDS a group of ticks at a time (the byte-parallel kernels:
// agent count, variant and order compiled in, 3 .. 8 agents); the host reserves the space for exactly these (susnet_capi.hip)
template <class S>
struct HasGroupWords { static constexpr bool value = !S::kGeneric && S::kStaticAw && S::kA >= 3 && S::kA <= 12; };

// ---------------------------------------------------------------------------------------------------
// word sources
// ---------------------------------------------------------------------------------------------------
// Philox4x32-10; key = (seed lo, hi), counter = (block lo, block hi, env lo, env hi), block = cursor >> 2,
// word = out[cursor & 3].  Identical mapping in oracle/susnet_oracle.c (philox_word).
// a ^ b ^ c in ONE instruction (v_bitop3_b32, truth table 0x96): a Philox round is two multiplies and two of these
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
struct PhiloxRng {
    static constexpr bool kNumpy = false;
    uint32_t k0, k1, e0, e1;
    uint64_t cur, blk;
    uint32_t w0, w1, w2, w3;
    bool ovf;

    __device__ __forceinline__ void init(uint64_t seed, uint64_t env, uint64_t cursor) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        e0 = (uint32_t)env; e1 = (uint32_t)(env >> 32);
        cur = cursor; blk = ~0ull; ovf = false;
        w0 = w1 = w2 = w3 = 0;
    }
    __device__ __forceinline__ void gen(uint64_t b) {
        uint32_t c0 = (uint32_t)b, c1 = (uint32_t)(b >> 32), c2 = e0, c3 = e1;
        uint32_t a = k0, d = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, a), n2 = xor3((uint32_t)(p0 >> 32), c3, d);
            c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
            a += 0x9E3779B9u; d += 0xBB67AE85u;
        }
        w0 = c0; w1 = c1; w2 = c2; w3 = c3;
        blk = b;
    }
    __device__ __forceinline__ uint32_t next() {
        uint64_t b = cur >> 2;
        if (__builtin_expect(b != blk, 0)) gen(b);
        uint32_t s = (uint32_t)cur & 3u;
        cur++;
        uint32_t lo = (s & 1u) ? w1 : w0, hi = (s & 1u) ? w3 : w2;
        return (s & 2u) ? hi : lo;
    }
    // production protocol: block-aligned at the start of reset / sample_actions / step, one word per
    // bounded draw (also for n == 1), multiply-shift mapping onto [0, n)
    __device__ __forceinline__ void align() { cur = (cur + 3ull) & ~3ull; }
    __device__ __forceinline__ uint32_t bounded(uint32_t n) { return __umulhi(next(), n); }
    // draw number k (compile-time, counted from the last align()) of a run: no cache check, no select
    __device__ __forceinline__ uint32_t bounded_at(uint32_t n, int k) {
        if ((k & 3) == 0) gen(cur >> 2);
        cur++;
        const uint32_t w = (k & 3) == 0 ? w0 : (k & 3) == 1 ? w1 : (k & 3) == 2 ? w2 : w3;
        return __umulhi(w, n);
    }
};

// ACTION stream of the production protocol (counter word 1 tagged with bit 31), indexed by `tick` = steps taken so
// far: the same for every env stepped in lockstep, so blocks are generated under wave-uniform control flow
// (see sample_actions_env for the word assignment; 1v1: one Philox block per FOUR ticks).
constexpr uint32_t kActionStreamTag = 0x80000000u;
// a third stream of the same key: the acting loop's exploration draws (train.py:359,371: one uniform number per agent and step),
// word tick * A + i; restated by the tests from the oracle's Philox function
constexpr uint32_t kExploreStreamTag = 0x40000000u;
struct ActionStream {
    uint64_t blk;          // block currently held (uniform across the wave)
    uint32_t w0, w1, w2, w3;
    uint32_t rem;          // what the last draw left of its word (the next draw of the same word continues from it)
    uint32_t tag = kActionStreamTag; // which stream of the key (also for objects that only ever gen(): GroupWords::refill); init(kExploreStreamTag) selects the other one
    __device__ __forceinline__ void init(uint32_t stream_tag = kActionStreamTag) { blk = ~0ull; w0 = w1 = w2 = w3 = 0; rem = 0; tag = stream_tag; }
    __device__ __forceinline__ void gen(const PhiloxRng &r, uint64_t b) {
        uint32_t c0 = (uint32_t)b, c1 = (uint32_t)(b >> 32) | tag, c2 = r.e0, c3 = r.e1;
        uint32_t a = r.k0, d = r.k1;
#pragma unroll
        for (int q = 0; q < 10; q++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, a), n2 = xor3((uint32_t)(p0 >> 32), c3, d);
            c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
            a += 0x9E3779B9u; d += 0xBB67AE85u;
        }
        w0 = c0; w1 = c1; w2 = c2; w3 = c3;
        blk = b;
    }
    // word q (compile-time after unrolling / inlining) of the held block
    __device__ __forceinline__ uint32_t at(int q) const { return q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : w3; }
    // word `index` of the stream, run-time index (wave-uniform): generates its block unless it is the one held
    // (two-level select: a 4-way chain on a run-time position is turned into an indexed stack array, i.e. scratch memory)
    __device__ __forceinline__ uint32_t word(const PhiloxRng &r, uint64_t index) {
        const uint64_t b = index >> 2;
        if (b != blk) gen(r, b);
        const uint32_t q = (uint32_t)index & 3u;
        const uint32_t lo = (q & 1u) ? w1 : w0, hi = (q & 1u) ? w3 : w2;
        return (q & 2u) ? hi : lo;
    }
    // word number `g` (compile-time) of the group of words that starts at stream index `base` (a multiple of 4): blocks are
    // generated exactly where a new one starts, the word selection is static
    __device__ __forceinline__ uint32_t word_in_group(const PhiloxRng &r, uint64_t base, int g) {
        if ((g & 3) == 0) gen(r, (base >> 2) + (uint64_t)(g >> 2));
        return at(g & 3);
    }
};

// RESET stream of the production protocol (counter word 1 tagged with bit 29; restated in oracle/susnet_oracle.c
// philox_reset_word): word j of the env's n-th reset is word j & 3 of the block with counter (n, tag | j >> 2, env).  A reset's
// draws are a function of (seed, env, n) alone -- not of how many words earlier episodes consumed -- so a fused rollout can draw an
// environment's NEXT episode ahead of time, all lanes of a wave at once (susnet_kernels.h NextEpisode), instead of sending the whole
// wave through the reset path whenever one lane's episode ends.
constexpr uint32_t kResetStreamTag = 0x20000000u;
struct ResetStream {
    static constexpr bool kNumpy = false;
    uint32_t k0, k1, e0, e1, n;
    uint32_t pos, held; // word position inside this reset; block whose words are held (~0u: none)
    uint32_t w0, w1, w2, w3;
    __device__ __forceinline__ void init(const PhiloxRng &r, uint32_t episode) {
        k0 = r.k0; k1 = r.k1; e0 = r.e0; e1 = r.e1; n = episode;
        pos = 0; held = ~0u;
        w0 = w1 = w2 = w3 = 0;
    }
    __device__ __forceinline__ void gen(uint32_t blk) {
        uint32_t c0 = n, c1 = kResetStreamTag | blk, c2 = e0, c3 = e1;
        uint32_t a = k0, d = k1;
#pragma unroll
        for (int q = 0; q < 10; q++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, a), n2 = xor3((uint32_t)(p0 >> 32), c3, d);
            c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
            a += 0x9E3779B9u; d += 0xBB67AE85u;
        }
        w0 = c0; w1 = c1; w2 = c2; w3 = c3;
        held = blk;
    }
    __device__ __forceinline__ uint32_t next() {
        const uint32_t b = pos >> 2;
        if (__builtin_expect(b != held, 0)) gen(b);
        const uint32_t q = pos & 3u;
        pos++;
        const uint32_t lo = (q & 1u) ? w1 : w0, hi = (q & 1u) ? w3 : w2;
        return (q & 2u) ? hi : lo;
    }
    __device__ __forceinline__ uint32_t bounded(uint32_t m) { return __umulhi(next(), m); }
    // draw number k (compile-time) of the reset: no block check, no select
    __device__ __forceinline__ uint32_t bounded_at(uint32_t m, int k) {
        if ((k & 3) == 0) gen((uint32_t)(k >> 2));
        pos = (uint32_t)k + 1u;
        const uint32_t w = (k & 3) == 0 ? w0 : (k & 3) == 1 ? w1 : (k & 3) == 2 ? w2 : w3;
        return __umulhi(w, m);
    }
    __device__ __forceinline__ void align() {}
};

// Caller-supplied raw words consumed with numpy-legacy semantics (masked rejection, nothing drawn for a
// one-element range): fed numpy's MT19937 output the decisions equal the reference's.
struct TapeRng {
    static constexpr bool kNumpy = true;
    const uint32_t *p;
    int64_t len;
    uint64_t cur;
    bool ovf;

    __device__ __forceinline__ void init(const uint32_t *tape, int64_t n, uint64_t cursor) {
        p = tape; len = n; cur = cursor; ovf = false;
    }
    __device__ __forceinline__ uint32_t next() {
        uint32_t w = 0;
        if ((int64_t)cur < len) w = p[cur];
        else ovf = true;
        cur++;
        return w;
    }
    __device__ __forceinline__ void align() {}
    __device__ __forceinline__ uint32_t bounded_at(uint32_t n, int) { return bounded(n); }
    __device__ __forceinline__ uint32_t bounded(uint32_t n) { // np.random.randint(0, n)
        if (n <= 1u) return 0u;
        uint32_t mx = n - 1u, mask = mx;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = next() & mask; } while (v > mx);
        return v;
    }
};

// ---------------------------------------------------------------------------------------------------
// wave-shared LDS tables + per-lane storage
// ---------------------------------------------------------------------------------------------------
struct Tables {
    const uint32_t *grid;  // [16] wall map rows
    const uint8_t *valid;  // [256] spawn cells, x | y << 4
    const uint8_t *move;   // [6][256] next cell per (action row, cell)
    uint32_t *comp;        // [16] flat observation component list (filled by the kernel when needed)
    const float *rew;      // [48] reward table (non-tagging variants)
    uint8_t *perm;         // [n_valid][64] spawn permutation (TAPE resets only)
    uint32_t *stage;       // observation staging
};

// grid + valid + obs components + move table + rewards, then the turn-rank tables of the byte-parallel rollouts (susnet_swar.h
// RankLut: 120 x 2 words for the first five agents, 336 x 2 words for agents 5 .. 7; filled by the kernel that uses them)
// LDS words: grid 16 | spawn cells 64 | obs components 16 | move table 7 x 64 (rows 0 .. 5 from Consts::move_tab, row 6 = row 5 again:
// the identity, so that FourRoomEnv's imposter list -- index 6 = KILL -- indexes the table as it is) | rewards 64
constexpr uint32_t kMoveRows = 7;
constexpr uint32_t kRankLut1Word = 16 + 64 + 16 + kMoveRows * 64 + 64, kRankLut2Word = kRankLut1Word + 240;
constexpr uint32_t kTableWords = kRankLut2Word + 672;

// LDS byte address of word `word` of the kernel's dynamic LDS (every kernel here uses dynamic LDS only, so the static
// size the intrinsic returns folds to a constant): table lookups through these integer addresses need no base add --
// a pointer derived from the `extern __shared__` symbol costs one `v_add_u32 v, <symbol>, v` per access.
typedef const __attribute__((address_space(3))) uint8_t *lds_u8_ptr;
typedef const __attribute__((address_space(3))) float *lds_f32_ptr;
// (No kernel that uses the tables declares static LDS, so the dynamic area starts at LDS address 0: written as a plain constant,
// the table base folds into the ds_read's offset field; going through __builtin_amdgcn_groupstaticsize() left one
// `v_add_u32 v, 0, v` per lookup in the code, the symbol being resolved only at link time.)
__device__ __forceinline__ constexpr uint32_t lds_table_addr(uint32_t word) { return 4u * word; }
constexpr uint32_t kMoveTableWord = 96, kRewardTableWord = kMoveTableWord + kMoveRows * 64; // (setup_lds)
__device__ __forceinline__ uint32_t lds_move_lookup(uint32_t row_cell) { return *(lds_u8_ptr)(uintptr_t)(lds_table_addr(kMoveTableWord) + row_cell); }
__device__ __forceinline__ float lds_reward_lookup(uint32_t byte_index) { return *(lds_f32_ptr)(uintptr_t)(lds_table_addr(kRewardTableWord) + byte_index); }

// The byte-parallel rollouts (susnet_swar.h GroupWords) stage the action-stream words of one GROUP of ticks behind the tables:
// at most 20 words x 64 environments (12 agents in a shuffled order: 5 words per tick, 4 ticks; cfg4: 3 words per tick -- with two lanes per
// environment 24 words x 32)
constexpr uint32_t kGroupWords = 1280;
// ... and, in the fused rollouts of those kernels, the per-environment cell -> job map behind the group words (susnet_swar.h JobMap:
// one 64-byte row per cell value x | y << 4 of an N x N grid)
__host__ __device__ inline uint32_t lds_jobmap_words(int N) { return ((uint32_t)(((N - 1) << 4) | (N - 1)) + 1u) * 16u; }
__host__ __device__ inline uint32_t lds_core_words(int A, int J, bool generic, bool group_words = false, int jobmap_N = 0) {
    return kTableWords + (generic ? (uint32_t)(2 * A + J) * kBlock : 0u) + (group_words ? kGroupWords : 0u) + (jobmap_N > 0 ? lds_jobmap_words(jobmap_N) : 0u);
}

// first LDS word of the group-words area (carve_lds: behind the tables; compiled-in configurations have no store columns)
template <class S>
__device__ __forceinline__ constexpr uint32_t kGroupWordsWord() { static_assert(!S::kGeneric, "compiled-in configurations"); return kTableWords; }

// generic flavour: [index][lane] columns in LDS (cell in bits 0-7, tag count in bits 8-15 of one word)
struct LdsStore {
    uint32_t *xyp, *jobp, *actp;
    int tid;
    __device__ __forceinline__ void init(uint32_t *base, int A, int J, int tid_) {
        xyp = base; jobp = xyp + A * kBlock; actp = jobp + J * kBlock; tid = tid_;
    }
    __device__ __forceinline__ uint32_t xy(int i) const { return xyp[i * kBlock + tid] & 0xffu; }
    __device__ __forceinline__ void set_xy(int i, uint32_t cell) { xyp[i * kBlock + tid] = (xyp[i * kBlock + tid] & ~0xffu) | cell; }
    __device__ __forceinline__ uint32_t cnt(int i) const { return xyp[i * kBlock + tid] >> 8; }
    __device__ __forceinline__ void set_cnt(int i, uint32_t v) { xyp[i * kBlock + tid] = (xyp[i * kBlock + tid] & 0xffu) | (v << 8); }
    __device__ __forceinline__ void set_agent(int i, uint32_t cell, uint32_t cnt) { xyp[i * kBlock + tid] = cell | (cnt << 8); }
    __device__ __forceinline__ uint32_t job(int j) const { return jobp[j * kBlock + tid]; }
    __device__ __forceinline__ void set_job(int j, uint32_t w) { jobp[j * kBlock + tid] = w; }
    __device__ __forceinline__ uint32_t act(int i) const { return actp[i * kBlock + tid]; }
    __device__ __forceinline__ void set_act(int i, uint32_t a) { actp[i * kBlock + tid] = a; }
};

// compiled-in flavour (A, J <= 8): byte lanes of packed VGPR words -- a data-dependent index is a shift,
// never a memory access (per-lane arrays indexed at run time would be demoted to scratch memory)
template <int N_, int TIER = (N_ > 8 ? 2 : (N_ > 4 ? 1 : 0))>
struct PackedBytes;
template <int N_>
struct PackedBytes<N_, 0> { // up to 4 bytes in one VGPR
    uint32_t w = 0;
    __device__ __forceinline__ uint32_t get(int i) const { return (w >> (8 * i)) & 0xffu; }
    __device__ __forceinline__ void set(int i, uint32_t v) { w = (w & ~(0xffu << (8 * i))) | ((v & 0xffu) << (8 * i)); }
};
template <int N_>
struct PackedBytes<N_, 1> { // 5..8 bytes in a VGPR pair (measured faster than two halves + selects)
    uint64_t w = 0;
    __device__ __forceinline__ uint32_t get(int i) const { return (uint32_t)(w >> (8 * i)) & 0xffu; }
    __device__ __forceinline__ void set(int i, uint32_t v) { w = (w & ~((uint64_t)0xffu << (8 * i))) | ((uint64_t)(v & 0xffu) << (8 * i)); }
};
template <int N_>
struct PackedBytes<N_, 2> { // 9..12 bytes: the pair + one more VGPR (the byte-parallel kernels for 9 .. 12 agents index it statically)
    uint64_t w = 0;
    uint32_t x = 0;
    __device__ __forceinline__ uint32_t get(int i) const { return i < 8 ? (uint32_t)(w >> (8 * (i & 7))) & 0xffu : (x >> (8 * (i & 3))) & 0xffu; }
    __device__ __forceinline__ void set(int i, uint32_t v) {
        const uint64_t nw = (w & ~((uint64_t)0xffu << (8 * (i & 7)))) | ((uint64_t)(v & 0xffu) << (8 * (i & 7)));
        const uint32_t nx = (x & ~(0xffu << (8 * (i & 3)))) | ((v & 0xffu) << (8 * (i & 3)));
        w = i < 8 ? nw : w;
        x = i < 8 ? x : nx;
    }
};

template <int A, int J>
struct RegStore {
    static_assert(A <= 12 && J <= 8, "RegStore packs at most 12 agents / 8 jobs");
    PackedBytes<A> xyw, actw;
    PackedBytes<(J > 0 ? J : 1)> jobw;
    uint32_t cntw = 0; // 4 bits per agent (tag counts: tagging games have at most 8 agents here)
    __device__ __forceinline__ void init(uint32_t *, int, int, int) {}
    __device__ __forceinline__ uint32_t xy(int i) const { return xyw.get(i); }
    __device__ __forceinline__ void set_xy(int i, uint32_t cell) { xyw.set(i, cell); }
    __device__ __forceinline__ uint32_t cnt(int i) const { return i < 8 ? (cntw >> (4 * (i & 7))) & 15u : 0u; }
    __device__ __forceinline__ void set_cnt(int i, uint32_t v) { cntw = i < 8 ? (cntw & ~(15u << (4 * (i & 7)))) | ((v & 15u) << (4 * (i & 7))) : cntw; }
    __device__ __forceinline__ void set_agent(int i, uint32_t cell, uint32_t c) { set_xy(i, cell); set_cnt(i, c); }
    __device__ __forceinline__ uint32_t job(int j) const { return jobw.get(j); }
    __device__ __forceinline__ void set_job(int j, uint32_t v) { jobw.set(j, v); }
    __device__ __forceinline__ uint32_t act(int i) const { return actw.get(i); }
    __device__ __forceinline__ void set_act(int i, uint32_t a) { actw.set(i, a); }
};

// compiled-in agent count: packed VGPR tables; a run-time job count gets the 8-slot table
template <class S>
struct StoreFor { using type = RegStore<S::kA, (S::kJ >= 0 ? S::kJ : 8)>; };
template <>
struct StoreFor<GenericSpec> { using type = LdsStore; };

// Fill the wave-shared tables (all 64 lanes take part) and carve the rest of the dynamic LDS.
// STEP_TABLES: also load the move and reward tables (kernels that step); the others only need grid + spawn cells.
// Two phases, so that a launch-latency-bound kernel (one step per launch) can put its state and action loads between them and
// pay ONE memory round trip for everything: issue() reads the tables from the kernel-argument segment into registers with
// unconditional loads (a lane past a table's end re-reads its last word: no branch, hence no wait, between the loads),
// commit() writes them to LDS.
template <bool STEP_TABLES>
struct TableLoad {
    uint32_t g0, g1, mv[6], rw, cp;
    __device__ __forceinline__ void issue(const Consts &c, const int32_t *comp, int tid) {
        const uint32_t *img = c.grid_rows; // grid_rows[16] and valid_xy[64] are adjacent, in LDS order
        static_assert(offsetof(Consts, valid_xy) == offsetof(Consts, grid_rows) + 4 * SUSNET_MAX_GRID, "table image");
        static_assert(offsetof(Consts, grid_rows) == kTableImageBegin && offsetof(Consts, dr) == kTableImageEnd, "warm_kernargs skips the table image");
        g0 = img[tid];
        g1 = img[64 + (tid & 15)];
        if (STEP_TABLES) {
#pragma unroll
            for (int k = 0; k < 6; k++) mv[k] = c.move_tab[k * kBlock + tid];
            rw = __float_as_uint(c.rew_tab[tid < 48 ? tid : 47]);
        }
        cp = comp ? (uint32_t)comp[tid & 15] : 0u;
    }
    __device__ __forceinline__ void commit(uint32_t *smem, int tid, bool with_comp) const {
        smem[tid] = g0;
        smem[64 + (tid & 15)] = g1;
        if (with_comp) smem[80 + (tid & 15)] = cp;
        if (STEP_TABLES) {
#pragma unroll
            for (int k = 0; k < 6; k++) smem[kMoveTableWord + k * kBlock + tid] = mv[k];
            smem[kMoveTableWord + 6 * kBlock + tid] = mv[5]; // row 6: the identity once more (see kMoveRows)
            smem[kRewardTableWord + (tid < 48 ? tid : 47)] = rw;
        }
    }
};
// smem: the table image -- at LDS address 0 of the workgroup (lds_table_addr: the byte-parallel kernels read it through absolute
// addresses); rest: the wave's own region behind it (k_step / the rollouts: smem + kTableWords; the one-kernel policy tick keeps ONE
// table image for its four waves and gives each wave a region of its own further up)
// JOBMAP: the kernel keeps the cell -> job map behind the group words (fused rollouts of the byte-parallel configurations)
// GROUPWORDS = false: the caller's wave regions hold no group-words area (the one-kernel policy tick never stages the action stream, and
// with the network image beside them its four wave regions are tight: susnet_qnet.h)
template <class S, bool JOBMAP = false, bool GROUPWORDS = true>
__device__ __forceinline__ Tables carve_lds(const Consts &c, uint32_t *smem, uint32_t *rest, int tid, typename StoreFor<S>::type &st) {
    Tables T;
    T.grid = smem;
    T.valid = reinterpret_cast<const uint8_t *>(smem + 16);
    T.comp = smem + 80;
    T.move = reinterpret_cast<const uint8_t *>(smem + kMoveTableWord);
    T.rew = reinterpret_cast<const float *>(smem + kRewardTableWord);
    st.init(rest, c.A, c.J, tid);
    if (S::kGeneric) rest += (2 * c.A + c.J) * kBlock;
    if (HasGroupWords<S>::value && GROUPWORDS) rest += kGroupWords;
    if (JOBMAP) rest += lds_jobmap_words(c.N);
    T.perm = reinterpret_cast<uint8_t *>(rest);
    T.stage = rest;
    return T;
}
template <class S, bool JOBMAP = false>
__device__ __forceinline__ Tables carve_lds(const Consts &c, uint32_t *smem, int tid, typename StoreFor<S>::type &st) {
    return carve_lds<S, JOBMAP>(c, smem, smem + kTableWords, tid, st);
}
__device__ __forceinline__ void wave_lds_publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class S, bool STEP_TABLES = true, bool JOBMAP = false>
__device__ __forceinline__ Tables setup_lds(const Consts &c, uint32_t *smem, int tid, typename StoreFor<S>::type &st) {
    Tables T = carve_lds<S, JOBMAP>(c, smem, tid, st);
    TableLoad<STEP_TABLES> tl;
    tl.issue(c, nullptr, tid);
    tl.commit(smem, tid, false);
    wave_lds_publish();
    return T;
}

// per-lane register state of one environment
struct Env {
    uint32_t alive, imp, used, jd; // bitmasks over agents / jobs
    uint32_t t, timer, flags;
    uint32_t ep;                   // resets drawn so far (index of the next one in the RESET stream)
    uint32_t m_steps, m_fix, m_sab, m_kv;
};

template <class S, class Store>
__device__ __forceinline__ void load_env(const Consts &c, const State &s, Store &st, int64_t b, Env &e) {
    const int A = S::A(c), J = S::J(c);
    e.alive = e.imp = e.used = 0;
#pragma unroll
    for (int i = 0; i < A; i++) {
        uint32_t w = s.agent[(size_t)i * c.Bp + b];
        st.set_agent(i, w & 0xffu, (w >> 11) & 15u);
        e.alive |= ((w >> 8) & 1u) << i;
        e.imp |= ((w >> 9) & 1u) << i;
        e.used |= ((w >> 10) & 1u) << i;
    }
#pragma unroll
    for (int j = 0; j < J; j++) st.set_job(j, s.job[(size_t)j * c.Bp + b]);
    e.jd = s.jobdone[b];
    e.t = s.t[b];
    e.timer = s.timer[b];
    e.flags = s.flags[b];
    e.ep = s.ep[b];
    e.m_steps = s.m_steps[b];
    e.m_fix = s.m_fix[b];
    e.m_sab = s.m_sab[b];
    e.m_kv = s.m_kv[b];
}

template <class S, class Store>
__device__ __forceinline__ void store_env(const Consts &c, const State &s, const Store &st, int64_t b, const Env &e, bool store_jobs) {
    const int A = S::A(c), J = S::J(c);
#pragma unroll
    for (int i = 0; i < A; i++) {
        s.agent[(size_t)i * c.Bp + b] =
            (uint16_t)pack_agent(st.xy(i), (e.alive >> i) & 1u, (S::imp(c, e.imp) >> i) & 1u, (e.used >> i) & 1u, st.cnt(i));
    }
    if (store_jobs) {
#pragma unroll
        for (int j = 0; j < J; j++) s.job[(size_t)j * c.Bp + b] = (uint8_t)st.job(j);
    }
    s.jobdone[b] = (uint16_t)e.jd;
    s.t[b] = (uint16_t)e.t;
    s.timer[b] = (uint16_t)e.timer;
    s.flags[b] = (uint8_t)e.flags;
    s.ep[b] = e.ep;
    s.m_steps[b] = e.m_steps;
    s.m_fix[b] = e.m_fix;
    s.m_sab[b] = e.m_sab;
    s.m_kv[b] = e.m_kv;
}

__device__ __forceinline__ uint32_t nibble(uint64_t v, int i) { return (uint32_t)(v >> (4 * i)) & 15u; }
__device__ __forceinline__ void nibble_swap(uint64_t &v, int i, int j) {
    uint64_t d = (uint64_t)(nibble(v, i) ^ nibble(v, j));
    v ^= (d << (4 * i)) ^ (d << (4 * j)); // cancels when i == j
}
// 32-bit flavour (up to 8 entries): what compiled-in configurations use
__device__ __forceinline__ uint32_t nibble(uint32_t v, int i) { return (v >> (4 * i)) & 15u; }
__device__ __forceinline__ void nibble_swap(uint32_t &v, int i, int j) {
    uint32_t d = nibble(v, i) ^ nibble(v, j);
    v ^= (d << (4 * i)) ^ (d << (4 * j));
}
__device__ __forceinline__ int nth_set_bit(uint32_t m, uint32_t r) {
    for (uint32_t k = 0; k < r; k++) m &= m - 1u;
    return __ffs((int)m) - 1;
}

// the agent order of a step, 4 bits per turn: 32 bits hold up to 8 agents (compiled-in agent counts)
template <class S>
using OrderOf = typename std::conditional<(!S::kGeneric && S::kA <= 8), uint32_t, uint64_t>::type;

// np.random.shuffle / permutation on a nibble-packed list (base.py:374): i = n-1 .. 1, j in [0, i]
// STATIC: n is a compile-time constant and the run starts right after rng.align()
template <bool STATIC, class RNG, class ORD>
__device__ __forceinline__ void shuffle_nibbles(RNG &rng, ORD &v, int n) {
#pragma unroll
    for (int i = n - 1; i >= 1; i--) {
        int j = (int)(STATIC ? rng.bounded_at((uint32_t)i + 1u, n - 1 - i) : rng.bounded((uint32_t)i + 1u));
        nibble_swap(v, i, j);
    }
}

// the same with the loop left ROLLED (9 .. 12 agents on numpy tapes: eleven unrolled rejection loops cost the fused rollouts their last registers)
template <class RNG, class ORD>
__device__ __forceinline__ void shuffle_nibbles_rolled(RNG &rng, ORD &v, int n) {
#pragma clang loop unroll(disable)
    for (int i = n - 1; i >= 1; i--) {
        const int j = (int)rng.bounded((uint32_t)i + 1u);
        nibble_swap(v, i, j);
    }
}

template <class S>
__device__ __forceinline__ uint32_t n_role_actions(const Consts &c, uint32_t is_imp) { return is_imp ? S::nr_imp(c) : S::nr_crew(c); }
template <class S>
__device__ __forceinline__ uint32_t n_actions(const Consts &c, uint32_t is_imp) {
    return n_role_actions<S>(c, is_imp) + (S::tagging(c) ? (uint32_t)(S::A(c) - 1) : 0u);
}

// role-relative index -> Action (base.py:82-99; pred_prey.py:4-19); caller guarantees idx < role count
template <class S>
__device__ __forceinline__ int role_action(const Consts &c, uint32_t is_imp, uint32_t idx) {
    if (idx <= 4u) return (int)idx;
    if (S::variant(c) == SUSNET_VARIANT_ITG) return ACT_KILL;  // imposter idx 5
    if (is_imp) return idx == 5u ? ACT_SABOTAGE : ACT_KILL;    // imposter idx 5, 6
    return ACT_FIX;                                            // crew idx 5
}

// ---------------------------------------------------------------------------------------------------
// reset (base.py:251-324).  Draw order: [imposter indices] -> agent cells -> job cells.
//   TAPE   : numpy semantics incl. the FULL permutation of the valid cells (drawn even when J == 0)
//   PHILOX : same distributions by sequential rejection of duplicates (n_imp + A + J draws, +rare retries)
// ---------------------------------------------------------------------------------------------------
// The placement draws of a reset at STATIC positions of the run that starts at the align(): [NPICK imposter picks], A agent cells,
// J job cells.  With the positions known, the run's Philox blocks are generated where they start and every word is picked
// without a cursor / block check (a reset is paid by the whole wave whenever one lane resets; with a block check -- i.e. control
// flow and an inlined block generator -- at every draw it was 14-16 % of a rollout's cycles).  Both rejection
// rules of the protocol (a pick that hits an earlier imposter is redrawn; a job cell that hits an earlier job is redrawn) keep the
// first draws when those are distinct, which is what the static positions assume: distinct picks -> true, and the jobs walk the
// loop from the first job draw only in a lane that sees a duplicate cell; colliding picks -> false, nothing the caller keeps has
// been touched but the cursor (the caller rewinds and takes the draw-by-draw path).
template <class S, int NPICK, class Store>
__device__ __forceinline__ bool place_static(const Consts &c, const Tables &T, Store &st, Env &e, ResetStream &rng) {
    constexpr int AA = S::kA, JJ = S::kJ;
    if (NPICK > 0) {
        uint32_t imp = 0;
        bool clash = false;
#pragma unroll
        for (int k = 0; k < NPICK; k++) {
            const uint32_t bit = 1u << rng.bounded_at((uint32_t)AA, k);
            clash |= (imp & bit) != 0u;
            imp |= bit;
        }
        if (__builtin_expect(clash, 0)) return false;
        e.imp = imp;
    }
    uint32_t ai[AA], ji[JJ > 0 ? JJ : 1], ac[AA], jc[JJ > 0 ? JJ : 1];
#pragma unroll
    for (int i = 0; i < AA; i++) ai[i] = rng.bounded_at((uint32_t)c.n_valid, NPICK + i);
    const uint32_t pos0 = rng.pos;
#pragma unroll
    for (int j = 0; j < JJ; j++) ji[j] = rng.bounded_at((uint32_t)c.n_valid, NPICK + AA + j);
#pragma unroll
    for (int i = 0; i < AA; i++) ac[i] = T.valid[ai[i]];
#pragma unroll
    for (int j = 0; j < JJ; j++) jc[j] = T.valid[ji[j]];
#pragma unroll
    for (int i = 0; i < AA; i++) st.set_agent(i, ac[i], 0u); // base.py:288-291, with replacement; tagging.py:64: counts cleared
    bool dup = false;
#pragma unroll
    for (int j = 1; j < JJ; j++)
#pragma unroll
        for (int k = 0; k < j; k++) dup |= jc[j] == jc[k];
    if (!dup) {
#pragma unroll
        for (int j = 0; j < JJ; j++) st.set_job(j, jc[j]);
    } else {
        rng.pos = pos0; // (the draw-by-draw walk re-generates its blocks by position)
        rng.held = ~0u;
#pragma unroll
        for (int j = 0; j < JJ; j++) {
            uint32_t xy;
            bool d2;
            do {
                xy = T.valid[rng.bounded((uint32_t)c.n_valid)];
                d2 = false;
#pragma unroll
                for (int k = 0; k < JJ; k++) d2 |= (k < j) && (st.job(k) == xy);
            } while (d2);
            st.set_job(j, xy);
        }
    }
    return true;
}

// rng: the reset's word source -- ResetStream (production protocol: this reset's own words, position 0 onwards) or TapeRng
template <class S, class RNG, class Store>
__device__ __forceinline__ void reset_draws(const Consts &c, const Tables &T, Store &st, int tid, Env &e, RNG &rng) {
    const int A = S::A(c), J = S::J(c);
    if constexpr (!RNG::kNumpy && !S::kGeneric && S::kJ >= 0 && S::kNI >= 1) {
        bool placed;
        if (S::shuffle_imp(c)) {
            placed = place_static<S, S::kNI>(c, T, st, e, rng);
        } else {
            e.imp = (1u << S::n_imp(c)) - 1u; // np.arange(n_imposters), base.py:278
            placed = place_static<S, 0>(c, T, st, e, rng);
        }
        if (__builtin_expect(placed, 1)) {
            e.alive = (1u << A) - 1u; // base.py:301
            e.jd = 0;                 // base.py:302
            e.used = 0;               // tagging.py:64-66
            e.timer = 0;
            e.t = 0;                  // base.py:315
            return;
        }
        rng.pos = 0; // two imposter picks collided: draw by draw, below
        rng.held = ~0u;
    }
    if (S::shuffle_imp(c)) {
        if (RNG::kNumpy) {
            uint64_t perm = 0xFEDCBA9876543210ull;
            shuffle_nibbles<false>(rng, perm, A); // choice(range(A), n_imp, replace=False) == permutation(A)[:n_imp]
            e.imp = 0;
            for (int k = 0; k < c.n_imp; k++) e.imp |= 1u << nibble(perm, k);
        } else {
            e.imp = 0;
            for (int k = 0; k < c.n_imp; k++) {
                uint32_t pick;
                do { pick = rng.bounded((uint32_t)A); } while ((e.imp >> pick) & 1u);
                e.imp |= 1u << pick;
            }
        }
    } else {
        e.imp = (1u << S::n_imp(c)) - 1u; // np.arange(n_imposters), base.py:278
    }
    // roles compiled in: the agent cells are draws 0..A-1 of the run that starts at the align() above, so the
    // words are picked statically (no block-cache check, no select)
    constexpr bool kStaticRun = !RNG::kNumpy && !S::kGeneric && S::kStaticRoles;
#pragma unroll
    for (int i = 0; i < A; i++) { // base.py:288-291, with replacement
        uint32_t cell = kStaticRun ? rng.bounded_at((uint32_t)c.n_valid, i) : rng.bounded((uint32_t)c.n_valid);
        st.set_agent(i, T.valid[cell], 0u); // tagging.py:64: counts cleared
    }
    if (RNG::kNumpy) {
        // base.py:295-299: permutation(n_valid)[:J]
        const int n = c.n_valid;
        for (int k = 0; k < n; k++) T.perm[k * kBlock + tid] = (uint8_t)k;
        for (int i = n - 1; i >= 1; i--) {
            int j = (int)rng.bounded((uint32_t)i + 1u);
            uint8_t a = T.perm[i * kBlock + tid], b2 = T.perm[j * kBlock + tid];
            T.perm[i * kBlock + tid] = b2;
            T.perm[j * kBlock + tid] = a;
        }
        for (int j = 0; j < J; j++) st.set_job(j, T.valid[T.perm[j * kBlock + tid]]);
    } else {
#pragma unroll
        for (int j = 0; j < J; j++) {
            uint32_t xy;
            bool dup;
            do {
                xy = T.valid[rng.bounded((uint32_t)c.n_valid)];
                dup = false;
#pragma unroll
                for (int k = 0; k < J; k++) dup |= (k < j) && (st.job(k) == xy);
            } while (dup);
            st.set_job(j, xy);
        }
    }
    e.alive = (1u << A) - 1u; // base.py:301
    e.jd = 0;                 // base.py:302
    e.used = 0;               // tagging.py:64-66 (counts cleared with xy above)
    e.timer = 0;
    e.t = 0;                  // base.py:315
}

// reset (base.py:251-324).  Production protocol: the draws of the env's reset number e.ep of the RESET stream; the event cursor
// (rng.cur: the kill draws) is not touched.  Numpy tapes: the env's own words, in numpy's order.
template <class S, class RNG, class Store>
__device__ __forceinline__ void reset_env(const Consts &c, const Tables &T, Store &st, int tid, Env &e, RNG &rng) {
    if constexpr (RNG::kNumpy) {
        reset_draws<S>(c, T, st, tid, e, rng);
    } else {
        ResetStream rs;
        rs.init(rng, e.ep);
        reset_draws<S>(c, T, st, tid, e, rs);
        e.ep += 1u;
    }
}

__device__ __forceinline__ void zero_metrics(Env &e) {
    e.m_steps = e.m_fix = e.m_sab = e.m_kv = 0;
    e.flags &= ~(FLAG_FRESH | FLAG_CREW_WON | FLAG_IMP_WON);
}

// base.py:326-330: one randint(len(agent_action_map[i])) per agent in index order.
// TAPE: numpy semantics on the env's own word stream.  PHILOX: word tick * A + i of the action stream.
template <class S, int POS = -1, class Store>
__device__ __forceinline__ void sample_actions_env(const Consts &c, Store &st, const Env &e, TapeRng &rng, ActionStream &, uint64_t) {
    const int A = S::A(c);
    for (int i = 0; i < A; i++) st.set_act(i, rng.bounded(n_actions<S>(c, (S::imp(c, e.imp) >> i) & 1u)));
}
// Production stream (see AwLayout).  POS: the tick's position inside its group of ticks when the caller knows it at
// compile time -- the fused rollout walks the stream in groups of 4 ticks (12 for the 1v1 game: 3 ticks per word) that start
// on a Philox block boundary, so blocks are generated where they start and every word selection is static; -1 = run time
// (any tick; blocks generated on demand).  Actions land in the store; `as.rem` keeps what the last action draw left of its
// word, which the tick's shuffle draws continue (order_from_stream).
template <class S, int POS = -1, class Store>
__device__ __forceinline__ void sample_actions_env(const Consts &c, Store &st, const Env &e, PhiloxRng &rng, ActionStream &as, uint64_t tick) {
    const int A = S::A(c);
    if (A <= 2) {
        uint32_t w;
        if (POS >= 0) {
            constexpr int q = POS >= 0 ? POS / kDuelTicksPerWord : 0, sl = POS >= 0 ? POS % kDuelTicksPerWord : 0;
            if (POS == 0) as.gen(rng, tick / (uint64_t)(4 * kDuelTicksPerWord));
            w = sl == 0 ? as.at(q) : as.rem; // (ticks of a group run in sequence: `rem` is what the previous tick left)
        } else {
            const uint32_t sl = (uint32_t)(tick % (uint64_t)kDuelTicksPerWord);
            w = as.word(rng, tick / (uint64_t)kDuelTicksPerWord);
            w *= sl >= 1u ? kDuelRange : 1u;
            w *= sl >= 2u ? kDuelRange : 1u;
        }
        const uint64_t p = (uint64_t)w * (uint64_t)n_actions<S>(c, S::imp(c, e.imp) & 1u);
        st.set_act(0, (uint32_t)(p >> 32));
        const uint64_t p1 = (uint64_t)(uint32_t)p * (uint64_t)n_actions<S>(c, (S::imp(c, e.imp) >> 1) & 1u);
        if (A == 2) st.set_act(1, (uint32_t)(p1 >> 32));
        as.rem = (uint32_t)p1;
    } else {
        const uint64_t W = (uint64_t)S::aw_W(c);
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < A; i++) {
            const int k = S::aw_word(c, i);
            if (i == 0 || k != S::aw_word(c, i - 1))
                w = (POS >= 0 && S::kStaticAw) ? as.word_in_group(rng, (tick - (uint64_t)(POS >= 0 ? POS : 0)) * W, (POS >= 0 ? POS : 0) * S::kAw.W + k)
                                               : as.word(rng, tick * W + (uint64_t)k);
            const uint64_t p = (uint64_t)w * (uint64_t)n_actions<S>(c, (S::imp(c, e.imp) >> i) & 1u);
            st.set_act(i, (uint32_t)(p >> 32));
            w = (uint32_t)p;
        }
        as.rem = w;
    }
}

// SWAR on bytes packed four to a 32-bit word (values < 0x80; bit 7 of a byte is the lane's flag)
constexpr uint32_t k01 = 0x01010101u, k80 = 0x80808080u, k7f = 0x7f7f7f7fu;
__device__ __forceinline__ uint32_t bcast_byte0(uint32_t v) { return __builtin_amdgcn_perm(v, v, 0u); } // byte 0 of v in all four bytes

// np.random.shuffle of the agent order (base.py:372-374) on the production stream, as turn RANKS: rank[i] = the turn at
// which agent i acts.  Agents are placed one after the other: draw k = 1 .. A-1 (range k + 1, the digits that follow the
// tick's action draws in its action-stream words) is the slot d agent k takes among agents 0 .. k, and every earlier agent
// at slot >= d moves one slot up -- each of the (k + 1)! arrangements of agents 0 .. k equally likely, so the final order
// is a uniform random permutation (same distribution as numpy's Fisher-Yates, different mapping from the words).
// R[w]: byte i % 4 of word i / 4 = rank[i] | 0x80 (the flag bit makes the byte-wise compare below borrow-free).
// have_rem: the caller sampled this tick's actions just before through the same ActionStream (its `rem` is valid);
// otherwise the shared word's remainder is rebuilt from the action ranges.
template <class S, int POS = -1, int NW, class AS = ActionStream>
__device__ __forceinline__ void ranks_from_stream(const Consts &c, uint32_t imp_bits, PhiloxRng &rng, AS &as, uint64_t tick, bool have_rem,
                                                  uint32_t (&R)[NW]) {
    const int A = S::A(c);
    const uint64_t W = (uint64_t)S::aw_W(c);
    uint32_t w = as.rem;
    auto fetch = [&](int k) __attribute__((always_inline)) {
        return (POS >= 0 && S::kStaticAw) ? as.word_in_group(rng, (tick - (uint64_t)(POS >= 0 ? POS : 0)) * W, (POS >= 0 ? POS : 0) * S::kAw.W + k)
                                          : as.word(rng, tick * W + (uint64_t)k);
    };
    if (!have_rem && A > 1 && S::aw_word(c, A) == S::aw_word(c, A - 1)) {
        w = fetch(S::aw_word(c, A));
#pragma unroll
        for (int q = 0; q < 4 * NW; q++)
            if (q < A && S::aw_word(c, q) == S::aw_word(c, A)) w *= n_actions<S>(c, (imp_bits >> q) & 1u);
    }
#pragma unroll
    for (int q = 0; q < NW; q++) R[q] = k80; // agent 0 at slot 0; the other bytes are overwritten when their agent is placed
#pragma unroll
    for (int k = 1; k < 4 * NW; k++) {
        if (k < A) { // (a guard, not a break: the loop must unroll completely -- R[] is indexed statically)
            const int d = A + k - 1; // draw number
            if (S::aw_word(c, d) != S::aw_word(c, d - 1)) w = fetch(S::aw_word(c, d));
            const uint64_t p = (uint64_t)w * (uint64_t)(uint32_t)(k + 1);
            w = (uint32_t)p;
            const uint32_t slot = (uint32_t)(p >> 32);
            const uint32_t sb = bcast_byte0(slot);
#pragma unroll
            for (int q = 0; q <= (k - 1) / 4; q++) R[q] += ((R[q] - sb) >> 7) & k01; // rank >= slot: one up (bytes of agents not placed yet are rewritten below)
            // byte k % 4 of word k / 4 := slot | 0x80
            const uint32_t sel = 0x03020100u ^ ((0x04u ^ (uint32_t)(k & 3)) << (8 * (k & 3))); // identity selector, position k & 3 takes byte 0 of the first operand
            R[k / 4] = __builtin_amdgcn_perm(slot | 0x80u, R[k / 4], sel);
        }
    }
    as.rem = w;
}

// turn order (4 bits per turn: order[rank[i]] = i) from the packed ranks -- what the per-turn kernels walk
template <class S, class ORD, int NW>
__device__ __forceinline__ void order_from_ranks(const Consts &c, const uint32_t (&R)[NW], ORD &order) {
    const int A = S::A(c);
    order = 0;
#pragma unroll
    for (int i = 0; i < 4 * NW; i++) {
        if (i < A) {
            const uint32_t r = (R[i / 4] >> (8 * (i & 3))) & 0x7fu;
            order |= (ORD)i << (4u * r);
        }
    }
}
constexpr int rank_words(int A) { return A > 0 ? (A + 3) / 4 : SUSNET_MAX_AGENTS / 4; }

// ---------------------------------------------------------------------------------------------------
// step
// ---------------------------------------------------------------------------------------------------
// Destination of one lane's row of a per-tick output.  PtrDst = plain address.  BufDst = buffer descriptor
// (wave-uniform base + size, hardware range check) + fixed per-lane byte offset + wave-uniform tick offset in an
// SGPR: the store needs no address arithmetic on the vector unit, the tick offset advances on the scalar unit.
struct PtrDst {
    uint8_t *p;
    __device__ __forceinline__ void st8(uint32_t off, uint32_t v) const { p[off] = (uint8_t)v; }
    __device__ __forceinline__ void st16(uint32_t off, uint32_t v) const { const uint16_t h = (uint16_t)v; __builtin_memcpy(p + off, &h, 2); }
    __device__ __forceinline__ void st32(uint32_t off, uint32_t v) const { __builtin_memcpy(p + off, &v, 4); }
    __device__ __forceinline__ void st64(uint32_t off, uint32_t a, uint32_t b) const { const uint2 w = make_uint2(a, b); __builtin_memcpy(p + off, &w, 8); }
    __device__ __forceinline__ void st128(uint32_t off, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
        const uint4 w = make_uint4(a, b, c, d); __builtin_memcpy(p + off, &w, 16);
    }
};
struct BufDst {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t r;
    uint32_t vo, so;
    __device__ __forceinline__ void st8(uint32_t off, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, r, vo + off, so, 0); }
    __device__ __forceinline__ void st16(uint32_t off, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, r, vo + off, so, 0); }
    __device__ __forceinline__ void st32(uint32_t off, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b32(v, r, vo + off, so, 0); }
    __device__ __forceinline__ void st64(uint32_t off, uint32_t a, uint32_t b) const {
        const u32x2 w = {a, b}; __builtin_amdgcn_raw_buffer_store_b64(w, r, vo + off, so, 0);
    }
    // A store of more than 64 bits reads its data registers over several cycles; a vector write to one of them in the
    // next two issue slots lands before the last lanes (12-15 / 28-31 of each row) were read.  The compiler guards
    // that window only when the scalar offset field is a constant (its hazard table says an SGPR offset needs no
    // guard; on gfx950 that was measured to be wrong: cfg4 records showed the NEXT tick's action digits in one dword
    // of a few waves).  So the tick offset is added on the vector unit here (one v_add per tick and destination,
    // the +off constants go into the instruction's immediate) and the offset field stays 0.
    __device__ __forceinline__ void st128(uint32_t off, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
        const u32x4 w = {a, b, c, d}; __builtin_amdgcn_raw_buffer_store_b128(w, r, vo + so + off, 0, 0);
    }
};
__device__ __forceinline__ BufDst make_buf_dst(void *base, uint64_t bytes, uint32_t lane_off) {
    // raw buffer (stride 0): num_records = bytes; word 3 as in the CDNA4 guide's descriptor recipe
    return BufDst{__builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(uint32_t)bytes, 0x00020000), lane_off, 0u};
}

struct RewardSink {
    void *ptr;       // NULL = drop
    int64_t sa, sb;  // element strides for (agent, env)
    int32_t f64;     // store double instead of float
    template <class RT>
    __device__ __forceinline__ void put(int i, int64_t b, RT r) const {
        if (__builtin_expect(ptr == nullptr, 0)) return;
        int64_t k = (int64_t)i * sa + b * sb;
        if (f64) reinterpret_cast<double *>(ptr)[k] = (double)r;
        else reinterpret_cast<float *>(ptr)[k] = (float)r;
    }
};

struct RewardRowSink : RewardSink { // rollout flavour
    BufDst buf;      // SINK_ON == 2: this env's float32 row of the current tick
    float *regs;     // SINK_ON == 3: the caller's register array (packed-record mode stores them itself)
};

// N consecutive dwords at a 4-byte aligned row destination: 16-byte stores, then an 8- and a 4-byte tail
template <int N, class D>
__device__ __forceinline__ void store_dwords(const D &d, const uint32_t *w) {
    constexpr int k4 = N / 4 * 4;
#pragma unroll
    for (int k = 0; k < k4; k += 4) d.st128(4u * k, w[k], w[k + 1], w[k + 2], w[k + 3]);
    if (N - k4 >= 2) d.st64(4u * k4, w[k4], w[k4 + 1]);
    if ((N - k4) & 1) d.st32(4u * (N - 1), w[N - 1]);
}

// N consecutive 4-byte values / N consecutive bytes at a row destination that is naturally aligned to N's largest
// power-of-two divisor times the element size: widest stores
template <int N, class D>
__device__ __forceinline__ void store_row_f32(const D &d, const float *v) {
    if (N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N; k += 4)
            d.st128(4u * k, __float_as_uint(v[k]), __float_as_uint(v[k + 1]), __float_as_uint(v[k + 2]), __float_as_uint(v[k + 3]));
    } else if (N % 2 == 0) {
#pragma unroll
        for (int k = 0; k < N; k += 2) d.st64(4u * k, __float_as_uint(v[k]), __float_as_uint(v[k + 1]));
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) d.st32(4u * k, __float_as_uint(v[k]));
    }
}
template <int N, class D>
__device__ __forceinline__ void store_row_u8(const D &d, const uint32_t *v) {
    constexpr int kW = N / 4 * 4;
#pragma unroll
    for (int k = 0; k < kW; k += 4)
        d.st32((uint32_t)k, (v[k] & 0xffu) | ((v[k + 1] & 0xffu) << 8) | ((v[k + 2] & 0xffu) << 16) | ((v[k + 3] & 0xffu) << 24));
    if (N - kW >= 2) d.st16((uint32_t)kW, (v[kW] & 0xffu) | ((v[kW + 1] & 0xffu) << 8));
    if ((N - kW) & 1) d.st8((uint32_t)(N - 1), v[N - 1]);
}

__device__ __forceinline__ void set_code(uint32_t &rc, int idx, uint32_t code) { rc = (rc & ~(3u << (2 * idx))) | (code << (2 * idx)); }

// lazy metrics.reset() of an auto-reset episode: the terminal step's info counters stay readable until the next
// step (select form: no branch on the stepping path)
__device__ __forceinline__ void clear_info_if_fresh(Env &e) {
    const bool fresh = (e.flags & FLAG_FRESH) != 0u;
    e.m_steps = fresh ? 0u : e.m_steps;
    e.m_fix = fresh ? 0u : e.m_fix;
    e.m_sab = fresh ? 0u : e.m_sab;
    e.m_kv = fresh ? 0u : e.m_kv;
    e.flags = fresh ? (e.flags & ~(FLAG_FRESH | FLAG_CREW_WON | FLAG_IMP_WON)) : e.flags;
}

// returns error bits (0 = stepped).  Actions are read from the store; rewards go to `sink` at env index b.
// SINK_ON: the reward sink is bound to this env's float32 row of the rollout trajectory [T][B][A]: 1 = by pointer
// (sink.ptr), 2 = by buffer descriptor (sink.buf), 3 = left in the caller's registers (sink.regs); 0 = generic strided put
// order_pre: production stream, shuffled order only -- the tick's agent order (order_from_stream()), 4 bits per turn.
// LAZY_INFO = false: the caller guarantees the env is not FRESH (the fused rollout clears once before its loop and
// zeroes the counters itself at an episode end that is not the launch's last tick)
template <class S, bool VALIDATE, int SINK_ON, bool LAZY_INFO = true, class RNG, class Store, class Sink>
__device__ __forceinline__ uint32_t step_env(const Consts &c, const Tables &T, Store &st, Env &e, RNG &rng, const Sink &sink,
                                             int64_t b, bool &done, bool &trunc, unsigned long long *sg = nullptr, uint64_t order_pre = 0xFEDCBA9876543210ull) {
#ifdef SUSNET_STAMPS
    unsigned long long sprev = __builtin_readcyclecounter();
#define SSTAMP(k) do { unsigned long long tn = __builtin_readcyclecounter(); if (sg) sg[k] += tn - sprev; sprev = tn; } while (0)
#else
#define SSTAMP(k) do {} while (0)
#endif
    const int A = S::A(c), J = S::J(c);
    const bool tagging = S::tagging(c);
    done = false;
    trunc = false;
    if (VALIDATE) { // base.py:357-362 / 379-382: validate before touching anything
        uint32_t space_n = 8u + (tagging ? (uint32_t)A : 0u);
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < A; i++) {
            int32_t a = (int32_t)st.act(i);
            if (a >= (int32_t)space_n) bits |= SUSNET_ERRBIT_ASSERT;
            else if (a < 0 || (uint32_t)a >= n_actions<S>(c, (S::imp(c, e.imp) >> i) & 1u)) bits |= SUSNET_ERRBIT_INDEX;
        }
        if (bits) {
            for (int i = 0; i < A; i++) sink.put(i, b, 0.0f);
            return bits;
        }
    }
    if (LAZY_INFO) clear_info_if_fresh(e);
    e.m_steps += 1; // base.py:366
    using RT = typename S::RT;
    uint32_t rc = 0; // 2-bit reward code per agent
    RT team = 0;

    using OrderT = OrderOf<S>;
    OrderT order = (OrderT)0xFEDCBA9876543210ull; // identity permutation, 4 bits per turn
    rng.align();
    const bool shuffled = S::order_random(c);
    if (shuffled) { // base.py:372-374
        if (RNG::kNumpy) shuffle_nibbles<false>(rng, order, A);
        else order = (OrderT)order_pre; // production protocol: the caller decoded the tick's shuffle draws
    }

    // The per-agent body is written as straight-line predicated code (selects instead of branches): with one
    // wave per SIMD a taken branch is an instruction-fetch bubble nothing else can hide, and a 64-lane wave
    // takes almost every data-dependent branch anyway.  Only genuinely rare work (a multi-candidate kill
    // draw, votes, episode ends) stays behind a branch.
    const bool itg = S::variant(c) == SUSNET_VARIANT_ITG;
    // Compiled-in configurations look every agent's destination cell up FRONT (an agent only ever moves itself, so
    // its cell cannot change before its turn): the A LDS reads are issued back to back instead of exposing one
    // LDS round trip per agent inside the sequential loop.
    PackedBytes<(S::kA > 0 ? S::kA : 1)> dest;
    if (!S::kGeneric) {
#pragma unroll
        for (int i = 0; i < A; i++) {
            const uint32_t ai = st.act(i);
            dest.set(i, T.move[(ai <= 4u ? ai : 0u) * 256u + st.xy(i)]);
        }
    }
    SSTAMP(0);
#pragma unroll
    for (int k = 0; k < A; k++) {
        const int idx = shuffled ? (int)nibble(order, k) : k;
        const uint32_t a = st.act(idx);
        const uint32_t is_imp = (S::imp(c, e.imp) >> idx) & 1u;
        const uint32_t nr = n_role_actions<S>(c, is_imp);
        bool acts = (e.alive >> idx) & 1u; // base.py:477: dead agents do nothing
        if (tagging) {
            // tagging.py:68-75,103-110: k-th OTHER agent ascending; the actor's own aliveness is not checked
            const bool is_tag = a >= nr;
            uint32_t target = a - nr;
            target += (target >= (uint32_t)idx) ? 1u : 0u;
            target = is_tag ? target : 0u;
            if (is_tag && !((e.used >> idx) & 1u) && ((e.alive >> target) & 1u)) {
                st.set_cnt((int)target, st.cnt((int)target) + 1u); // tag_counts[target] += 1
                e.used |= 1u << idx;
            }
            acts = acts && !is_tag;
        }
        // role-relative index -> Action (base.py:82-99; pred_prey.py:4-19)
        const bool is_move = acts && a <= 4u;
        const bool is_kill = acts && is_imp && a == (itg ? 5u : 6u);
        const bool is_sab = acts && !itg && is_imp && a == 5u;
        const bool is_fix = acts && !itg && !is_imp && a == 5u;
        const uint32_t xy = st.xy(idx);
        { // move: base.py:484-487 = one lookup in the (action, cell) table; rows 0 and 5 are the identity
            if (S::kGeneric) {
                const uint32_t arow = is_move ? a : 0u;
                st.set_xy(idx, T.move[arow * 256u + xy]);
            } else {
                st.set_xy(idx, is_move ? dest.get(idx) : xy);
            }
        }
        SSTAMP(1);
        // KILL: base.py:490-515.  Two wave-uniform gates (ballots): the candidate search runs only if some lane's
        // agent attempts a kill this turn, the resolution only if some lane found a victim.
        // (no outer gate with a single crew member compiled in: the search is one compare, and an imposter picks
        // KILL with probability 1/6, so some lane of a 64-env wave nearly always does)
        constexpr bool kOneCrewGate = !S::kGeneric && S::kA == 2;
        // 1v1 with roles and order compiled in (imposter = agent 0, its only possible victim = agent 1): the whole
        // kill is a handful of selects, no gate, no branch
        constexpr bool kDuel = !S::kGeneric && S::kA == 2 && S::kStaticRoles && S::kFixedOrder && !RNG::kNumpy;
        if (kDuel) {
            if (k == 0) {
                const bool hit = is_kill && ((e.alive >> 1) & 1u) && st.xy(1) == xy;
                rng.cur += hit ? 1ull : 0ull;            // production protocol: one (unused) word per kill
                e.m_kv += hit ? 1u : 0u;                  // IMP_KILLED_CREW, base.py:508
                e.alive &= hit ? ~2u : ~0u;               // base.py:511
                rc = hit ? ((rc & ~15u) | (RC_KILL << 2) | RC_KILL) : rc; // base.py:514-515
            }
        } else if (kOneCrewGate || __builtin_amdgcn_ballot_w64(is_kill) != 0ull) {
            uint32_t cm = 0;
            const uint32_t crew = is_kill ? (e.alive & ~S::imp(c, e.imp)) : 0u;
#pragma unroll
            for (int i = 0; i < A; i++) cm |= (((crew >> i) & 1u) && st.xy(i) == xy) ? (1u << i) : 0u;
            if (__builtin_amdgcn_ballot_w64(cm != 0u) != 0ull) {
                const uint32_t nc = (uint32_t)__popc(cm);
                uint32_t r = 0;
                // with a single crew member compiled in there is never more than one candidate: no draw logic at all
                constexpr bool kOneCrew = !S::kGeneric && S::kA == 2;
                if (kOneCrew) {
                    if (!RNG::kNumpy) rng.cur += nc; // production protocol: one (unused) word per kill
                } else if (RNG::kNumpy) {
                    if (__builtin_expect(nc > 1u, 0)) r = rng.bounded(nc); // base.py:497; numpy draws nothing for a single candidate
                } else {
                    // production protocol: one word per kill, its value only matters with several candidates
                    if (__builtin_expect(nc > 1u, 0)) r = rng.bounded(nc);
                    else rng.cur += (nc == 1u) ? 1ull : 0ull;
                }
                const bool hit = nc != 0u;
                const int victim = kOneCrew ? (__ffs((int)(cm | 0x10000u)) - 1) & 15 : (hit ? nth_set_bit(cm, r) : 0);
                e.m_kv += hit ? 1u : 0u;                      // IMP_KILLED_CREW, base.py:508
                e.alive &= ~(hit ? (1u << victim) : 0u);      // base.py:511
                const uint32_t m = hit ? ((3u << (2 * victim)) | (3u << (2 * idx))) : 0u;
                rc = (rc & ~m) | (((RC_KILL << (2 * victim)) | (RC_KILL << (2 * idx))) & m); // base.py:514-515
            }
        }
        SSTAMP(2);
        if (!itg && J > 0 && __builtin_amdgcn_ballot_w64(is_fix || is_sab) != 0ull) { // FIX (base.py:518-524) / SABOTAGE (527-533): first job on the cell (544-546)
            uint32_t jm = 0;
#pragma unroll
            for (int j = 0; j < J; j++) jm |= (st.job(j) == xy) ? (1u << j) : 0u;
            const uint32_t jbit = jm & (0u - jm); // lowest set bit = first job index
            const bool isdone = (e.jd & jbit) != 0u;
            const bool fix = is_fix && jbit != 0u && !isdone;
            const bool sab = is_sab && jbit != 0u && isdone;
            e.jd = fix ? (e.jd | jbit) : (sab ? (e.jd & ~jbit) : e.jd);
            e.m_fix += fix ? 1u : 0u;
            e.m_sab += sab ? 1u : 0u;
            const uint32_t m = (fix || sab) ? (3u << (2 * idx)) : 0u;
            rc = (rc & ~m) | (((fix ? RC_FIX : RC_SAB) << (2 * idx)) & m);
        }
    }

    SSTAMP(3);
    if (tagging) {
        // tagging.py:180: tag_counts *= alive_agents
        for (int i = 0; i < A; i++)
            if (!((e.alive >> i) & 1u)) st.set_cnt(i, 0u);
        e.timer += 1u; // tagging.py:182
        if (__builtin_expect(e.timer >= (uint32_t)c.tag_interval, 0)) { // tagging.py:184-207
            uint32_t best = 0, highest = st.cnt(0);
            for (int i = 1; i < A; i++) { // np.argmax: first maximum
                uint32_t v = st.cnt(i);
                if (v > highest) { highest = v; best = (uint32_t)i; }
            }
            uint32_t quorum = ((uint32_t)__popc(e.alive) + 1u) / 2u;
            if (highest >= quorum) {
                e.alive &= ~(1u << best);
                bool vimp = (S::imp(c, e.imp) >> best) & 1u;
                team += rw<RT>(c, RW_VOTE) * (vimp ? (RT)-1 : (RT)1); // tagging.py:196, sign as coded
                e.m_kv += vimp ? (1u << 16) : (1u << 24);
            }
            for (int i = 0; i < A; i++) st.set_cnt(i, 0u); // tagging.py:237-241
            e.used = 0;
            e.timer = 0;
        }
    }

    // check_win_condition: base.py:409-460 / pred_prey.py:78-99
    uint32_t wsel = 0; // reward-table row of THIS step's outcome: 0 none, 16 crew won, 32 imposters won
    {
        const int alive_imp = __popc(e.alive & S::imp(c, e.imp)), alive_all = __popc(e.alive), done_jobs = __popc(e.jd);
        RT win = 0;
        const RT r_end = rw<RT>(c, RW_END);
        if (S::variant(c) == SUSNET_VARIANT_ITG) {
            if (J != 0 && done_jobs == J) { done = true; e.flags |= FLAG_CREW_WON; win = r_end; wsel = 16u; }
            else if (alive_all - alive_imp == 0) { done = true; e.flags |= FLAG_IMP_WON; win = (RT)-1 * r_end; wsel = 32u; }
        } else {
            if (alive_imp == 0 || done_jobs == J) { done = true; e.flags |= FLAG_CREW_WON; win = r_end; wsel = 16u; }
            else if (alive_all - alive_imp <= alive_imp) { done = true; e.flags |= FLAG_IMP_WON; win = (RT)-1 * r_end; wsel = 32u; }
        }
        team += win;
    }

    SSTAMP(4);
    // per-agent rewards: assignments (codes) -> _merge_rewards (base.py:553-563) -> zero fill (389-390)
    if (!S::kGeneric && !tagging) {
        // compiled-in, non-tagging: one lookup per agent in the host-evaluated table (all reads issued back to back)
        const uint32_t tsel = wsel;
        float rr[S::kA > 0 ? S::kA : 1];
#pragma unroll
        for (int i = 0; i < A; i++) {
            const uint32_t code = (rc >> (2 * i)) & 3u;
            const uint32_t dead = ((e.alive >> i) & 1u) ? 0u : 4u;
            const uint32_t neg = (i < S::n_imp(c)) ? 8u : 0u;
            const float r = T.rew[tsel + neg + dead + code];
            if (SINK_ON) rr[i] = r;
            else sink.put(i, b, r);
        }
        if constexpr (SINK_ON == 3) {
#pragma unroll
            for (int i = 0; i < A; i++) sink.regs[i] = rr[i];
        } else if constexpr (SINK_ON == 2) store_row_f32<(S::kA > 0 ? S::kA : 1)>(sink.buf, rr);
        else if (SINK_ON == 1) store_row_f32<(S::kA > 0 ? S::kA : 1)>(PtrDst{reinterpret_cast<uint8_t *>(sink.ptr)}, rr);
    } else {
#pragma unroll
        for (int i = 0; i < A; i++) {
            uint32_t code = (rc >> (2 * i)) & 3u;
            RT r = tagging ? (RT)1 * rw<RT>(c, RW_TSR) : (RT)0; // tagging.py:162 / base.py:369
            if (code == RC_KILL) r = rw<RT>(c, RW_KILL);
            else if (code == RC_FIX) r = rw<RT>(c, RW_FIX);
            else if (code == RC_SAB) r = (RT)-1 * rw<RT>(c, RW_SAB);
            r += team;
            if (i < S::n_imp(c)) r *= (RT)-1; // indices [:n_imposters], NOT the imposter mask (base.py:559)
            if (!((e.alive >> i) & 1u)) r = rw<RT>(c, RW_DEAD); // base.py:562
            if (!tagging && r == (RT)0) r = rw<RT>(c, RW_TSR);  // base.py:389-390 (tagging.py has no fill)
            if constexpr (SINK_ON == 3) sink.regs[i] = (float)r;
            else if constexpr (SINK_ON == 2) sink.buf.st32(4u * (uint32_t)i, __float_as_uint((float)r));
            else if (SINK_ON == 1) reinterpret_cast<float *>(sink.ptr)[i] = (float)r;
            else sink.put(i, b, r);
        }
    }
    // base.py:392-395: t saturates at max_time_steps - 1
    if (e.t == (uint32_t)(c.max_t - 1)) trunc = true;
    else e.t += 1u;
    SSTAMP(5);
    return 0;
}

// Register-resident flavour for the fused rollout (VGPRs are plentiful at one wave per SIMD): branch-free adds
// at episode end, one flush per launch.
// wave-uniform copy of a 64-bit value every lane holds identically
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    // (the builtin returns int: widen through uint32_t, or a low word with bit 31 set smears into the high one)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

struct LifeAcc {
    uint32_t v[10];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = 0;
    }
    __device__ __forceinline__ void add_episode(const Env &e, bool trunc) {
        v[SUSNET_L_EPISODES] += 1u;
        v[SUSNET_L_CREW_WON] += (e.flags & FLAG_CREW_WON) ? 1u : 0u;
        v[SUSNET_L_IMPOSTER_WON] += (e.flags & FLAG_IMP_WON) ? 1u : 0u;
        v[SUSNET_L_TRUNCATED] += trunc ? 1u : 0u;
        v[SUSNET_L_KILLS] += e.m_kv & 0xffffu;
        v[SUSNET_L_COMPLETED_JOBS] += e.m_fix;
        v[SUSNET_L_SABOTAGED_JOBS] += e.m_sab;
        v[SUSNET_L_IMP_VOTED_OUT] += (e.m_kv >> 16) & 0xffu;
        v[SUSNET_L_CREW_VOTED_OUT] += e.m_kv >> 24;
        v[SUSNET_L_EPISODE_STEPS] += e.m_steps;
    }
    // n_steps: the ticks this launch advanced the environment by (SUSNET_L_ENV_STEPS counts every step taken, finished episode or not)
    __device__ __forceinline__ void flush(const Consts &c, const State &s, int64_t b, uint32_t n_steps) const {
#pragma unroll
        for (int k = 0; k < 10; k++) s.life[(size_t)k * c.Bp + b] += v[k];
        s.life[(size_t)SUSNET_L_ENV_STEPS * c.Bp + b] += n_steps;
    }
};

// episode bookkeeping at an auto-reset: per-env lifetime sums (the multi-GPU metrics reduction input).
// Fire-and-forget atomics on the lane's own words: no load latency on the stepping path.
__device__ __forceinline__ void accumulate_lifetime(const Consts &c, const State &s, int64_t b, const Env &e, bool trunc) {
    uint32_t *L = s.life + b;
    const size_t st = (size_t)c.Bp;
    atomicAdd(&L[SUSNET_L_EPISODES * st], 1u);
    if (e.flags & FLAG_CREW_WON) atomicAdd(&L[SUSNET_L_CREW_WON * st], 1u);
    if (e.flags & FLAG_IMP_WON) atomicAdd(&L[SUSNET_L_IMPOSTER_WON * st], 1u);
    if (trunc) atomicAdd(&L[SUSNET_L_TRUNCATED * st], 1u);
    if (e.m_kv & 0xffffu) atomicAdd(&L[SUSNET_L_KILLS * st], e.m_kv & 0xffffu);
    if (e.m_fix) atomicAdd(&L[SUSNET_L_COMPLETED_JOBS * st], e.m_fix);
    if (e.m_sab) atomicAdd(&L[SUSNET_L_SABOTAGED_JOBS * st], e.m_sab);
    if ((e.m_kv >> 16) & 0xffu) atomicAdd(&L[SUSNET_L_IMP_VOTED_OUT * st], (e.m_kv >> 16) & 0xffu);
    if (e.m_kv >> 24) atomicAdd(&L[SUSNET_L_CREW_VOTED_OUT * st], e.m_kv >> 24);
    atomicAdd(&L[SUSNET_L_EPISODE_STEPS * st], e.m_steps);
}

} // namespace susnet
