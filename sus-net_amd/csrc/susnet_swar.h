// susnet_swar.h -- index-order step for the compiled-in multi-agent games (3..8 agents, up to 8 jobs, FourRoomEnv /
// ImposterTrainingGround / FourRoomEnvWithTagging rules): one lane per environment, every per-agent quantity a BYTE of a
// packed 32-bit word, the whole step written as byte-parallel (SWAR) arithmetic over all agents at once.
//
// Reference behaviour (paths relative to the reference repo root):
//   step             src/environment/base.py:332-407
//   _agent_step      src/environment/base.py:462-533   (move / KILL / FIX / SABOTAGE)
//   win conditions   src/environment/base.py:409-460, src/environment/pred_prey.py:78-99
//   _merge_rewards   src/environment/base.py:553-563, zero fill 389-390
//   tagging          src/environment/tagging.py:68-75,103-110 (tag actions), 148-213 (step), 180-207 (vote), 237-241 (reset)
//
// Why the agent loop of base.py:377-382 can be evaluated in INDEX order although the reference walks a shuffled order:
//   * a move depends on nothing but the agent's own cell and action (base.py:484-487);
//   * FIX / SABOTAGE act on the agent's own (unmoved) cell; two agents only interact when they work on the SAME job in the
//     same step (rare: resolved in turn order behind a wave-uniform branch);
//   * KILL is the one real ordering point: the killer sees every other agent where it stands at the killer's turn (agents
//     with an earlier turn have moved, later ones have not) and its victim, if its own turn comes later, never acts.
//     At most n_imposters kills per step: they are resolved in turn order, each over all agents at once, using the
//     agents' turn RANKS (rank[i] < rank[killer] <=> agent i has already acted).
//   * a TAG action (tagging.py:103-110) moves nothing and reads one thing that depends on the order: whether its target is
//     still alive at the tagger's turn -- i.e. alive after this step's kills, or killed by a killer whose turn comes later.
// Everything else (vote, win check, reward merge, truncation) is order-free.
#pragma once

#include "susnet_device.h"
#include "susnet_obs.h"

namespace susnet {

// f(std::integral_constant<int, I>{}) for I = BEGIN .. END - 1 (susnet_kernels.h has the same helper for the kernels)
template <int BEGIN, int END, class F>
__device__ __forceinline__ void swar_static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        swar_static_for<BEGIN + 1, END>(f);
    }
}

template <class S>
struct UseSwar {
    // (kJ = -1: the job count is read at run time -- at most 8 -- : the FAMILY of byte-parallel kernels, one instantiation per agent
    // count, variant, order and imposter count, serving every job count: susnet_kernels.h SpecFam)
    // (9 .. 12 agents -- three words of agent bytes: FourRoomEnv / ImposterTrainingGround rules; the tag section pairs words by v_perm and stays at 8)
    static constexpr bool value = !S::kGeneric && S::kA >= 3 && S::kJ >= -1 && S::kJ <= 8 && S::kOrd >= 0 &&
                                  ((S::kA <= 8 && (S::kVar == SUSNET_VARIANT_BASE || S::kVar == SUSNET_VARIANT_ITG || S::kVar == SUSNET_VARIANT_TAGGING)) ||
                                   (S::kA <= 12 && (S::kVar == SUSNET_VARIANT_BASE || S::kVar == SUSNET_VARIANT_ITG))) &&
                                  (S::kNI >= 1 && S::kNI <= 3);
};

// 0x80 flags -> 0xff bytes
__device__ __forceinline__ uint32_t ff_from80(uint32_t m) { return (m - (m >> 7)) | m; }
// 0x80 in every byte of x that is zero (exact, no cross-byte carries)
__device__ __forceinline__ uint32_t zero80(uint32_t x) {
    const uint32_t t = (x & k7f) + k7f;
    return __builtin_amdgcn_bitop3_b32(t, x, k80, 0x02); // ~t & ~x & k80 (truth table index = t << 2 | x << 1 | k80)
}
// bytes of a where the 0xff mask m is set, else bytes of b
__device__ __forceinline__ uint32_t sel_bytes(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

// ---- the per-environment cell -> job map of the fused rollouts -------------------------------------------------------------------
// _get_job_at_pos (base.py:544-546) asks "which job, if any, lies on this cell".  Job cells are constant within an episode, so the
// fused rollouts keep the answer in LDS, [cell][environment column] bytes: 0 = no job, 0x80 | j = job j.  Written at launch and by the
// in-launch reset (the old cells cleared, the new ones set: the only writers), read once per agent and tick -- one ds_read_u8
// instead of one byte-compare chain per job and word (that chain was a quarter of cfg3's vector instructions), and the job count
// drops out of the step altogether.  The one-step kernels keep the chain (NoJobMap): a launch per step would have to zero the map
// first, and they are latency-bound anyway.
constexpr uint32_t kJobMapColumns = 64; // bytes per row: one per lane (kernels with two lanes per environment use the first 32)
static_assert(kJobMapColumns == kBlock, "lds_jobmap_words (susnet_device.h) reserves 64-byte rows");
// first LDS word of the map: behind the tables and the group words (carve_lds<S, true>)
constexpr uint32_t kJobMapWord = kTableWords + kGroupWords;
struct NoJobMap {
    static constexpr bool kOn = false;
};
struct JobMap {
    static constexpr bool kOn = true;
    typedef __attribute__((address_space(3))) uint8_t *lds_u8_wptr;
    uint32_t col; // LDS byte address of my environment's column in row 0
    __device__ __forceinline__ void init(uint32_t first_word, int column) { col = lds_table_addr(first_word) + (uint32_t)column; }
    __device__ __forceinline__ uint32_t at(uint32_t cell) const { return *(lds_u8_ptr)(uintptr_t)(col + cell * kJobMapColumns); }
    __device__ __forceinline__ void put(uint32_t cell, uint32_t v) const { *(lds_u8_wptr)(uintptr_t)(col + cell * kJobMapColumns) = (uint8_t)v; }
    // all 64 lanes: zero the whole map (once per launch)
    static __device__ __forceinline__ void clear_all(uint32_t first_word, int N, int tid) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) u32x4 *lds_u4_wptr;
        const uint32_t n16 = lds_jobmap_words(N) / 4u; // (rows * 64 bytes: a multiple of 16)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (uint32_t k = (uint32_t)tid; k < n16; k += kBlock) *(lds_u4_wptr)(uintptr_t)(lds_table_addr(first_word) + 16u * k) = zero;
    }
    // the job cells of the store: set (0x80 | j) / cleared; n_jobs may be a run-time count
    template <class Store>
    __device__ __forceinline__ void set_jobs(const Store &st, int n_jobs) const {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < n_jobs) put(st.job(j), 0x80u | (uint32_t)j);
    }
    template <class Store>
    __device__ __forceinline__ void clear_jobs(const Store &st, int n_jobs) const {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < n_jobs) put(st.job(j), 0u);
    }
};

// ---- the action stream of a fused rollout, one GROUP of ticks at a time -------------------------------------------------------
// A tick owns W consecutive words of the action stream (susnet_device.h AwLayout), a Philox block has four: a group of G ticks
// (4; 8 with two lanes per environment) covers whole blocks.  At the first tick of a group the wave generates the group's blocks
// in ONE rolled loop and parks the words in LDS, [word][environment]; a tick then reads its W words back by position.  The tick
// loop itself is NOT unrolled: one copy of the step and of the in-launch reset path, whatever W is.  (Round 2 unrolled a whole
// group with static word selection: cfg4's kernel was 36 000 instructions / 240 KB with 255 spilled SGPRs, the tagging kernel
// used all 256 VGPRs + 73 AGPRs.)  PAIR: lanes L and L + 32 serve one environment (susnet_swar2.h); lane h generates the blocks
// 2k + h and both lanes read all of them -- the exchange that cost a v_permlane32_swap per word is the LDS round trip itself.
template <int W_, bool PAIR>
struct GroupWords {
    static constexpr int W = W_, G = PAIR ? 8 : 4, NBLK = W * G / 4, kCols = PAIR ? 32 : 64;
    static_assert(W * G * kCols <= (int)kGroupWords && (W * G) % (PAIR ? 8 : 4) == 0, "group size");
    typedef __attribute__((address_space(3))) uint32_t *lds_u32_wptr;
    uint32_t col; // LDS byte address of word 0 of my environment's column
    uint32_t h;   // PAIR: which half of the pair I am
    __device__ __forceinline__ void init(uint32_t first_word, int tid) {
        col = lds_table_addr(first_word) + 4u * (uint32_t)(PAIR ? (tid & 31) : tid);
        h = PAIR ? (uint32_t)tid >> 5 : 0u;
    }
    // all blocks of group number `group` (= absolute tick / G)
    __device__ __forceinline__ void refill(const PhiloxRng &r, uint64_t group) const {
        constexpr int kMine = PAIR ? NBLK / 2 : NBLK; // blocks this lane generates
        const uint64_t b0 = group * (uint64_t)NBLK + (uint64_t)h;
#pragma clang loop unroll(disable)
        for (int k = 0; k < kMine; k++) {
            ActionStream blk;
            blk.gen(r, b0 + (uint64_t)(PAIR ? 2 * k : k));
            const uint32_t a = col + (uint32_t)(PAIR ? 8 * k : 4 * k) * (4u * kCols) + (PAIR ? h * (16u * kCols) : 0u);
            *(lds_u32_wptr)(uintptr_t)(a) = blk.w0;
            *(lds_u32_wptr)(uintptr_t)(a + 4u * kCols) = blk.w1;
            *(lds_u32_wptr)(uintptr_t)(a + 8u * kCols) = blk.w2;
            *(lds_u32_wptr)(uintptr_t)(a + 12u * kCols) = blk.w3;
        }
    }
    // word k (compile-time) of the tick at position `pos` of the group
    __device__ __forceinline__ uint32_t read(uint32_t pos, int k) const {
        typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
        return *(lds_u32_ptr)(uintptr_t)(col + pos * (uint32_t)(W * 4 * kCols) + (uint32_t)(k * 4 * kCols));
    }
};
// the W words of one tick in registers, behind the interface sample_actions_swar / sample_actions_pair / ranks_from_lut expect
// of a word source (they are called with POS = 0: "word g of the group that starts at this tick")
template <int W_>
struct TickWords {
    uint32_t wd[W_];
    uint32_t rem;
    __device__ __forceinline__ uint32_t word_in_group(const PhiloxRng &, uint64_t, int g) const { return wd[g]; }
    __device__ __forceinline__ uint32_t word(const PhiloxRng &, uint64_t) const { return 0u; } // (run-time positions: not used with POS = 0)
};

template <class S>
struct Swar {
    // J: the compiled-in job count, or -1 = run time (then JMAX = 8 slots are carried and every loop over jobs is guarded)
    static constexpr int A = S::kA, J = S::kJ, JMAX = S::kJ >= 0 ? S::kJ : 8, NW = (S::kA + 3) / 4, NI = S::kNI > 0 ? S::kNI : 1, JW = JMAX > 4 ? 2 : 1;
    static constexpr bool kTag = S::kVar == SUSNET_VARIANT_TAGGING;
    static constexpr bool kBase = S::kVar != SUSNET_VARIANT_ITG; // FourRoomEnv action lists (base.py:82-99); tagging.py appends the tag actions
    uint32_t xy[NW];           // cell x | y << 4, one byte per agent
    uint32_t al[NW];           // alive: 0x01 per agent
    // the same flags in the forms the step consumes every tick, kept next to `al` by the (rare) code that changes it -- a kill that
    // landed, a vote that ejected somebody, a reset -- instead of being derived from it tick after tick:
    uint32_t al80[NW];         //   alive: 0x80 per agent
    uint32_t crew80[NW];       //   living crew: 0x80 per agent (base.py:535-542)
    uint32_t ridx[NW];         //   per agent the byte index into the reward table of "no assignment, nobody won": 16 if dead + 32 for indices [:n_imposters] (base.py:559,562)
    uint32_t im80[NW];         // imposter: 0x80 per agent (constant within an episode)
    uint32_t isel[NI];         // v_perm selector that extracts imposter s's byte (zeros elsewhere)
    uint32_t iselb[NW > 2 ? 1 : NI]; // v_perm selector that puts imposter s's byte into ALL four bytes (over the pair {word 1, word 0}); with three
                               // words of agents it is formed from isel where it is used, together with its twin over word 2 (swar_isel_all)
    // (three words of agents -- 9 .. 12 -- keep neither ihot nor nact: both follow from isel / im80 in a few instructions where they
    // are needed, and the fused rollouts of those games have no registers to spare)
    static constexpr bool kLean = NW > 2;
    uint32_t ihot[kLean ? 1 : NI][kLean ? 1 : NW]; // 0x80 at imposter s's byte
    uint32_t jb[JMAX > 0 ? JMAX : 1]; // job cell in all four bytes (constant within an episode)
    uint32_t jobs_obs[JMAX > 4 ? 4 : 2]; // x0 y0 x1 y1 | x2 y2 x3 y3 | ... of the job cells (observation bytes; constant within an episode)
    uint32_t jd[JW];           // completed: 0x01 per job
    uint32_t nact[kLean ? 1 : A]; // len(agent_action_map[i]) (constant within an episode)
    // tagging.py: used_tag_actions (0x01 per agent), tag_counts (one byte per agent), tag_reset_timer
    uint32_t used[NW], cnt[NW], timer;
};

// al80 / crew80 / ridx from al and im80 (see Swar)
template <class S>
__device__ __forceinline__ void swar_refresh_alive(Swar<S> &w) {
    using W = Swar<S>;
    constexpr uint32_t neg32 = (W::NI >= 1 ? 0x20u : 0u) | (W::NI >= 2 ? 0x2000u : 0u) | (W::NI >= 3 ? 0x200000u : 0u); // indices [:n_imposters], NOT the imposter mask (base.py:559)
#pragma unroll
    for (int q = 0; q < W::NW; q++) {
        w.al80[q] = (w.al[q] & k01) << 7;
        w.crew80[q] = w.al80[q] & ~w.im80[q];
        w.ridx[q] = (((w.al[q] & k01) ^ k01) << 4) + (q == 0 ? neg32 : 0u); // (bytes past the last agent read as dead: never looked up)
    }
}

// (Env bitmasks + packed store) -> byte-parallel form.  Runs once per launch and after each reset.
template <class S, class Store>
__device__ __forceinline__ void to_swar(const Consts &c, const Store &st, const Env &e, Swar<S> &w) {
    using W = Swar<S>;
    const uint32_t imp = S::imp(c, e.imp);
#pragma unroll
    for (int q = 0; q < W::NW; q++) { w.xy[q] = 0; w.al[q] = 0; w.im80[q] = 0; w.used[q] = 0; w.cnt[q] = 0; }
    w.timer = e.timer;
#pragma unroll
    for (int s = 0; s < W::NI; s++) {
        w.isel[s] = 0x0c0c0c0cu;
        if constexpr (!W::kLean) {
#pragma unroll
            for (int q = 0; q < W::NW; q++) w.ihot[s][q] = 0;
        }
    }
    uint32_t seen = 0; // imposters met so far (ascending agent index)
#pragma unroll
    for (int i = 0; i < W::A; i++) {
        const int q = i / 4, sh = 8 * (i & 3);
        w.xy[q] |= st.xy(i) << sh;
        w.al[q] |= ((e.alive >> i) & 1u) << sh;
        const uint32_t is = (imp >> i) & 1u;
        w.im80[q] |= (is << 7) << sh;
        if constexpr (!W::kLean) w.nact[i] = S::nr_crew(c) + is + (W::kTag ? (uint32_t)(W::A - 1) : 0u); // imposters have one more action (base.py:82-99, pred_prey.py:4-19); tagging.py:68-75
        if (W::kTag) {
            w.used[q] |= ((e.used >> i) & 1u) << sh;
            w.cnt[q] |= st.cnt(i) << sh;
        }
#pragma unroll
        for (int s = 0; s < W::NI; s++) {
            const bool mine = is && seen == (uint32_t)s;
            w.isel[s] = mine ? (0x0c0c0c00u | (uint32_t)i) : w.isel[s]; // byte i of {word 1, word 0} -> byte 0 (i >= 8: of word 2, see iselb / iselc)
            if constexpr (!W::kLean) w.ihot[s][q] |= mine ? (0x80u << sh) : 0u;
        }
        seen += is;
    }
    swar_refresh_alive<S>(w);
#pragma unroll
    for (int s = 0; s < W::NI; s++) {
        const uint32_t idx = w.isel[s] & 0xffu; // (0x0c: the slot has no imposter)
        if constexpr (W::NW <= 2) w.iselb[s] = idx * k01;
    }
#pragma unroll
    for (int q = 0; q < W::JW; q++) w.jd[q] = 0;
#pragma unroll
    for (int q = 0; q < (W::JMAX > 4 ? 4 : 2); q++) w.jobs_obs[q] = 0;
    const int Jr = S::J(c);
#pragma unroll
    for (int j = 0; j < W::JMAX; j++) {
        const uint32_t cell = j < Jr ? st.job(j) : 0u;
        w.jb[j] = cell * k01;
        w.jd[j / 4] |= (j < Jr ? (e.jd >> j) & 1u : 0u) << (8 * (j & 3));
        w.jobs_obs[j / 2] |= ((cell & 15u) | ((cell >> 4) << 8)) << (16 * (j & 1));
    }
}

template <class S, class Store>
__device__ __forceinline__ void from_swar(const Consts &c, const Swar<S> &w, Store &st, Env &e) {
    using W = Swar<S>;
    e.alive = 0;
    e.jd = 0;
#pragma unroll
    for (int i = 0; i < W::A; i++) {
        const int q = i / 4, sh = 8 * (i & 3);
        st.set_xy(i, (w.xy[q] >> sh) & 0xffu);
        e.alive |= ((w.al[q] >> sh) & 1u) << i;
    }
#pragma unroll
    for (int j = 0; j < W::JMAX; j++) e.jd |= ((w.jd[j / 4] >> (8 * (j & 3))) & 1u) << j; // (slots past the job count hold 0)
    if (W::kTag) {
        e.used = 0;
#pragma unroll
        for (int i = 0; i < W::A; i++) {
            const int q = i / 4, sh = 8 * (i & 3);
            e.used |= ((w.used[q] >> sh) & 1u) << i;
            st.set_cnt(i, (w.cnt[q] >> sh) & 15u);
        }
        e.timer = w.timer;
    }
}

// imposter flag (0 / 1) of agent i
template <class S>
__device__ __forceinline__ uint32_t swar_is_imp(const Swar<S> &w, int i) { return (w.im80[i / 4] >> (8 * (i & 3) + 7)) & 1u; }
// len(agent_action_map[i]): kept per agent, or (kLean) the role list's length from the imposter flag (base.py:82-99, pred_prey.py:4-19)
template <class S>
__device__ __forceinline__ uint32_t swar_nact(const Consts &c, const Swar<S> &w, int i) {
    if constexpr (Swar<S>::kLean) return S::nr_crew(c) + swar_is_imp(w, i);
    else return w.nact[i];
}
template <class S>
__device__ __forceinline__ uint32_t swar_imp_bits(const Swar<S> &w) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < Swar<S>::A; i++) m |= swar_is_imp(w, i) << i;
    return m;
}

// N values below 256, one per register, as bytes of packed words: byte picks (v_perm takes the low bytes of two registers at once)
// instead of a shift and an OR per value
template <int N>
__device__ __forceinline__ void pack_digits(const uint32_t (&d)[N], uint32_t (&out)[(N + 3) / 4]) {
#pragma unroll
    for (int q = 0; q < (N + 3) / 4; q++) {
        const int n = N - 4 * q < 4 ? N - 4 * q : 4;
        if (n == 1) out[q] = d[4 * q];
        else {
            uint32_t lo = __builtin_amdgcn_perm(d[4 * q + 1], d[4 * q], 0x0c0c0400u); // d0 | d1 << 8
            if (n == 3) lo |= d[4 * q + 2] << 16;
            if (n == 4) lo |= __builtin_amdgcn_perm(d[4 * q + 3], d[4 * q + 2], 0x04000c0cu); // d2 << 16 | d3 << 24
            out[q] = lo;
        }
    }
}

// base.py:326-330 on the production stream (see sample_actions_env): the action bytes, packed like every per-agent value
template <class S, int POS = -1, class WT = Swar<S>, class AS = ActionStream>
__device__ __forceinline__ void sample_actions_swar(const Consts &c, const WT &w, PhiloxRng &rng, AS &as, uint64_t tick,
                                                    uint32_t (&act)[Swar<S>::NW]) {
    using W = Swar<S>;
    const uint64_t Wt = (uint64_t)S::kAw.W;
    uint32_t word = 0;
    uint32_t dg[W::A];
#pragma unroll
    for (int i = 0; i < W::A; i++) {
        const int k = S::kAw.word[i];
        if (i == 0 || k != S::kAw.word[i - 1])
            word = POS >= 0 ? as.word_in_group(rng, (tick - (uint64_t)(POS >= 0 ? POS : 0)) * Wt, (POS >= 0 ? POS : 0) * S::kAw.W + k)
                            : as.word(rng, tick * Wt + (uint64_t)k);
        uint32_t n_i;
        if constexpr (std::is_same<WT, Swar<S>>::value) n_i = swar_nact<S>(c, w, i);
        else n_i = w.nact[i];
        const uint64_t p = (uint64_t)word * (uint64_t)n_i;
        dg[i] = (uint32_t)(p >> 32);
        word = (uint32_t)p;
    }
    as.rem = word;
    pack_digits<W::A>(dg, act);
}
template <class S>
__device__ __forceinline__ void sample_actions_swar(const Consts &c, const Swar<S> &w, TapeRng &rng, uint32_t (&act)[Swar<S>::NW]) {
    using W = Swar<S>;
#pragma unroll
    for (int q = 0; q < W::NW; q++) act[q] = 0;
#pragma unroll
    for (int i = 0; i < W::A; i++) act[i / 4] |= rng.bounded(swar_nact<S>(c, w, i)) << (8 * (i & 3));
}

// ranks (byte = rank | 0x80) from a turn order (4 bits per turn): the numpy-parity path shuffles an order list
template <class S, class ORD>
__device__ __forceinline__ void ranks_from_order(ORD order, uint32_t (&R)[Swar<S>::NW]) {
    using W = Swar<S>;
    if constexpr (W::NW <= 2) {
        uint64_t r = 0;
#pragma unroll
        for (int k = 0; k < W::A; k++) r |= (uint64_t)(0x80u | (uint32_t)k) << (8u * nibble(order, k));
        R[0] = (uint32_t)r;
        if (W::NW > 1) R[W::NW - 1] = (uint32_t)(r >> 32);
    } else { // (three words: the word is picked with masks -- a run-time index into R[] would put it in scratch memory)
#pragma unroll
        for (int q = 0; q < W::NW; q++) R[q] = 0u;
#pragma unroll
        for (int k = 0; k < W::A; k++) {
            const uint32_t pos = nibble(order, k), v = (0x80u | (uint32_t)k) << (8u * (pos & 3u));
#pragma unroll
            for (int q = 0; q < W::NW; q++) R[q] |= (pos >> 2) == (uint32_t)q ? v : 0u;
        }
    }
}
template <class S>
__device__ __forceinline__ void identity_ranks(uint32_t (&R)[Swar<S>::NW]) {
#pragma unroll
    for (int q = 0; q < Swar<S>::NW; q++) R[q] = 0x83828180u + 0x04040404u * (uint32_t)q;
}

// ---- turn ranks from tables ---------------------------------------------------------------------------------------------------
// ranks_from_stream (susnet_device.h) places agent k = 1 .. A-1 at slot d_k = the k-th shuffle digit of the tick; consecutive
// digits of one action-stream word are the mixed-radix digits of ONE multiply -- hi32(w * (2*3*..)) = ((d_1 * 3 + d_2) * 4 +
// d_3) * 5 + d_4, exactly, with lo32 the word's remainder -- so the rollouts look the result of the insertions up instead of
// performing them: table 1 (index over d_1 .. d_4) holds the ranks of agents 0 .. 4 among themselves; table 2 (index over
// d_5 .. d_{A-1}) holds the final ranks of agents 5 .. A-1 and, for every intermediate rank 0 .. 4, the final rank it ends up
// at -- applied to table 1's bytes with one v_perm_b32.  Both tables are built in LDS by the kernel's own lanes at launch
// (with the same insertion rule), so they cannot drift from ranks_from_stream, which the one-step kernels keep using.
template <class S>
struct RankLut {
    static constexpr int A = S::kA;
    static constexpr int K1 = A - 1 < 4 ? A - 1 : 4, K2 = A - 1 - K1; // digits served by table 1 / table 2
    static constexpr uint32_t prod(int lo, int hi) { uint32_t p = 1; for (int k = lo; k <= hi; k++) p *= (uint32_t)k; return p; }
    static constexpr uint32_t P1 = prod(2, K1 + 1), P2 = K2 > 0 ? prod(6, A) : 1u;
    static constexpr bool same_word(int d0, int n) {
        for (int d = d0 + 1; d < d0 + n; d++) if (S::kAw.word[d] != S::kAw.word[d0]) return false;
        return true;
    }
    static constexpr bool kOk = S::kOrd > 0 && S::kStaticAw && A >= 2 && A <= 8 && same_word(A, K1) && (K2 == 0 || same_word(A + K1, K2));
    static constexpr int kW1 = K1 + 1 > 4 ? 2 : 1; // words per entry of table 1
};

template <class S>
__device__ __forceinline__ void build_rank_lut(uint32_t *smem, int tid) {
    using L = RankLut<S>;
    for (uint32_t p = (uint32_t)tid; p < L::P1; p += kBlock) {
        uint32_t r[5] = {0, 0, 0, 0, 0}, d[5] = {0, 0, 0, 0, 0};
        uint32_t x = p;
#pragma unroll
        for (int k = L::K1; k >= 1; k--) { d[k] = x % (uint32_t)(k + 1); x /= (uint32_t)(k + 1); } // d_1 is the most significant digit
#pragma unroll
        for (int k = 1; k <= L::K1; k++) {
#pragma unroll
            for (int q = 0; q < k; q++) r[q] += r[q] >= d[k] ? 1u : 0u;
            r[k] = d[k];
        }
        smem[kRankLut1Word + p * L::kW1] = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        if (L::kW1 > 1) smem[kRankLut1Word + p * L::kW1 + 1] = r[4];
    }
    if (L::K2 > 0) {
        for (uint32_t p = (uint32_t)tid; p < L::P2; p += kBlock) {
            uint32_t f[5] = {0, 1, 2, 3, 4}, r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t x = p;
#pragma unroll
            for (int k = L::A - 1; k >= 5; k--) { d[k] = x % (uint32_t)(k + 1); x /= (uint32_t)(k + 1); }
#pragma unroll
            for (int k = 5; k < L::A; k++) {
#pragma unroll
                for (int q = 0; q < 5; q++) f[q] += f[q] >= d[k] ? 1u : 0u;
#pragma unroll
                for (int q = 5; q < k; q++) r[q] += r[q] >= d[k] ? 1u : 0u;
                r[k] = d[k];
            }
            smem[kRankLut2Word + 2 * p] = (f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24)) | k80;
            smem[kRankLut2Word + 2 * p + 1] = (f[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24)) | k80; // (agents that do not exist: rank 0, like the insertion's initial value)
        }
    }
}

// The two table reads of a tick, issued (ranks_lut_issue) and turned into rank bytes (ranks_lut_finish) separately: the fused
// rollouts issue them one tick ahead, in the middle of the previous tick's step, so their LDS round trip is never waited for.
struct RankRaw {
    uint32_t lo, hi, flo, fhi;
};
template <class S, int POS, class AS = ActionStream>
__device__ __forceinline__ RankRaw ranks_lut_issue(PhiloxRng &rng, AS &as, uint64_t tick) {
    using L = RankLut<S>;
    constexpr int A = L::A;
    const uint64_t W = (uint64_t)S::kAw.W;
    auto fetch = [&](int k) __attribute__((always_inline)) {
        return POS >= 0 ? as.word_in_group(rng, (tick - (uint64_t)(POS >= 0 ? POS : 0)) * W, (POS >= 0 ? POS : 0) * S::kAw.W + k)
                        : as.word(rng, tick * W + (uint64_t)k);
    };
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    RankRaw raw = {0u, 0u, 0u, 0u};
    uint32_t w = as.rem; // what the tick's action draws left of their last word
    if (S::kAw.word[A] != S::kAw.word[A - 1]) w = fetch(S::kAw.word[A]);
    const uint64_t p1 = (uint64_t)w * (uint64_t)L::P1;
    w = (uint32_t)p1;
    const uint32_t a1 = lds_table_addr(kRankLut1Word) + (uint32_t)(p1 >> 32) * (4u * L::kW1);
    raw.lo = *(lds_u32_ptr)(uintptr_t)a1;
    if (L::kW1 > 1) raw.hi = *(lds_u32_ptr)(uintptr_t)(a1 + 4u);
    if (L::K2 > 0) {
        if (S::kAw.word[A + L::K1] != S::kAw.word[A + L::K1 - 1]) w = fetch(S::kAw.word[A + L::K1]);
        const uint64_t p2 = (uint64_t)w * (uint64_t)L::P2;
        w = (uint32_t)p2;
        const uint32_t a2 = lds_table_addr(kRankLut2Word) + (uint32_t)(p2 >> 32) * 8u;
        raw.flo = *(lds_u32_ptr)(uintptr_t)a2;
        raw.fhi = *(lds_u32_ptr)(uintptr_t)(a2 + 4u);
    }
    as.rem = w;
    return raw;
}
template <class S, int NW>
__device__ __forceinline__ void ranks_lut_finish(const RankRaw &raw, uint32_t (&R)[NW]) {
    using L = RankLut<S>;
    if (L::K2 == 0) {
        R[0] = raw.lo | k80;
        if (NW > 1) R[NW - 1] = raw.hi | k80;
    } else {
        R[0] = __builtin_amdgcn_perm(raw.fhi, raw.flo, raw.lo);                 // final rank of agents 0 .. 3
        const uint32_t t = __builtin_amdgcn_perm(raw.fhi, raw.flo, raw.hi);     // byte 0: agent 4
        if (NW > 1) R[NW - 1] = __builtin_amdgcn_perm(raw.fhi, t, 0x07060500u); // agent 4 | agents 5 .. 7
    }
}
template <class S, int POS, int NW, class AS = ActionStream>
__device__ __forceinline__ void ranks_from_lut(PhiloxRng &rng, AS &as, uint64_t tick, uint32_t (&R)[NW]) {
    const RankRaw raw = ranks_lut_issue<S, POS>(rng, as, tick);
    ranks_lut_finish<S>(raw, R);
}

// One step.  act: role-relative action bytes (valid for their roles); R: turn ranks (byte = rank | 0x80).
// Rewards go to rr[] (float32: the compiled-in kernels are only selected when every reward constant is float-exact).
// mid: independent work the caller wants done inside the step -- the fused rollouts' sampling of the NEXT tick, in two parts: run(0) the
// action digits, run(1) the turn ranks.  Where it goes: between the COMPARE of a ballot gate and its BRANCH.  At one wave per SIMD a gate
// costs 35-41 cycles when the branch follows its compare directly -- the compare's way to the scalar side -- and ~20 with a dozen
// independent instructions in between (profiles/r05_step_sections_cfg4.md); the compiler sinks a compare it knows down to the branch and
// lifts independent work above one it does not, so the compare is written as assembly and the FIRST operand of the work passes through it
// (tie(part): a register the part reads first).  In the two-lane step (susnet_swar2.h) part 0 sits in the first kill turn's candidate gate,
// part 1 in the job section's gate: cfg4 +1.1 % for the first gate alone (gpurun_out/r05af), +2.6 % for both (r05ag).  The one-lane step
// below runs both parts between its kill and job sections, as before: inside the gates it measured slower (see there).
struct NoMid {
    static constexpr bool kOn = false;
    mutable uint32_t none = 0;
    __device__ __forceinline__ void run(int) const {}
    __device__ __forceinline__ uint32_t &tie(int) const { return none; } // (never reached: the callers test kOn)
};
template <class T> struct MidOf { typedef T type; };
template <class T> struct MidOf<T &> { typedef T type; };
template <class T> struct MidOf<T &&> { typedef T type; };
template <class F>
struct MidParts {
    static constexpr bool kOn = true;
    F &f;
    uint32_t &t0, &t1;
    __device__ __forceinline__ void run(int part) const { f(part); }
    __device__ __forceinline__ uint32_t &tie(int part) const { return part ? t1 : t0; }
};
// ballot(v != 0) with the compare pinned HERE, above whatever reads `tie` next
__device__ __forceinline__ uint64_t early_ballot_nz(uint32_t v, uint32_t &tie) {
    uint64_t m;
    asm volatile("v_cmp_ne_u32_e64 %0, 0, %2" : "=s"(m), "+v"(tie) : "v"(v));
    return m;
}
// jm: the fused rollouts' cell -> job map (JobMap), or NoJobMap = match the job cells by compare chain (one-step kernels).
// check_win = false: the caller guarantees that the state it hands over is not a won one (every tick of a fused rollout but its
// first: the previous tick's check covered it, or a reset replaced it) -- the win rules then run only behind a ballot on "something
// they read changed this step" (a kill landed, a job flipped, a vote ejected somebody), or always where a game is won without
// any of that (FourRoomEnv with no jobs, base.py:430).
// realign (fused rollouts, every tick but the launch's last): a landed kill leaves the event cursor at the start of the next block --
// where the NEXT step's alignment would put it -- so that no tick has to align a cursor that moves once in five hundred steps.
template <class S, class RNG, class MID = NoMid, class JM = NoJobMap>
__device__ __forceinline__ void step_swar(const Consts &c, const Tables &T, Swar<S> &w, Env &e, RNG &rng, const uint32_t (&act)[Swar<S>::NW],
                                          const uint32_t (&R)[Swar<S>::NW], float (&rr)[Swar<S>::A], bool &done, bool &trunc,
                                          unsigned long long *sg = nullptr, MID &&mid = MID(), const JM &jm = JM(), bool check_win = true,
                                          bool realign = false) {
#ifdef SUSNET_STAMPS // diagnostic build only (tools/stamps.py): cycles of the step's sections, one wave
    unsigned long long sprev = __builtin_readcyclecounter();
#define WSTAMP(k) do { unsigned long long tn = __builtin_readcyclecounter(); if (sg) sg[k] += tn - sprev; sprev = tn; } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif
    using W = Swar<S>;
    constexpr int A = W::A, J = W::JMAX, NW = W::NW, NI = W::NI; // (J: job SLOTS; the job count itself is Jn)
    const int Jn = S::J(c);
    static_assert(!W::kTag || NW <= 2, "the tag section pairs the two agent words by v_perm");
    constexpr uint32_t kLive[3] = {A >= 4 ? 0xffffffffu : (1u << (8 * (A & 3))) - 1u, A >= 8 ? 0xffffffffu : (A > 4 ? (1u << (8 * (A & 3))) - 1u : 0u),
                                   A >= 12 ? 0xffffffffu : (A > 8 ? (1u << (8 * (A & 3))) - 1u : 0u)};
    e.m_steps += 1; // base.py:366
    // production protocol: the event cursor is block-aligned at the start of a step.  It only ever moves in the kill tail (one word
    // per landed kill), so the fused rollouts (JobMap flavour) align once per launch and again right after a kill (realign_after_kill)
    if (!JM::kOn) rng.align();

    // ---- action classes (0x80 per agent): alive agents only (base.py:477) --------------------------------------------------
    uint32_t al80[NW], kill80[NW], fix80[NW], sab80[NW], mv80[NW], rows[NW], tag80[NW];
#pragma unroll
    for (int q = 0; q < NW; q++) {
        const uint32_t a = act[q];
        const uint32_t g5 = (a + 0x7b7b7b7bu) & k80, g6 = (a + 0x7a7a7a7au) & k80; // action index >= 5 / >= 6
        al80[q] = w.al80[q];
        tag80[q] = 0;
        if (W::kTag) { // base lists + the tag actions behind them (tagging.py:68-75): crew 6.., imposter 7..
            const uint32_t g7 = (a + 0x79797979u) & k80;
            tag80[q] = (g7 | (g6 & ~w.im80[q])) & kLive[q]; // NOT masked by alive: tagging.py:103-110 never checks the actor
            kill80[q] = g6 & ~g7 & w.im80[q] & al80[q];
            const uint32_t j5 = g5 & ~g6 & al80[q];
            sab80[q] = j5 & w.im80[q];
            fix80[q] = j5 & ~w.im80[q];
            rows[q] = sel_bytes(ff_from80(g5), 0x05050505u, a); // movers keep their row, everybody else reads the identity row
        } else if (W::kBase) { // crew: 5 = FIX; imposter: 5 = SABOTAGE, 6 = KILL (base.py:82-99)
            kill80[q] = g6 & al80[q];
            const uint32_t j5 = g5 & ~g6 & al80[q];
            sab80[q] = j5 & w.im80[q];
            fix80[q] = j5 & ~w.im80[q];
            rows[q] = a; // row of the (action, cell) table: 0..4 = the move actions, 5 and 6 = the identity (susnet_device.h kMoveRows)
        } else { // pred_prey.py:4-19: imposter 5 = KILL, no job actions
            kill80[q] = g5 & al80[q];
            sab80[q] = fix80[q] = 0;
            rows[q] = a;
        }
        mv80[q] = ~g5 & al80[q];
    }
    // ---- destinations: one lookup per agent in the (action, cell) table (move() + _is_valid_position(), base.py:69-79, 548-551)
    uint32_t dest[NW];
    {
        uint32_t d[A];
#pragma unroll
        for (int i = 0; i < A; i++) {
            // address = row << 8 | cell: byte i of `rows` and of `xy`
            const uint32_t sel = 0x0c0c0000u | ((4u + (uint32_t)(i & 3)) << 8) | (uint32_t)(i & 3);
#ifdef SUSNET_EXP_NO_DEST_LDS // diagnostic builds only (tools/build_variant.sh): what the lookup's round trip costs -- WRONG results
            d[i] = __builtin_amdgcn_perm(rows[i / 4], w.xy[i / 4], sel) & 0xffu;
#else
            d[i] = lds_move_lookup(__builtin_amdgcn_perm(rows[i / 4], w.xy[i / 4], sel));
#endif
        }
#pragma unroll
        for (int q = 0; q < NW; q++) dest[q] = 0;
#pragma unroll
        for (int i = 0; i < A; i++) dest[i / 4] |= d[i] << (8 * (i & 3));
    }
    // ... and, with the cell -> job map, the job under every agent (0x80 | j, or 0): job actors do not move, so the cell an agent
    // stands on NOW is the one its FIX / SABOTAGE would act on; issued here, used after the kill section
    uint32_t jobat[NW];
#pragma unroll
    for (int q = 0; q < NW; q++) jobat[q] = 0;
    if constexpr (JM::kOn) {
        if (W::kBase) {
            uint32_t mj[A];
#pragma unroll
#ifdef SUSNET_EXP_NO_JOBMAP_LDS
            for (int i = 0; i < A; i++) mj[i] = ((w.xy[i / 4] >> (8 * (i & 3))) & 0xffu) == 0x33u ? 0x80u : 0u;
#else
            for (int i = 0; i < A; i++) mj[i] = jm.at((w.xy[i / 4] >> (8 * (i & 3))) & 0xffu);
#endif
#pragma unroll
            for (int i = 0; i < A; i++) jobat[i / 4] |= mj[i] << (8 * (i & 3));
        }
    }
    // positions if every living mover moved (kills below may cancel a victim's move)
    uint32_t newt[NW];
#pragma unroll
    for (int q = 0; q < NW; q++) newt[q] = sel_bytes(ff_from80(mv80[q]), dest[q], w.xy[q]);

    WSTAMP(0);
    // ---- KILL (base.py:490-515), imposters in turn order --------------------------------------------------------------------
    uint32_t kc80[NW], pend80[NW]; // killers that landed a kill; victims killed before their own turn
    uint32_t vk80[NI][NW], gek80[NI][NW]; // per kill turn: the victim; the agents that had not acted yet (tagging only)
    uint32_t idx4[NW];             // reward-table byte index per agent (see the reward section): the episode's base + what this step adds
#pragma unroll
    for (int q = 0; q < NW; q++) {
        kc80[q] = pend80[q] = 0;
        idx4[q] = w.ridx[q];
    }
#pragma unroll
    for (int it = 0; it < NI; it++)
#pragma unroll
        for (int q = 0; q < NW; q++) vk80[it][q] = gek80[it][q] = 0;
#ifndef SUSNET_EXP_SKIP_KILL // diagnostic builds only (tools/build_variant.sh, profiles/r04_step_sections.md): a section's share of the tick -- WRONG results
    {
        const uint64_t cur0 = rng.cur; // the step's (aligned) event cursor: a landed kill takes word cur0 + kills landed before it
        uint32_t kb[NI], rb[NI], cb[NI]; // per imposter slot, in ALL four bytes: kill flag (0x80 / 0), rank | 0x80, cell
#pragma unroll
        for (int s = 0; s < NI; s++) {
            if constexpr (NW > 2) { // imposter s among agents 0 .. 7: its byte of the pair {word 1, word 0}; among 8 .. 11: of word 2 (the other
                                    // selector yields zeros; 0x0c0c0c0c: no imposter in the slot)
                const uint32_t idx = w.isel[s] & 0xffu;
                const uint32_t sb = idx < 8u ? idx * k01 : 0x0c0c0c0cu, sc = (idx >= 8u && idx < 12u) ? (idx - 8u) * k01 : 0x0c0c0c0cu;
                kb[s] = __builtin_amdgcn_perm(kill80[1], kill80[0], sb) | __builtin_amdgcn_perm(0u, kill80[NW > 2 ? 2 : 0], sc);
                rb[s] = __builtin_amdgcn_perm(R[1], R[0], sb) | __builtin_amdgcn_perm(0u, R[NW > 2 ? 2 : 0], sc);
                cb[s] = __builtin_amdgcn_perm(w.xy[1], w.xy[0], sb) | __builtin_amdgcn_perm(0u, w.xy[NW > 2 ? 2 : 0], sc);
            } else {
                kb[s] = __builtin_amdgcn_perm(NW > 1 ? kill80[NW > 1 ? 1 : 0] : 0u, kill80[0], w.iselb[s]);
                rb[s] = __builtin_amdgcn_perm(NW > 1 ? R[NW > 1 ? 1 : 0] : 0u, R[0], w.iselb[s]);
                cb[s] = __builtin_amdgcn_perm(NW > 1 ? w.xy[NW > 1 ? 1 : 0] : 0u, w.xy[0], w.iselb[s]);
            }

        }
        bool second_first = false; // two imposters: the one with the earlier turn kills first
        if (NI == 2) second_first = kb[1] != 0u && (kb[0] == 0u || rb[1] < rb[0]);
        // three imposters: the attempting ones in turn order -- turn `it` goes to the slot with the it-th smallest key (rank of an
        // attempting imposter, 0x100 + slot for the others: they come last and do nothing)
        uint32_t key3[3] = {0u, 0u, 0u};
        if (NI == 3) {
#pragma unroll
            for (int s = 0; s < 3; s++) key3[s] = kb[s < NI ? s : 0] != 0u ? (rb[s < NI ? s : 0] & 0x7fu) : 0x100u + (uint32_t)s;
        }
        // (one copy of the body per kill turn, made by the template machinery: `#pragma unroll` is a request the optimiser may turn
        // down -- silently under -Wno-pass-failed --, and a loop left rolled indexes vk80[it] / gek80[it] at run time, i.e. in scratch
        // memory: that is what the three-imposter kernels did at first)
        swar_static_for<0, NI>([&](auto itc) __attribute__((always_inline)) {
            constexpr int it = decltype(itc)::value;
            constexpr int s0 = it, s1 = NI - 1 - it; // two imposters: slot if the natural order holds / if it is swapped
            uint32_t kbi, rbi, cbi, hot_of[NW];
            if (NI == 3) {
                const uint32_t m = key3[0] < key3[1] ? (key3[0] < key3[2] ? 0u : 2u) : (key3[1] < key3[2] ? 1u : 2u); // the slot whose turn it is
                // (picked with masks, not with selects: a select between elements of one array is turned into ONE load with a selected
                // index -- a run-time index into `w`, which then lives in scratch memory)
                const uint32_t p0 = 0u - (m == 0u ? 1u : 0u), p1 = 0u - (m == 1u ? 1u : 0u), p2 = 0u - (m == 2u ? 1u : 0u);
                constexpr int i1 = NI > 1 ? 1 : 0, i2 = NI > 2 ? 2 : 0;
                kbi = (kb[0] & p0) | (kb[i1] & p1) | (kb[i2] & p2);
                rbi = (rb[0] & p0) | (rb[i1] & p1) | (rb[i2] & p2);
                cbi = (cb[0] & p0) | (cb[i1] & p1) | (cb[i2] & p2);
                if constexpr (W::kLean) {
                    const uint32_t idx = ((w.isel[0] & p0) | (w.isel[i1] & p1) | (w.isel[i2] & p2)) & 0xffu;
#pragma unroll
                    for (int q = 0; q < NW; q++) hot_of[q] = (idx >> 2) == (uint32_t)q ? 0x80u << (8u * (idx & 3u)) : 0u;
                } else {
#pragma unroll
                    for (int q = 0; q < NW; q++) hot_of[q] = (w.ihot[0][q] & p0) | (w.ihot[W::kLean ? 0 : i1][q] & p1) | (w.ihot[W::kLean ? 0 : i2][q] & p2);
                }
                key3[0] |= p0 & 0x200u; // (taken)
                key3[1] |= p1 & 0x200u;
                key3[2] |= p2 & 0x200u;
            } else {
                kbi = second_first ? kb[s1] : kb[s0];
                rbi = second_first ? rb[s1] : rb[s0];
                cbi = second_first ? cb[s1] : cb[s0];
                if constexpr (W::kLean) {
                    const uint32_t idx = (second_first ? w.isel[s1] : w.isel[s0]) & 0xffu;
#pragma unroll
                    for (int q = 0; q < NW; q++) hot_of[q] = (idx >> 2) == (uint32_t)q ? 0x80u << (8u * (idx & 3u)) : 0u;
                } else {
#pragma unroll
                    for (int q = 0; q < NW; q++) hot_of[q] = second_first ? w.ihot[W::kLean ? 0 : s1][q] : w.ihot[W::kLean ? 0 : s0][q];
                }
            }
            // (no ballot on "somebody attempts" for the first kill turn: in a wave of 64 environments somebody nearly always does;
            // the second turn only has work where BOTH imposters attempt)
            if (it > 0 && __builtin_amdgcn_ballot_w64(kbi != 0u) == 0ull) return;
            const uint32_t tb = rbi & k7f;
            uint32_t ge80[NW], cand[NW];
            uint32_t nc = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                ge80[q] = (R[q] - tb) & k80; // rank >= the killer's: has not acted yet (the killer itself included)
                const uint32_t pos = sel_bytes(ff_from80(ge80[q]), w.xy[q], newt[q]);
                cand[q] = zero80(pos ^ cbi) & w.crew80[q] & kbi; // living crew NOW on the killer's cell (base.py:535-542), if it attempts
                nc += (uint32_t)__popc(cand[q]);
            }
            // A crew member on the killer's cell is rare (a fraction of a percent per environment): everything below the candidate
            // search sits behind a ballot.
            if (__builtin_amdgcn_ballot_w64(nc != 0u) == 0ull) return;
            // base.py:497: uniform among the candidates (ascending agent index).  With one candidate -- nearly always --
            // the victim is the lowest set flag; several candidates (rare) are handled behind a wave-uniform branch.
            uint32_t v80[NW];
            {
                uint32_t seen = 0u; // (0 / all ones) a lower word had a candidate
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    v80[q] = (cand[q] & (0u - cand[q])) & ~seen;
                    seen |= 0u - (cand[q] != 0u ? 1u : 0u);
                }
            }
            uint32_t before = 0; // kills of THIS step that this environment landed already
#pragma unroll
            for (int q = 0; q < NW; q++) before += (uint32_t)__popc(kc80[q]);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(nc > 1u) != 0ull, 0)) {
                if (nc > 1u) {
                    if (!RNG::kNumpy) rng.cur = cur0 + (uint64_t)before; // production protocol: one word per landed kill, its value only matters here
                    const uint32_t r = rng.bounded(nc); // numpy draws nothing for a single candidate: only here
                    uint32_t cw[NW];
#pragma unroll
                    for (int q = 0; q < NW; q++) cw[q] = cand[q];
                    for (uint32_t k = 0; k < r; k++) { // drop the lowest candidate (ascending agent index), r times
                        uint32_t dropped = 0u;
#pragma unroll
                        for (int q = 0; q < NW; q++) {
                            const uint32_t here = (cw[q] != 0u ? 1u : 0u) & ~dropped;
                            cw[q] = here ? (cw[q] & (cw[q] - 1u)) : cw[q];
                            dropped |= here;
                        }
                    }
                    uint32_t seen = 0u;
#pragma unroll
                    for (int q = 0; q < NW; q++) {
                        v80[q] = (cw[q] & (0u - cw[q])) & ~seen;
                        seen |= 0u - (cw[q] != 0u ? 1u : 0u);
                    }
                }
            }
            const bool hit = nc != 0u;
            if (!RNG::kNumpy) rng.cur = hit ? (realign ? cur0 + 4ull : cur0 + (uint64_t)before + 1ull) : rng.cur;
            e.m_kv += hit ? 1u : 0u; // IMP_KILLED_CREW, base.py:508
#pragma unroll
            for (int q = 0; q < NW; q++) {
                w.al[q] &= ~(v80[q] >> 7);                                      // base.py:511
                w.al80[q] &= ~v80[q];
                w.crew80[q] &= ~v80[q];
                w.ridx[q] += v80[q] >> 3;                                       // the victim's rewards come from the "dead" rows from now on (base.py:562)
                const uint32_t hot = hit ? hot_of[q] : 0u;
                kc80[q] |= hot;                                                 // base.py:514-515 (the victim's slot ends as dead_penalty)
                idx4[q] += (v80[q] >> 3) + (hot >> 5);                          // RC_KILL * 4 for the killer
                pend80[q] |= v80[q] & ge80[q];                                  // killed before its own turn: it never acts
                if (W::kTag) { vk80[it][q] = v80[q]; gek80[it][q] = ge80[q]; }
            }
        });
    }
#endif
    // final positions: a victim that had not acted yet stays where it was
#pragma unroll
    for (int q = 0; q < NW; q++) w.xy[q] = sel_bytes(ff_from80(pend80[q]), w.xy[q], newt[q]);
    // (both parts here, between the kill section and the job section: inside the gates, as in susnet_swar2.h, the one-lane kernels came out
    // 1.6-1.8 % SLOWER -- cfg3 127.4 -> 125.5 G, tag5 51.0 -> 50.1 G, gpurun_out/r05ag -- where the two-lane kernel gained 2.6 %)
    mid.run(0);
    mid.run(1);

    WSTAMP(1);
    // ---- FIX (base.py:518-524) / SABOTAGE (527-533): first job on the agent's own cell (544-546; job cells are distinct) ------
    // Evaluated for all jobs without a branch per job, in AGENT space: on[j] = the actors standing on job j; an actor succeeds
    // when the job's status is the one its role changes (crew: open, imposter: completed) -- one 3-input bit operation per word
    // against the job's status broadcast to every byte; a job whose actor succeeded flips.  Exact as long as no job has two
    // actors in the same step; a job with two actors in some env (rare: two agents on one cell, both working) sends the wave to
    // the turn-ordered loop instead.
    uint32_t fc80[NW], sc80[NW];
#pragma unroll
    for (int q = 0; q < NW; q++) fc80[q] = sc80[q] = 0;
#ifndef SUSNET_EXP_SKIP_JOBS
    if constexpr (JM::kOn) {
        // With the cell -> job map (fused rollouts): jobat = 0x80 | j under every agent that stands on a job.  An actor standing on
        // a job is rare (a percent or two per environment), so all that follows the flag test sits behind a ballot; the job count
        // appears nowhere.
        if (W::kBase) {
            constexpr int JW = W::JW;
            uint32_t hj80[NW], anyj = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                hj80[q] = jobat[q] & (fix80[q] | sab80[q]) & ~pend80[q]; // (flag bits only) a living job actor, not killed before its turn, on a job
                anyj |= hj80[q];
            }
            if (__builtin_amdgcn_ballot_w64(anyj != 0u) != 0ull) {
                const uint32_t jhi = JW > 1 ? JW - 1 : 0;
                uint32_t sel[NW], oh[NW], succ[NW];
                uint32_t abits = 0, nactors = 0;
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    sel[q] = jobat[q] & 0x07070707u; // the job's index: a v_perm selector into the status bytes / a one-hot table
                    const uint32_t dj80 = __builtin_amdgcn_perm(JW > 1 ? w.jd[jhi] : 0u, w.jd[0], sel[q]) << 7; // completed? (0x80 / 0)
                    succ[q] = hj80[q] & ~(w.im80[q] ^ dj80); // crew (flag 0) on an open job, imposter (0x80) on a completed one
                    oh[q] = __builtin_amdgcn_perm(0x80402010u, 0x08040201u, sel[q]); // 1 << j per agent byte
                    const uint32_t t = hj80[q] - (hj80[q] >> 7);
                    const uint32_t mine = oh[q] & (t | hj80[q]); // (0x80 flag -> 0xff mask) the actors' one-hots
                    abits = __builtin_amdgcn_sad_u8(mine, 0u, abits); // sum over the bytes: jobs with an actor -- a bit set per job ...
                    nactors += (uint32_t)__popc(hj80[q]);
                }
                // ... unless two actors share a job: the sum of n powers of two has n bits set exactly when they are distinct
                const bool crowd = (uint32_t)__popc(abits) != nactors;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(crowd) != 0ull, 0)) {
                    // two agents work on the SAME job in one env of this wave: the actors in turn order (base.py:377-382).  Jobs are
                    // independent of each other (distinct cells, base.py:295-299), so the turns are the outer, ROLLED loop: rare code
                    // (written as integer arithmetic on 0 / 1 values, not as booleans: a boolean per job and word lives in a scalar
                    // register pair, and this rare block alone spilled two dozen of them)
                    uint32_t jbits = 0; // job status as a bit mask
#pragma unroll
                    for (int j = 0; j < 4 * JW; j++) jbits |= ((w.jd[j / 4] >> (8 * (j & 3))) & 1u) << j;
#pragma clang loop unroll(disable)
                    for (uint32_t turn = 0; turn < (uint32_t)A; turn++) {
                        const uint32_t tb = (turn | 0x80u) * k01;
#pragma unroll
                        for (int q = 0; q < NW; q++) {
                            const uint32_t me = zero80(R[q] ^ tb) & hj80[q]; // the agent whose turn it is, if it works on a job (one byte at most)
                            const uint32_t myjob = __builtin_amdgcn_sad_u8(sel[q] & ff_from80(me), 0u, 0u);
                            const uint32_t actor = __builtin_amdgcn_sad_u8(me >> 7, 0u, 0u), imp = __builtin_amdgcn_sad_u8((me & w.im80[q]) >> 7, 0u, 0u);
                            const uint32_t status = (jbits >> myjob) & 1u;
                            const uint32_t ok = actor & ~(imp ^ status); // crew (0) on an open job (0), imposter (1) on a completed one (1)
                            jbits ^= ok << myjob;
                            const uint32_t f = ok & ~imp, sb = ok & imp;
                            e.m_fix += f;
                            e.m_sab += sb;
                            fc80[q] |= me & (0u - f);
                            sc80[q] |= me & (0u - sb);
                        }
                    }
                    w.jd[0] = ((jbits & 15u) * 0x00204081u) & k01; // bits -> 0x01 per job byte (see below)
                    if (JW > 1) w.jd[jhi] = ((jbits >> 4) * 0x00204081u) & k01;
#pragma unroll
                    for (int q = 0; q < NW; q++) idx4[q] += (fc80[q] >> 4) + (sc80[q] >> 5) + (sc80[q] >> 4); // RC_FIX 2, RC_SAB 3, times 4
                } else {
                    uint32_t tbits = 0;
#pragma unroll
                    for (int q = 0; q < NW; q++) {
                        const uint32_t t = succ[q] - (succ[q] >> 7);
                        tbits = __builtin_amdgcn_sad_u8(oh[q] & (t | succ[q]), 0u, tbits); // jobs that flip
                        fc80[q] = succ[q] & ~w.im80[q];
                        sc80[q] = succ[q] & w.im80[q];
                        e.m_fix += (uint32_t)__popc(fc80[q]);
                        e.m_sab += (uint32_t)__popc(sc80[q]);
                        idx4[q] += (succ[q] >> 4) + (sc80[q] >> 5); // RC_FIX 2 / RC_SAB 3, times 4
                    }
                    // bits -> 0x01 per job byte: b * (1 + 2^7 + 2^14 + 2^21) puts bit k of a 4-bit b at bit 8k (the four shifted copies
                    // occupy disjoint bit ranges: no carries)
                    w.jd[0] ^= ((tbits & 15u) * 0x00204081u) & k01;
                    if (JW > 1) w.jd[jhi] ^= ((tbits >> 4) * 0x00204081u) & k01;
                }
            }
        }
    } else if (W::kBase && J > 0) {
        constexpr int JW = W::JW;
        uint32_t ja80[NW], on[J][NW], acted[NW], tog[JW];
#pragma unroll
        for (int q = 0; q < NW; q++) {
            ja80[q] = (fix80[q] | sab80[q]) & ~pend80[q]; // job actors did not move: w.xy is still their cell
            acted[q] = 0;
        }
#pragma unroll
        for (int q = 0; q < JW; q++) tog[q] = 0;
        uint32_t crowd = 0; // bit 1 and up: some job has more than one actor on it
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t dj80 = (0u - ((w.jd[j / 4] >> (8 * (j & 3))) & 1u)) & k80; // the job's status at every agent byte
            uint32_t n = 0, n_on = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                on[j][q] = (W::J >= 0 || j < Jn) ? zero80(w.xy[q] ^ w.jb[j]) & ja80[q] : 0u; // (a slot past a run-time job count matches nobody)
                n_on += (uint32_t)__popc(on[j][q]);
                const uint32_t succ = on[j][q] & ~(w.im80[q] ^ dj80); // crew (flag 0) on an open job, imposter (0x80) on a completed one
                acted[q] |= succ;
                n += (uint32_t)__popc(succ);
            }
            crowd |= n_on;
            tog[j / 4] |= n << (8 * (j & 3));
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(crowd > 1u) != 0ull, 0)) {
            // two agents work on the SAME job in one env of this wave: the actors in turn order (base.py:377-382).  Jobs are
            // independent of each other (distinct cells, base.py:295-299), so the turns are the outer, ROLLED loop: rare code, kept small
            uint32_t dj[J];
#pragma unroll
            for (int j = 0; j < J; j++) dj[j] = (w.jd[j / 4] >> (8 * (j & 3))) & 1u;
#pragma clang loop unroll(disable)
            for (uint32_t turn = 0; turn < (uint32_t)A; turn++) {
                const uint32_t tb = (turn | 0x80u) * k01;
                uint32_t mine[NW];
#pragma unroll
                for (int q = 0; q < NW; q++) mine[q] = zero80(R[q] ^ tb); // the agent whose turn it is
#pragma unroll
                for (int j = 0; j < J; j++) {
#pragma unroll
                    for (int q = 0; q < NW; q++) {
                        const uint32_t me = mine[q] & on[j][q]; // ... if it works on this job
                        const bool is_sab = (me & w.im80[q]) != 0u, is_fix = (me & ~w.im80[q]) != 0u;
                        const bool f = is_fix && dj[j] == 0u, sb = is_sab && dj[j] != 0u;
                        dj[j] = f ? 1u : (sb ? 0u : dj[j]);
                        e.m_fix += f ? 1u : 0u;
                        e.m_sab += sb ? 1u : 0u;
                        fc80[q] |= f ? me : 0u;
                        sc80[q] |= sb ? me : 0u;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < JW; q++) w.jd[q] = 0;
#pragma unroll
            for (int j = 0; j < J; j++) w.jd[j / 4] |= dj[j] << (8 * (j & 3));
        } else {
#pragma unroll
            for (int q = 0; q < JW; q++) w.jd[q] ^= tog[q];
#pragma unroll
            for (int q = 0; q < NW; q++) {
                fc80[q] = acted[q] & ~w.im80[q];
                sc80[q] = acted[q] & w.im80[q];
                e.m_fix += (uint32_t)__popc(fc80[q]);
                e.m_sab += (uint32_t)__popc(sc80[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < NW; q++) idx4[q] += (fc80[q] >> 4) + (sc80[q] >> 5) + (sc80[q] >> 4); // RC_FIX 2, RC_SAB 3, times 4
    }
#endif

    WSTAMP(2);
    // ---- tag actions (tagging.py:103-110) and the vote (tagging.py:180-207) ------------------------------------------------------
    float team = 0.0f; // team reward: vote outcome, then the win reward (tagging.py:196, 209-213)
    bool voted_out = false; // the vote ejected somebody this step
#ifndef SUSNET_EXP_SKIP_TAG
    if (W::kTag) {
        const uint32_t hi = NW > 1 ? NW - 1 : 0;
        uint32_t tgt[NW], vt80[NW];
        uint32_t any_new = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            // k-th OTHER agent ascending: target = action - len(role list), + 1 from the actor's own index on (tagging.py:68-75)
            const uint32_t nrb = 0x06060606u + (w.im80[q] >> 7);
            uint32_t t = ((act[q] | k80) - nrb) & k7f; // (meaningful where tag80 is set)
            const uint32_t idxb = q == 0 ? 0x03020100u : 0x07060504u;
            t += (((t | k80) - idxb) & k80) >> 7;
            tgt[q] = t & 0x07070707u;
        }
#pragma unroll
        for (int q = 0; q < NW; q++) {
            // the target's alive flag at the tagger's turn: alive now (after this step's kills) ...
            uint32_t tal80 = __builtin_amdgcn_perm(NW > 1 ? w.al[hi] : 0u, w.al[0], tgt[q]) << 7;
            // ... or killed this step by a killer whose turn comes AFTER the tagger's (the tagger had already acted)
#pragma unroll
            for (int it = 0; it < NI; it++)
                tal80 |= __builtin_amdgcn_perm(NW > 1 ? vk80[it][hi] : 0u, vk80[it][0], tgt[q]) & ~gek80[it][q];
            vt80[q] = tag80[q] & ~(w.used[q] << 7) & tal80 & k80; // first tag of the interval, living target
            w.used[q] |= vt80[q] >> 7;
            any_new |= vt80[q];
        }
        if (__builtin_amdgcn_ballot_w64(any_new != 0u) != 0ull) { // tag_counts[target] += 1 per counted tag
            // the step's increments first, FOUR bits per target in one register (a target collects at most A - 1 <= 7 tags): a shift-and-add
            // per tagger (round 4: a 64-bit shift and a 64-bit add per tagger on the byte counts themselves: ~35 instructions a tick) ...
            uint32_t inc4 = 0;
            uint32_t sh4[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) sh4[q] = tgt[q] << 2; // 4 * target per agent byte (< 32)
#pragma unroll
            for (int i = 0; i < A; i++) {
                const uint32_t bit = __builtin_amdgcn_ubfe(vt80[i / 4], 8u * (uint32_t)(i & 3) + 7u, 1u);
                const uint32_t sh = __builtin_amdgcn_ubfe(sh4[i / 4], 8u * (uint32_t)(i & 3), 5u);
                inc4 += bit << sh;
            }
            // ... then spread to one byte per target and added to the counts (no byte overflows: counts stay below 8)
            uint32_t lo = inc4 & 0xffffu;
            lo = (lo | (lo << 8)) & 0x00ff00ffu;
            lo = (lo | (lo << 4)) & 0x0f0f0f0fu;
            w.cnt[0] += lo;
            if (NW > 1) {
                uint32_t hi4 = inc4 >> 16;
                hi4 = (hi4 | (hi4 << 8)) & 0x00ff00ffu;
                hi4 = (hi4 | (hi4 << 4)) & 0x0f0f0f0fu;
                w.cnt[hi] += hi4;
            }
        }
        // tagging.py:180: tag_counts *= alive_agents.  A count only ever grows on a LIVING target (tagging.py:105), so the product
        // changes something exactly when this step killed somebody (a vote's ejection is followed by the reset of all counts) --
        // or when the state came from outside (check_win: the first tick of a launch)
        {
            uint32_t killed = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) killed |= kc80[q];
            if (__builtin_amdgcn_ballot_w64(check_win || killed != 0u) != 0ull) {
#pragma unroll
                for (int q = 0; q < NW; q++) w.cnt[q] &= ff_from80(w.al80[q]);
            }
        }
        w.timer += 1u;                                                               // tagging.py:182
        const bool due = w.timer >= (uint32_t)c.tag_interval;
        if (__builtin_amdgcn_ballot_w64(due) != 0ull) { // tagging.py:184-207
            // np.argmax (first maximum) as ONE maximum over keys count << 3 | (7 - index): a larger count wins, among equal counts the
            // lower index (counts < 8: a key fits its byte)
            uint32_t kmax = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const uint32_t keys = (w.cnt[q] << 3) + (q == 0 ? 0x04050607u : 0x00010203u);
#pragma unroll
                for (int i = 4 * q; i < A && i < 4 * q + 4; i++) {
                    const uint32_t kv = __builtin_amdgcn_ubfe(keys, 8u * (uint32_t)(i & 3), 8u);
                    kmax = kv > kmax ? kv : kmax;
                }
            }
            const uint32_t best = 7u - (kmax & 7u), highest = kmax >> 3;
            uint32_t alive_sum = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) alive_sum += (uint32_t)__popc(w.al[q] & kLive[q] & k01);
            const bool out = due && highest >= ((alive_sum + 1u) >> 1);
            voted_out = out;
            const uint64_t hot = (uint64_t)(out ? 1u : 0u) << (8u * best);
            const uint64_t im = (uint64_t)w.im80[0] | ((uint64_t)(NW > 1 ? w.im80[hi] : 0u) << 32);
            const bool vimp = ((im >> (8u * best + 7u)) & 1ull) != 0ull;
            w.al[0] &= ~(uint32_t)hot;
            if (NW > 1) w.al[hi] &= ~(uint32_t)(hot >> 32);
#pragma unroll
            for (int q = 0; q < NW; q++) { // (the derived forms of the alive flags: see Swar)
                const uint32_t h1 = q == 0 ? (uint32_t)hot : (uint32_t)(hot >> 32);
                w.al80[q] &= ~(h1 << 7);
                w.crew80[q] &= ~(h1 << 7);
                w.ridx[q] += h1 << 4;
                idx4[q] += h1 << 4;
            }
            const float vote = c.fr[RW_VOTE];
            team += out ? vote * (vimp ? -1.0f : 1.0f) : 0.0f; // tagging.py:196, sign as coded
            e.m_kv += out ? (vimp ? (1u << 16) : (1u << 24)) : 0u;
#pragma unroll
            for (int q = 0; q < NW; q++) { // tagging.py:237-241
                w.cnt[q] = due ? 0u : w.cnt[q];
                w.used[q] = due ? 0u : w.used[q];
            }
            w.timer = due ? 0u : w.timer;
        }
    }
#endif

    WSTAMP(3);
    // ---- check_win_condition: base.py:409-460 / pred_prey.py:78-99 --------------------------------------------------------------
    uint32_t wsel = 0u; // reward-table row of THIS step's outcome: 0 none, 16 crew won, 32 imposters won
    done = false;
    bool win_inputs_changed = check_win || (W::kBase && Jn == 0); // (no jobs: FourRoomEnv's "all jobs done" holds at every step, base.py:430)
    if (!win_inputs_changed) { // alive flags / job status moved this step?
        uint32_t ev = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) ev |= kc80[q] | fc80[q] | sc80[q];
        win_inputs_changed = ev != 0u || voted_out;
    }
    if (__builtin_amdgcn_ballot_w64(win_inputs_changed) != 0ull) {
        int alive_imp = 0, alive_all = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            alive_all += __popc(w.al80[q]);
            alive_imp += __popc(w.al80[q] & w.im80[q]);
        }
        int done_jobs = 0;
#pragma unroll
        for (int q = 0; q < W::JW; q++) done_jobs += __popc(w.jd[q]);
        bool crew_won, imp_won; // (selects, no branches: every lane evaluates both rules)
        if (!W::kBase) {
            crew_won = Jn != 0 && done_jobs == Jn;
            imp_won = !crew_won && alive_all - alive_imp == 0;
        } else {
            crew_won = alive_imp == 0 || done_jobs == Jn;
            imp_won = !crew_won && alive_all - alive_imp <= alive_imp;
        }
        done = crew_won || imp_won;
        e.flags |= (crew_won ? FLAG_CREW_WON : 0u) | (imp_won ? FLAG_IMP_WON : 0u);
        wsel = crew_won ? 16u : (imp_won ? 32u : 0u);
        if (!W::kTag) {
#pragma unroll
            for (int q = 0; q < NW; q++) idx4[q] += (wsel << 2) * k01; // the table's "crew won" / "imposters won" block
        }
    }
    // ---- rewards: assignments -> _merge_rewards (base.py:553-563) -> zero fill (389-390), one lookup per agent in the
    // host-evaluated table [win][index < n_imposters][dead][assignment code]; byte = 4 * table index.  idx4 started from the episode's
    // base (dead / index < n_imposters: Swar::ridx) and collected this step's assignment codes and outcome where they arose
#ifdef SUSNET_EXP_SKIP_REWARDS
#pragma unroll
    for (int i = 0; i < A; i++) rr[i] = __uint_as_float(idx4[i / 4] ^ kc80[i / 4]);
    if (false) {
#else
    if (!W::kTag) {
#endif
#pragma unroll
#ifdef SUSNET_EXP_NO_REWARD_LDS
        for (int i = 0; i < A; i++) rr[i] = (float)((idx4[i / 4] >> (8 * (i & 3))) & 0xffu);
#else
        for (int i = 0; i < A; i++) rr[i] = lds_reward_lookup((idx4[i / 4] >> (8 * (i & 3))) & 0xffu);
#endif
    } else {
        // tagging.py:162-213: every agent starts from time_step_reward (no zero fill afterwards), assignments overwrite, the team
        // reward (vote, then win) is added, indices [:n_imposters] are negated, the dead get dead_penalty.  float32 is exact here:
        // the compiled-in kernels are only selected when every constant is a small integer
        const float end = c.fr[RW_END];
        team += wsel == 16u ? end : (wsel == 32u ? -1.0f * end : 0.0f);
        // the assignment a step left on an agent (none = time_step_reward, kill, fix, -sabotage): a 4-entry table the host put at
        // the head of the LDS reward table for this variant
        const uint32_t dead_bits = __float_as_uint(c.fr[RW_DEAD]);
        uint32_t code4[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) code4[q] = (kc80[q] >> 5) | (fc80[q] >> 4) | (sc80[q] >> 5) | (sc80[q] >> 4); // RC_KILL 1, RC_FIX 2, RC_SAB 3, times 4
#pragma unroll
        for (int i = 0; i < A; i++) {
            const int q = i / 4, sh = 8 * (i & 3);
            float r = lds_reward_lookup((code4[q] >> sh) & 0xffu) + team;
            if (i < NI) r = -r; // indices [:n_imposters], NOT the imposter mask (base.py:559); x * -1 == -x, zeros included
            const uint32_t live = 0u - ((w.al[q] >> sh) & 1u);
            rr[i] = __uint_as_float((__float_as_uint(r) & live) | (dead_bits & ~live)); // base.py:562
        }
    }
    WSTAMP(4);
    // base.py:392-395: t saturates at max_time_steps - 1
    trunc = false;
    if (e.t == (uint32_t)(c.max_t - 1)) trunc = true;
    else e.t += 1u;
}

// flatten_state (base.py:234-235) of the byte-parallel state as packed dwords: positions, alive, job cells, job status
// (JJ: the job count the row is built for -- the compiled-in one, or, for the family, the case of a wave-uniform switch)
template <class S, int JJ>
struct RawRowOf {
    static constexpr int A = S::kA, F = 3 * A + ((JJ > 0 || S::kVar == SUSNET_VARIANT_TAGGING) ? 3 * JJ : 0) + (S::kVar == SUSNET_VARIANT_TAGGING ? 2 * A + 1 : 0);
    static constexpr int kDwords = (F + 3) / 4;
};
template <class S, int JJ>
__device__ __forceinline__ void raw_row_swar_n(const Swar<S> &w, uint32_t (&row)[RawRowOf<S, JJ>::kDwords], uint32_t tag_interval = 0u) {
    using W = Swar<S>;
    constexpr int A = W::A, J = JJ, F = RawRowOf<S, JJ>::F;
    static_assert(J <= W::JMAX, "job slots");
    uint8_t b[(F + 3) / 4 * 4];
    // assembled bytewise from a handful of words; the compiler folds the static byte moves into v_perm / shifts
    uint32_t pos[2 * W::NW];
#pragma unroll
    for (int q = 0; q < W::NW; q++) {
        const uint32_t x = w.xy[q] & 0x0f0f0f0fu, y = (w.xy[q] >> 4) & 0x0f0f0f0fu;
        pos[2 * q] = __builtin_amdgcn_perm(y, x, 0x05010400u);     // x0 y0 x1 y1
        pos[2 * q + 1] = __builtin_amdgcn_perm(y, x, 0x07030602u); // x2 y2 x3 y3
    }
    int k = 0;
#pragma unroll
    for (int i = 0; i < 2 * A; i++) b[k++] = (uint8_t)(pos[i / 4] >> (8 * (i & 3)));
#pragma unroll
    for (int i = 0; i < A; i++) b[k++] = (uint8_t)((w.al[i / 4] >> (8 * (i & 3))) & 1u);
    if (J > 0) {
#pragma unroll
        for (int i = 0; i < 2 * J; i++) b[k++] = (uint8_t)(w.jobs_obs[i / 4] >> (8 * (i & 3)));
#pragma unroll
        for (int j = 0; j < J; j++) b[k++] = (uint8_t)((w.jd[j / 4] >> (8 * (j & 3))) & 1u);
    }
    if (W::kTag) { // tagging.py:220-230: used_tag_actions, tag_counts, steps until the vote
#pragma unroll
        for (int i = 0; i < A; i++) b[k++] = (uint8_t)((w.used[i / 4] >> (8 * (i & 3))) & 1u);
#pragma unroll
        for (int i = 0; i < A; i++) b[k++] = (uint8_t)(w.cnt[i / 4] >> (8 * (i & 3)));
        b[k++] = (uint8_t)(tag_interval - w.timer);
    }
#pragma unroll
    for (; k < (F + 3) / 4 * 4; k++) b[k] = 0;
#pragma unroll
    for (int d = 0; d < (F + 3) / 4; d++)
        row[d] = (uint32_t)b[4 * d] | ((uint32_t)b[4 * d + 1] << 8) | ((uint32_t)b[4 * d + 2] << 16) | ((uint32_t)b[4 * d + 3] << 24);
}
template <class S>
__device__ __forceinline__ void raw_row_swar(const Swar<S> &w, uint32_t (&row)[(S::kRawF + 3) / 4], uint32_t tag_interval = 0u) {
    static_assert(S::kJ >= 0 && RawRowOf<S, S::kJ>::F == S::kRawF, "compiled-in job count");
    raw_row_swar_n<S, S::kJ>(w, row, tag_interval);
}

} // namespace susnet
