// susnet_swar2.h -- the byte-parallel step (susnet_swar.h) with TWO lanes per environment, for 8-agent games at batches that fill
// only half of every wave (32 768 envs per GPU = the 8-GPU shard of BASELINE configs[3]: 1 024 waves of 32 envs, one per SIMD).
//
// Lane L (0..31) holds agents 0..3 of env L, lane L + 32 agents 4..7: every per-agent quantity is ONE packed word per lane
// instead of two, so the per-word part of the step costs half the instructions per wave.  What concerns the whole environment
// (step counters, job status, the random streams, win / truncation) is kept identically in both lanes; the few places where the
// two words meet -- the killer's cell and rank, kill candidates, job toggles, alive counts -- go through v_permlane32_swap_b32,
// which hands both lanes the pair (word of lanes 0..31, word of lanes 32..63) in one instruction.
//
// Reference behaviour: as susnet_swar.h (base.py:332-563, pred_prey.py:78-99).
#pragma once

#include "susnet_swar.h"

namespace susnet {

template <class S>
struct UseSplit {
    static constexpr bool value = UseSwar<S>::value && S::kA == 8 && S::kVar != SUSNET_VARIANT_TAGGING && S::kJ >= 0 && S::kJ <= 4;
};

struct Pair {
    uint32_t lo, hi; // the value lanes 0..31 hold / the value lanes 32..63 hold (of the same environment), in both lanes
};
__device__ __forceinline__ Pair both_halves(uint32_t v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); // vdst[32..63] <-> src[0..31]
    return Pair{r[0], r[1]};
}

template <class S>
struct Swar2 {
    static constexpr int A = S::kA, J = S::kJ, NI = S::kNI > 0 ? S::kNI : 1;
    static constexpr bool kBase = S::kVar != SUSNET_VARIANT_ITG;
    uint32_t h;                 // 0: this lane holds agents 0..3, 1: agents 4..7
    uint32_t xy, al, im80;      // my word of cells / alive (0x01) / imposter flags (0x80)
    uint32_t al80, crew80, ridx; // derived forms of the alive flags, kept by the code that changes them (see Swar)
    uint32_t isel[NI];          // v_perm selector of imposter s's byte over the PAIR {hi word, lo word}
    uint32_t iselb[NI];         // ... into ALL four bytes
    uint32_t ihot[NI];          // 0x80 at imposter s's byte if it is in my word
    uint32_t jb[J > 0 ? J : 1]; // job cell in all four bytes
    uint32_t jobs_obs[2], jd;   // observation bytes of the job cells; completed: 0x01 per job
    uint32_t nq[4];             // len(agent_action_map[i]) of MY four agents
    uint32_t pre;               // what my first action draw's word is multiplied with first (see sample_actions_pair)
    uint32_t imp_bits;          // imposter bitmask of the episode
};

template <class S, class Store>
__device__ __forceinline__ void to_swar2(const Consts &c, const Store &st, const Env &e, uint32_t h, Swar2<S> &w) {
    Swar<S> f;
    to_swar<S>(c, st, e, f);
    w.h = h;
    w.xy = h ? f.xy[1] : f.xy[0];
    w.al = h ? f.al[1] : f.al[0];
    w.im80 = h ? f.im80[1] : f.im80[0];
    w.al80 = h ? f.al80[1] : f.al80[0];
    w.crew80 = h ? f.crew80[1] : f.crew80[0];
    w.ridx = h ? f.ridx[1] : f.ridx[0];
#pragma unroll
    for (int s = 0; s < Swar2<S>::NI; s++) {
        w.isel[s] = f.isel[s];
        w.iselb[s] = f.iselb[s];
        w.ihot[s] = h ? f.ihot[s][1] : f.ihot[s][0];
    }
#pragma unroll
    for (int j = 0; j < Swar2<S>::J; j++) w.jb[j] = f.jb[j];
    w.jobs_obs[0] = f.jobs_obs[0];
    w.jobs_obs[1] = f.jobs_obs[1];
    w.jd = f.jd[0];
#pragma unroll
    for (int q = 0; q < 4; q++) w.nq[q] = h ? f.nact[4 + q] : f.nact[q];
    w.pre = h ? f.nact[0] * f.nact[1] * f.nact[2] * f.nact[3] : 1u;
    w.imp_bits = swar_imp_bits(f);
}

// base.py:326-330 on the production stream, MY four action bytes only.  The tick's action draws are nested multiply-shift digits of
// its words (susnet_device.h AwLayout; 8 agents with at most 7 actions: agents 0-4 in word 0, agents 5-7 and the first shuffle digits
// in word 1).  The low lane takes the first four digits of word 0.  The high lane needs the FIFTH digit of word 0: what four
// draws leave of a word is lo32(word * n0 n1 n2 n3) -- multiplication mod 2^32 is associative -- so one multiply by the episode's
// product gets it there; it then takes the three digits of word 1.  Same instructions in both lanes, lane-specific operands:
// one multiply, four multiply-shifts, one select instead of eight multiply-shifts and the packing of both words.
// `as.rem` is left as ranks_from_lut expects it: what the action draws left of word 1 (the high lane holds it).
template <class S, int POS, class AS>
__device__ __forceinline__ uint32_t sample_actions_pair(const Swar2<S> &w, PhiloxRng &rng, AS &as, uint64_t tick) {
    static_assert(S::kA == 8 && S::kAw.word[0] == 0 && S::kAw.word[4] == 0 && S::kAw.word[5] == 1 && S::kAw.word[7] == 1 && S::kAw.word[8] == 1,
                  "agents 0-4 draw from word 0, agents 5-7 from word 1, the shuffle digits follow in word 1");
    const uint64_t Wt = (uint64_t)S::kAw.W;
    auto fetch = [&](int k) __attribute__((always_inline)) {
        return POS >= 0 ? as.word_in_group(rng, (tick - (uint64_t)(POS >= 0 ? POS : 0)) * Wt, (POS >= 0 ? POS : 0) * S::kAw.W + k)
                        : as.word(rng, tick * Wt + (uint64_t)k);
    };
    const uint32_t w0 = fetch(0), w1 = fetch(1);
    uint32_t x = w0 * w.pre;
    uint64_t p = (uint64_t)x * (uint64_t)w.nq[0];
    uint32_t act = (uint32_t)(p >> 32);
    x = w.h ? w1 : (uint32_t)p;
#pragma unroll
    for (int q = 1; q < 4; q++) {
        p = (uint64_t)x * (uint64_t)w.nq[q];
        act |= (uint32_t)(p >> 32) << (8 * q);
        x = (uint32_t)p;
    }
    as.rem = both_halves(x).hi;
    return act;
}

// the whole environment again (both lanes of the pair must be active)
template <class S>
__device__ __forceinline__ void gather_swar2(const Swar2<S> &w, Swar<S> &f) {
    const Pair x = both_halves(w.xy), a = both_halves(w.al), m = both_halves(w.im80);
    f.xy[0] = x.lo; f.xy[1] = x.hi;
    f.al[0] = a.lo; f.al[1] = a.hi;
    f.im80[0] = m.lo; f.im80[1] = m.hi;
    f.jobs_obs[0] = w.jobs_obs[0];
    f.jobs_obs[1] = w.jobs_obs[1];
    f.jd[0] = w.jd;
}

// One step; act / R: MY word of the action bytes and of the turn ranks.  rr: the rewards of my four agents.
// jm: the environment's column of the cell -> job map (susnet_swar.h JobMap; both lanes of a pair read the same column);
// check_win: see step_swar.
// realign: see step_swar (the caller aligned the event cursor once; a landed kill leaves it block-aligned for the next step).
template <class S, class RNG, class MID = NoMid>
__device__ __forceinline__ void step_swar2(const Consts &c, Swar2<S> &w, Env &e, RNG &rng, uint32_t act, uint32_t R, float (&rr)[4], bool &done,
                                           bool &trunc, const JobMap &jm, bool check_win, bool realign, MID &&mid = MID()) {
    using W = Swar2<S>;
    constexpr int A = W::A, J = W::J, NI = W::NI;
    const uint32_t h = w.h;
    e.m_steps += 1; // base.py:366

    // ---- action classes (0x80 per agent): alive agents only (base.py:477) --------------------------------------------------
    const uint32_t g5 = (act + 0x7b7b7b7bu) & k80, g6 = (act + 0x7a7a7a7au) & k80; // action index >= 5 / >= 6
    const uint32_t al80 = w.al80;
    uint32_t kill80, fix80 = 0, sab80 = 0, rows;
    if (W::kBase) { // crew: 5 = FIX; imposter: 5 = SABOTAGE, 6 = KILL (base.py:82-99)
        kill80 = g6 & al80;
        const uint32_t j5 = g5 & ~g6 & al80;
        sab80 = j5 & w.im80;
        fix80 = j5 & ~w.im80;
        rows = act; // (rows 5 and 6 of the table are both the identity: susnet_device.h kMoveRows)
    } else { // pred_prey.py:4-19: imposter 5 = KILL, no job actions
        kill80 = g5 & al80;
        rows = act;
    }
    const uint32_t mv80 = ~g5 & al80;
    // ---- destinations: one lookup per agent in the (action, cell) table (move() + _is_valid_position(), base.py:69-79, 548-551)
    uint32_t dest;
    {
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t sel = 0x0c0c0000u | ((4u + (uint32_t)i) << 8) | (uint32_t)i; // address = row << 8 | cell
            d[i] = lds_move_lookup(__builtin_amdgcn_perm(rows, w.xy, sel));
        }
        dest = d[0] | (d[1] << 8) | (d[2] << 16) | (d[3] << 24);
    }
    const uint32_t newt = sel_bytes(ff_from80(mv80), dest, w.xy); // positions if every living mover moved
    // the job under each of my agents (0x80 | j, or 0): job actors do not move; issued here, used after the kill section
    uint32_t jobat = 0;
    if (W::kBase) {
        uint32_t mj[4];
#pragma unroll
        for (int i = 0; i < 4; i++) mj[i] = jm.at((w.xy >> (8 * i)) & 0xffu);
        jobat = mj[0] | (mj[1] << 8) | (mj[2] << 16) | (mj[3] << 24);
    }

    // ---- KILL (base.py:490-515), imposters in turn order --------------------------------------------------------------------
    uint32_t pend80 = 0;
    uint32_t idx4 = w.ridx; // reward-table byte index per agent: the episode's base + what this step adds (see step_swar)
    bool changed = check_win; // something the win rules read moved this step (the same in both lanes of a pair)
#ifndef SUSNET_EXP_SKIP_KILL // diagnostic builds only (tools/build_variant_tu.sh, profiles/r05_step_sections_cfg4.md): a section's share of the tick -- WRONG results
    {
        const uint64_t cur0 = rng.cur; // the step's (aligned) event cursor: a landed kill takes word cur0 + kills landed before it
        const Pair pk = both_halves(kill80), pr = both_halves(R), px = both_halves(w.xy);
        uint32_t kb[NI], rb[NI], cb[NI]; // per imposter slot, in ALL four bytes: kill flag (0x80 / 0), rank | 0x80, cell
#pragma unroll
        for (int s = 0; s < NI; s++) {
            kb[s] = __builtin_amdgcn_perm(pk.hi, pk.lo, w.iselb[s]);
            rb[s] = __builtin_amdgcn_perm(pr.hi, pr.lo, w.iselb[s]);
            cb[s] = __builtin_amdgcn_perm(px.hi, px.lo, w.iselb[s]);
        }
        bool second_first = false; // two imposters: the one with the earlier turn kills first
        if (NI == 2) second_first = kb[1] != 0u && (kb[0] == 0u || rb[1] < rb[0]);
        uint32_t landed = 0; // kills this environment landed so far in this step
        // the second turn's gate ("somebody attempts": the second turn only has work where BOTH imposters do) is known HERE: its compare is
        // pinned above the whole first turn (the ranks pass through it: susnet_swar.h NoMid), its branch finds the mask long settled
        uint32_t Rt = R;
        uint64_t second_any = 0ull;
        if (NI == 2) second_any = early_ballot_nz(second_first ? kb[0] : kb[1], Rt);
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int s0 = it, s1 = NI - 1 - it;
            const uint32_t kbi = second_first ? kb[s1] : kb[s0], rbi = second_first ? rb[s1] : rb[s0], cbi = second_first ? cb[s1] : cb[s0];
            // (no ballot on "somebody attempts" for the first kill turn: some environment of the wave nearly always does)
            if (it > 0 && (NI == 2 ? second_any == 0ull : __builtin_amdgcn_ballot_w64(kbi != 0u) == 0ull)) continue;
            const uint32_t ge80 = (Rt - (rbi & k7f)) & k80; // rank >= the killer's: has not acted yet (the killer itself included)
            const uint32_t pos = sel_bytes(ff_from80(ge80), w.xy, newt);
            const uint32_t cand = zero80(pos ^ cbi) & w.crew80 & kbi; // living crew NOW on the killer's cell (base.py:535-542), if it attempts
            // (a crew member on the killer's cell is rare: the rest -- the exchange between the two lanes included -- sits behind a ballot)
            bool none;
            if constexpr (MidOf<MID>::type::kOn) { // (the next tick's action digits between the first gate's compare and its branch: susnet_swar.h NoMid)
                if (it == 0) {
                    const uint64_t any = early_ballot_nz(cand, mid.tie(0));
                    mid.run(0);
                    none = any == 0ull;
                } else none = __builtin_amdgcn_ballot_w64(cand != 0u) == 0ull;
            } else none = __builtin_amdgcn_ballot_w64(cand != 0u) == 0ull;
            if (none) continue;
            const Pair pc = both_halves(cand);
            const uint32_t nc = (uint32_t)__popc(pc.lo) + (uint32_t)__popc(pc.hi);
            // base.py:497: uniform among the candidates (ascending agent index); with one candidate the lowest set flag
            uint32_t v0 = pc.lo & (0u - pc.lo), v1 = pc.lo != 0u ? 0u : (pc.hi & (0u - pc.hi));
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(nc > 1u) != 0ull, 0)) {
                if (nc > 1u) {
                    if (!RNG::kNumpy) rng.cur = cur0 + (uint64_t)landed; // production protocol: one word per landed kill, its value only matters here
                    const uint32_t r = rng.bounded(nc);
                    uint32_t c0 = pc.lo, c1 = pc.hi;
                    for (uint32_t k = 0; k < r; k++) {
                        const bool lo = c0 != 0u;
                        c0 = lo ? (c0 & (c0 - 1u)) : c0;
                        c1 = lo ? c1 : (c1 & (c1 - 1u));
                    }
                    v0 = c0 & (0u - c0);
                    v1 = c0 != 0u ? 0u : (c1 & (0u - c1));
                }
            }
            const uint32_t v80 = h ? v1 : v0;
            const bool hit = nc != 0u;
            if (!RNG::kNumpy) rng.cur = hit ? (realign ? cur0 + 4ull : cur0 + (uint64_t)landed + 1ull) : rng.cur;
            landed += hit ? 1u : 0u;
            changed |= hit;
            e.m_kv += hit ? 1u : 0u; // IMP_KILLED_CREW, base.py:508
            w.al &= ~(v80 >> 7);     // base.py:511
            w.al80 &= ~v80;
            w.crew80 &= ~v80;
            w.ridx += v80 >> 3;      // the victim's rewards come from the "dead" rows from now on (base.py:562)
            const uint32_t hot = hit ? (second_first ? w.ihot[s1] : w.ihot[s0]) : 0u;
            idx4 += (v80 >> 3) + (hot >> 5); // RC_KILL * 4 for the killer; base.py:514-515: the victim's slot ends as dead_penalty
            pend80 |= v80 & ge80;    // killed before its own turn: it never acts
        }
    }
#endif
    w.xy = sel_bytes(ff_from80(pend80), w.xy, newt); // a victim that had not acted yet stays where it was
#ifdef SUSNET_EXP_SKIP_KILL
    mid.run(0);
#endif
    constexpr bool kMidInJobGate = MidOf<MID>::type::kOn && W::kBase; // (part 1 sits in the job section's gate where there is one)
#ifdef SUSNET_EXP_SKIP_JOBS
    mid.run(1);
#else
    if constexpr (!kMidInJobGate) mid.run(1);
#endif

    // ---- FIX (base.py:518-524) / SABOTAGE (527-533) through the cell -> job map (see susnet_swar.h) ------------------------------
    uint32_t fc80 = 0, sc80 = 0;
#ifndef SUSNET_EXP_SKIP_JOBS
    if (W::kBase) {
        static_assert(J <= 4, "one word of job status bytes");
        const uint32_t hj80 = jobat & (fix80 | sab80) & ~pend80; // (flag bits only) a living job actor, not killed before its turn, on a job
        uint64_t actors;
        if constexpr (kMidInJobGate) { // (the next tick's turn ranks between this gate's compare and its branch)
            actors = early_ballot_nz(hj80, mid.tie(1));
            mid.run(1);
        } else actors = __builtin_amdgcn_ballot_w64(hj80 != 0u);
#ifdef SUSNET_EXP_FREE_JOBS // experiment: the body without its gate (same results)
        {
#else
        if (actors != 0ull) {
#endif
            const uint32_t sel = jobat & 0x03030303u;                                  // the job's index
            const uint32_t dj80 = __builtin_amdgcn_perm(0u, w.jd, sel) << 7;          // completed? (0x80 / 0)
            const uint32_t succ = hj80 & ~(w.im80 ^ dj80);                             // crew on an open job, imposter on a completed one
            const uint32_t oh = __builtin_amdgcn_perm(0u, 0x08040201u, sel);          // 1 << j per agent byte
            // per lane: sum of the actors' one-hots | sum of the succeeding actors' << 8 | number of actors << 16; the pair's sums
            // follow from ONE exchange (every field stays below its 8 bits: at most 8 actors on jobs 0 .. 3)
            const uint32_t ta = hj80 - (hj80 >> 7), ts = succ - (succ >> 7);
            const uint32_t mine = __builtin_amdgcn_sad_u8(oh & (ta | hj80), 0u, 0u) | (__builtin_amdgcn_sad_u8(oh & (ts | succ), 0u, 0u) << 8) |
                                  ((uint32_t)__popc(hj80) << 16);
            const Pair pm = both_halves(mine);
            const uint32_t both = pm.lo + pm.hi;
            const uint32_t abits = both & 0xffu, sbits = (both >> 8) & 0xffu, nactors = both >> 16;
            // two actors share a job <=> the sum of their one-hots has fewer bits set than there are actors
            const bool crowd = (uint32_t)__popc(abits) != nactors;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(crowd) != 0ull, 0)) {
                // two agents work on the SAME job in one env of this wave: the actors in turn order over BOTH words (base.py:377-382);
                // jobs are independent of each other, so the turns are the outer, ROLLED loop (rare code, kept small)
                const Pair pr = both_halves(R), pi = both_halves(w.im80), ph = both_halves(hj80), ps = both_halves(sel);
                const uint32_t R2[2] = {pr.lo, pr.hi}, im2[2] = {pi.lo, pi.hi}, hj2[2] = {ph.lo, ph.hi}, sel2[2] = {ps.lo, ps.hi};
                uint32_t f2[2] = {0, 0}, s2[2] = {0, 0};
                // (integer arithmetic on 0 / 1 values, not booleans: see step_swar)
                uint32_t jbits = 0; // job status as a bit mask
#pragma unroll
                for (int j = 0; j < 4; j++) jbits |= ((w.jd >> (8 * j)) & 1u) << j;
#pragma clang loop unroll(disable)
                for (uint32_t turn = 0; turn < (uint32_t)A; turn++) {
                    const uint32_t tb = (turn | 0x80u) * k01;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const uint32_t me = zero80(R2[q] ^ tb) & hj2[q]; // the agent whose turn it is, if it works on a job (one byte at most)
                        const uint32_t myjob = __builtin_amdgcn_sad_u8(sel2[q] & ff_from80(me), 0u, 0u);
                        const uint32_t actor = __builtin_amdgcn_sad_u8(me >> 7, 0u, 0u), imp = __builtin_amdgcn_sad_u8((me & im2[q]) >> 7, 0u, 0u);
                        const uint32_t status = (jbits >> myjob) & 1u;
                        const uint32_t ok = actor & ~(imp ^ status); // crew (0) on an open job (0), imposter (1) on a completed one (1)
                        jbits ^= ok << myjob;
                        const uint32_t f = ok & ~imp, sb = ok & imp;
                        e.m_fix += f;
                        e.m_sab += sb;
                        f2[q] |= me & (0u - f);
                        s2[q] |= me & (0u - sb);
                    }
                }
                w.jd = (jbits * 0x00204081u) & k01; // bits -> 0x01 per job byte
                fc80 = h ? f2[1] : f2[0];
                sc80 = h ? s2[1] : s2[0];
                idx4 += (fc80 >> 4) + (sc80 >> 5) + (sc80 >> 4); // RC_FIX 2, RC_SAB 3, times 4
                changed |= (f2[0] | f2[1] | s2[0] | s2[1]) != 0u;
            } else {
                // bits -> 0x01 per job byte (see step_swar); without a crowd every job has at most one actor, so the sums are ORs
                const uint32_t tog_env = (sbits * 0x00204081u) & k01;
                e.m_fix += (uint32_t)__popc(tog_env & ~w.jd);
                e.m_sab += (uint32_t)__popc(tog_env & w.jd);
                w.jd ^= tog_env;
                fc80 = succ & ~w.im80;
                sc80 = succ & w.im80;
                idx4 += (succ >> 4) + (sc80 >> 5); // RC_FIX 2 / RC_SAB 3, times 4
                changed |= tog_env != 0u;
            }
        }
    }
#endif

    // ---- check_win_condition: base.py:409-460 / pred_prey.py:78-99 --------------------------------------------------------------
    uint32_t wsel = 0u;
    done = false;
    if (W::kBase && J == 0) changed = true; // FourRoomEnv's "all jobs done" holds at every step when there are none (base.py:430)
#ifdef SUSNET_EXP_SKIP_WIN
    if (false) {
#elif defined(SUSNET_EXP_FREE_WIN)
    {
#else
    if (__builtin_amdgcn_ballot_w64(changed) != 0ull) { // (only when a kill landed / a job flipped this step: see step_swar)
#endif
        const uint32_t mine = (uint32_t)__popc(w.al80) | ((uint32_t)__popc(w.al80 & w.im80) << 8);
        const Pair pa = both_halves(mine);
        const uint32_t sum = pa.lo + pa.hi;
        const int alive_all = (int)(sum & 0xffu), alive_imp = (int)(sum >> 8);
        const int done_jobs = __popc(w.jd);
        bool crew_won, imp_won;
        if (!W::kBase) {
            crew_won = J != 0 && done_jobs == J;
            imp_won = !crew_won && alive_all - alive_imp == 0;
        } else {
            crew_won = alive_imp == 0 || done_jobs == J;
            imp_won = !crew_won && alive_all - alive_imp <= alive_imp;
        }
        done = crew_won || imp_won;
        e.flags |= (crew_won ? FLAG_CREW_WON : 0u) | (imp_won ? FLAG_IMP_WON : 0u);
        wsel = crew_won ? 16u : (imp_won ? 32u : 0u);
        idx4 += (wsel << 2) * k01; // the table's "crew won" / "imposters won" block
    }
    // ---- rewards: one lookup per agent in the host-evaluated table [win][index < n_imposters][dead][assignment code]; idx4 started
    // from the episode's base (Swar2::ridx) and collected this step's assignment codes and outcome where they arose
    {
#pragma unroll
#ifdef SUSNET_EXP_SKIP_REWARDS
        for (int i = 0; i < 4; i++) rr[i] = __uint_as_float(idx4 ^ (uint32_t)i);
#else
        for (int i = 0; i < 4; i++) rr[i] = lds_reward_lookup((idx4 >> (8 * i)) & 0xffu);
#endif
    }
    trunc = false; // base.py:392-395: t saturates at max_time_steps - 1
    if (e.t == (uint32_t)(c.max_t - 1)) trunc = true;
    else e.t += 1u;
}

} // namespace susnet
