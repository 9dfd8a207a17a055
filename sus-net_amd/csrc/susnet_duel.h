// susnet_duel.h -- the 1v1 ImposterTrainingGround game as straight-line register arithmetic, no branch on the stepping path: on a grid
// WITHOUT walls (BASELINE.json configs[1], the configuration the headline metric is quoted on: no LDS lookup either) and, round 5, on a
// WALL map (DuelWallTable below: the reference's own 1v1 experiments run on its four-room map).
//
// Reference behaviour (paths relative to the reference repo root):
//   ImposterTrainingGround          src/environment/pred_prey.py:20-99 (one imposter = agent 0, fixed order, no jobs here)
//   step / _agent_step / move       src/environment/base.py:332-407, 462-533, 69-79, 548-551
//   _merge_rewards + zero fill      src/environment/base.py:553-563, 389-390
//
// State of one environment (one lane): the four coordinates as bytes of ONE register, biased by +1 so that a step off the
// grid shows up as a byte value 0 or n + 1 (never a borrow into the neighbouring byte):
//   pq = (x0 + 1) | (y0 + 1) << 8 | (x1 + 1) << 16 | (y1 + 1) << 24
// A move adds a signed byte from an 6-entry table selected by the action (v_perm: one instruction looks both agents' deltas
// up); on a grid without walls _is_valid_position() only checks the bounds, so an invalid move is exactly "the changed
// byte left [1, n]" and is undone bytewise.  Rewards: with both orders of events fixed (agent 0 acts first) a tick's
// reward pair is a function of three bits {kill landed, imposter dead, crew dead}; the host evaluates the reference's
// assign -> merge -> zero-fill chain for the 8 combinations per agent (from the same table the other kernels index in LDS)
// and the kernel looks the result up in a byte table held in registers (selected only when all of them are integers in
// [-127, 127]; otherwise the handle runs the table kernels).
#pragma once

#include "susnet_device.h"
#include "susnet_obs.h"

namespace susnet {

struct Duel {
    uint32_t pq;    // biased coordinates, see above
    uint32_t lv;    // who is alive, AS THE MASK of their coordinate bytes: 0x0000ffff imposter (agent 0) | 0xffff0000 crew member
                    // (a dead agent's move is masked with it in one instruction; flags, indices and stored bytes are bit picks of it)
    uint32_t m0, m1; // wall maps only (DuelWallTable): bit a = role-relative move action a is blocked on the cell agent 0 / 1 stands on
};

// ---- the same game on a WALL map (the reference's default: include_walls=True, base.py:119; notebooks/experiment_1v1.ipynb envs['Wall']) ----
// _is_valid_position (base.py:548-551: the bounds, then grid[pos[1], pos[0]] -- transposed, like every lookup the step makes) decides a
// move from the cell the agent stands on and the action alone, and a 1v1 tick has two of them.  So the fused rollout keeps, per CELL, the
// four answers as bits 1..4 of one byte -- built at launch from the (action, cell) move table the other kernels look their destinations up
// in (Consts::move_tab: a blocked move leaves the cell unchanged) -- indexed by the agent's two BIASED coordinate bytes as they lie in
// `pq` ((y + 1) << 8 | (x + 1): one AND / one shift per agent, no index arithmetic), and reads the bytes of the cells the agents have just
// ARRIVED on at the end of a tick: a whole tick of other work covers the LDS round trip.  The step itself is the no-walls one with the
// bounds test replaced by two bit picks (the table knows the border too): no LDS wait and no branch on the stepping path.
constexpr uint32_t kDuelWallWords = 18u * 64u; // rows y + 1 = 0 .. 17 of 256 bytes (N <= 16)
struct DuelWallTable {
    typedef __attribute__((address_space(3))) uint8_t *lds_u8_wptr;
    static constexpr uint32_t kBase = 4u * kTableWords; // LDS byte address: behind the table image (SpecCfg2 has no store columns, no group words)
    // all 64 lanes, after the table image has been published
    static __device__ __forceinline__ void build(int N, int tid) {
        for (int idx = tid; idx < N * N; idx += kBlock) {
            const uint32_t x = (uint32_t)(idx % N), y = (uint32_t)(idx / N), cell = x | (y << 4);
            uint32_t m = 0;
#pragma unroll
            for (uint32_t a = 1; a <= 4; a++) m |= (lds_move_lookup((a << 8) | cell) == cell ? 1u : 0u) << a; // STAY / KILL (bits 0, 5) never move
            *(lds_u8_wptr)(uintptr_t)(kBase + (((y + 1u) << 8) | (x + 1u))) = (uint8_t)m;
        }
    }
    static __device__ __forceinline__ void lookup(Duel &d) {
        d.m0 = *(lds_u8_ptr)(uintptr_t)(kBase + (d.pq & 0xffffu));
        d.m1 = *(lds_u8_ptr)(uintptr_t)(kBase + (d.pq >> 16));
    }
};
// alive flags as flatten_state stores them next to the cells: byte 0 = imposter alive, byte 1 = crew member alive
__device__ __forceinline__ uint32_t duel_alive_bytes(const Duel &d) { return __builtin_amdgcn_perm(0u, d.lv, 0x0c0c0200u) & 0x0101u; }

// wave-uniform constants of a launch (scalar registers)
struct DuelConsts {
    uint32_t hi_probe;  // byte + hi_probe has bit 7 set  <=>  byte == n + 1   (0x7f - n in every byte)
    uint32_t lut0_lo, lut0_hi, lut1_lo, lut1_hi; // reward bytes: index = kill landed | imposter dead << 1 | crew dead << 2
    uint32_t max_t_m1;
};
__device__ __forceinline__ DuelConsts make_duel_consts(const Consts &c) {
    DuelConsts k;
    k.hi_probe = (0x7fu - (uint32_t)c.N) * k01;
    k.lut0_lo = (uint32_t)c.duel_lut[0]; k.lut0_hi = (uint32_t)(c.duel_lut[0] >> 32);
    k.lut1_lo = (uint32_t)c.duel_lut[1]; k.lut1_hi = (uint32_t)(c.duel_lut[1] >> 32);
    k.max_t_m1 = (uint32_t)(c.max_t - 1);
    return k;
}

template <class Store>
__device__ __forceinline__ void to_duel(const Store &st, const Env &e, Duel &d) {
    const uint32_t c0 = st.xy(0), c1 = st.xy(1);
    d.pq = ((c0 & 15u) | ((c0 >> 4) << 8) | ((c1 & 15u) << 16) | ((c1 >> 4) << 24)) + k01;
    d.lv = ((0u - (e.alive & 1u)) & 0x0000ffffu) | ((0u - ((e.alive >> 1) & 1u)) & 0xffff0000u);
    d.m0 = d.m1 = 0u;
}
template <class Store>
__device__ __forceinline__ void from_duel(const Duel &d, Store &st, Env &e) {
    const uint32_t p = d.pq - k01;
    st.set_xy(0, (p & 15u) | (((p >> 8) & 15u) << 4));
    st.set_xy(1, ((p >> 16) & 15u) | (((p >> 24) & 15u) << 4));
    e.alive = (d.lv & 1u) | ((d.lv >> 31) << 1);
}

// The step proper, with role-relative actions a0 in [0, 6) (5 = KILL, pred_prey.py:12-19) and a1 in [0, 5): cells, who is alive, the
// landed kill, the rewards (float32), done.  Everything an episode merely COUNTS (steps, t, kills, flags, the kill's word of the
// event stream) is duel_step's, below.
template <bool WALLS = false>
__device__ __forceinline__ void duel_core(const DuelConsts &k, Duel &d, uint32_t a0, uint32_t a1, float &r0, float &r1, uint32_t &done, uint32_t &hit) {
    // KILL first (agent 0 acts first, base.py:377-382 with a fixed order): both alive, same cell (base.py:490-515)
    const uint32_t same = ((d.pq >> 16) == (d.pq & 0xffffu)) ? 1u : 0u;
    hit = (a0 == 5u ? 1u : 0u) & same & (d.lv == 0xffffffffu ? 1u : 0u);
    d.lv &= hit ? 0x0000ffffu : 0xffffffffu; // base.py:511
    // moves (base.py:484-487): table of (dx, dy) per action -- STAY, UP (y + 1), DOWN, LEFT (x - 1), RIGHT, KILL (no move)
    const uint32_t selx = a0 | (a1 << 16) | 0x0c000c00u;       // byte 0 <- table[a0], byte 2 <- table[a1]
    // (v_perm: selector values 0..3 pick bytes of the SECOND operand, 4..7 of the first)
    // a step of -1 is held as 0x7f: byte + 0x7f, then bit 7 flipped, is byte - 1 without a carry into the neighbouring byte
    const uint32_t dx = __builtin_amdgcn_perm(0x00000001u, 0x7f000000u, selx); // dx by action: 0 0 0 -1 | +1 0
    const uint32_t dy = __builtin_amdgcn_perm(0x00000000u, 0x007f0100u, selx); // dy by action: 0 +1 -1 0 | 0 0
    uint32_t delta = dx | (dy << 8);
    // dead agents do not act (base.py:477); a crew member killed this tick never gets its turn
    delta &= d.lv;
    if constexpr (WALLS) {
        // _is_valid_position (base.py:548-551) from the cells' blocked-move bits: a blocked agent's two delta bytes are cleared
        const uint32_t b0 = __builtin_amdgcn_ubfe(d.m0, a0, 1u), b1 = __builtin_amdgcn_ubfe(d.m1, a1, 1u);
        delta &= ~(((0u - b0) & 0x0000ffffu) | ((0u - b1) << 16));
        d.pq = (d.pq + delta) ^ ((delta << 1) & k80);
        DuelWallTable::lookup(d); // for the NEXT tick (issued here, waited for there)
    } else {
        const uint32_t q = (d.pq + delta) ^ ((delta << 1) & k80); // bytes in [0, n + 1]
        // undo a step off the grid: byte == 0 or byte == n + 1 (_is_valid_position, base.py:548-551: bounds only, no walls here)
        const uint32_t bad80 = ((q + k.hi_probe) | ~((q | k80) - k01)) & k80;
        const uint32_t badff = (bad80 - (bad80 >> 7)) | bad80;
        d.pq = (d.pq & badff) | (q & ~badff);
    }
    // win (pred_prey.py:78-99 with no jobs): the imposter wins when no crew member is alive
    const uint32_t n = ~d.lv; // bit 1: imposter dead, bits 18 / 31: crew member dead
    done = n >> 31;
    // rewards: byte tables over {kill landed, imposter dead, crew dead} = selector bits 0, 1, 2 (only byte 0 of the picked word is
    // read below, so whatever else the selector's upper bytes pick does not matter)
    const uint32_t t = n & 0x00040002u;
    const uint32_t idx = hit | t | (t >> 16);
    const int32_t b0 = (int32_t)(int8_t)__builtin_amdgcn_perm(k.lut0_hi, k.lut0_lo, idx);
    const int32_t b1 = (int32_t)(int8_t)__builtin_amdgcn_perm(k.lut1_hi, k.lut1_lo, idx);
    r0 = (float)b0;
    r1 = (float)b1;
}

// One step.  Out: rewards, done, truncated.  e: t, flags, info counters; cur: the env's event-stream cursor (production protocol: a
// landed kill takes one word -- its value is never needed with a single candidate; numpy draws nothing there, base.py:497).
// hit_out: the caller advances the cursor itself (k_rollout_duel: inside its episode-end branch).
template <bool NUMPY, bool WALLS = false>
__device__ __forceinline__ void duel_step(const DuelConsts &k, Duel &d, Env &e, uint64_t &cur, uint32_t a0, uint32_t a1, float &r0, float &r1,
                                          uint32_t &done, uint32_t &trunc, uint32_t *hit_out = nullptr) {
    uint32_t hit;
    duel_core<WALLS>(k, d, a0, a1, r0, r1, done, hit);
    e.m_steps += 1;                      // base.py:366
    e.m_kv += hit;                       // IMP_KILLED_CREW, base.py:508
    if (hit_out) *hit_out = hit;
    else if (!NUMPY) cur += (uint64_t)hit;
    e.flags |= done << 2;                // FLAG_IMP_WON (metrics.update, pred_prey.py:96)
    // base.py:392-395: t saturates at max_time_steps - 1
    trunc = e.t == k.max_t_m1 ? 1u : 0u;
    e.t += trunc ^ 1u;
}

} // namespace susnet
