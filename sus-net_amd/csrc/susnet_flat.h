// susnet_flat.h -- compile-time writers for the FlatFeaturizer layouts the reference's live experiments use.
//
// Reference behaviour (paths relative to the reference repo root):
//   FlatFeaturizer.fit                 src/features/model_ready.py:309-370 (float32 rows, components concatenated)
//   OneHotAgentPositionFeaturizer      src/features/component.py:221-247   [onehot_N(x) | onehot_N(y)] per agent, zeros if dead
//   AliveCrewFeaturizer                src/features/component.py:406-425   alive flags of agents 1 .. A-1
//   ClosestAliveCrewFeaturizer         src/features/component.py:455-482   one-hot argmin L1 distance to agent 0 (dead = N + N, first minimum)
// The two layouts: `onehot_pos` (notebooks/experiment_1v1.ipynb: 36 floats on the 9x9 1v1 game) and `onehot_pos + alive_crew +
// closest_crew` (88 floats on the 14x14 1v2 game, BASELINE config 5's policy input).
//
// Every element of these rows is 0.0f or 1.0f, so a row is a BIT MASK (2 or 3 registers) that the owning lane builds with a few
// shifts; the wave then writes its 64 rows -- which are contiguous in the [B][F] output -- cooperatively: masks to LDS (768
// bytes), and every lane expands float4 chunks of the flat image, one 16-byte store per chunk, each store instruction covering
// 1 KiB of consecutive bytes.  No byte image, no zero fill, no run-time component list (susnet_obs.h keeps serving every other
// layout); rows with F % 4 == 0 never share a float4 with their neighbour.
#pragma once

#include "susnet_device.h"

namespace susnet {

enum : int { FEAT_ONEHOT = 1, FEAT_ONEHOT_ALIVE_CLOSEST = 2, FEAT_COORD = 3 };

template <int FEAT, int A_, int N_>
struct FlatRow {
    static constexpr int A = A_, N = N_, kOneHot = A * 2 * N;
    static constexpr int F = kOneHot + (FEAT == FEAT_ONEHOT_ALIVE_CLOSEST ? 2 * (A - 1) : 0);
    // what the Q-network kernel (susnet_qnet.h) needs to know of a layout: the bits behind the position one-hots, and that a dead agent's
    // positions are all zero (component.py:226-240)
    static constexpr int kTailBits = F - kOneHot;
    static constexpr bool kDeadZero = true;
    static_assert(F % 4 == 0 && A >= 2 && N <= 16, "float4 chunks must not straddle rows");
    static constexpr int MW = (F + 31) / 32; // mask words per row
    static constexpr int C = F / 4;          // float4 chunks per row
    uint32_t m[MW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int w = 0; w < MW; w++) m[w] = 0u;
    }
    // set bit `pos` if `on`; pos is known to lie in [LO, HI): only the words that range touches are updated
    template <int LO, int HI>
    __device__ __forceinline__ void set(uint32_t pos, bool on) {
        const uint32_t bit = on ? (1u << (pos & 31u)) : 0u;
        constexpr int w0 = LO / 32, w1 = (HI - 1) / 32;
#pragma unroll
        for (int w = w0; w <= w1; w++) m[w] |= (w0 == w1 || (pos >> 5) == (uint32_t)w) ? bit : 0u;
    }
    // x[i], y[i] < N, alive[i] in {0, 1}
    __device__ __forceinline__ void build(const uint32_t (&x)[A], const uint32_t (&y)[A], const uint32_t (&alive)[A]) {
        clear();
        static_for_agents<0>(x, y, alive);
        if constexpr (FEAT == FEAT_ONEHOT_ALIVE_CLOSEST) {
#pragma unroll
            for (int i = 1; i < A; i++) m[(kOneHot + i - 1) / 32] |= alive[i] << ((kOneHot + i - 1) & 31); // component.py:411-421
            uint32_t best = 0, bestd = 1u << 20; // component.py:460-478: argmin, first minimum, dead = N + N
#pragma unroll
            for (int i = 1; i < A; i++) {
                const int dx = (int)x[0] - (int)x[i], dy = (int)y[0] - (int)y[i];
                const uint32_t d = alive[i] ? (uint32_t)((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy)) : (uint32_t)(2 * N);
                const bool lt = d < bestd;
                bestd = lt ? d : bestd;
                best = lt ? (uint32_t)(i - 1) : best;
            }
            set<kOneHot + (A - 1), kOneHot + 2 * (A - 1)>((uint32_t)(kOneHot + (A - 1)) + best, true);
        }
    }

  private:
    template <int I>
    __device__ __forceinline__ void static_for_agents(const uint32_t (&x)[A], const uint32_t (&y)[A], const uint32_t (&alive)[A]) {
        if constexpr (I < A) { // component.py:226-240: all zeros for a dead agent
            set<I * 2 * N, I * 2 * N + N>((uint32_t)(I * 2 * N) + x[I], alive[I] != 0u);
            set<I * 2 * N + N, I * 2 * N + 2 * N>((uint32_t)(I * 2 * N + N) + y[I], alive[I] != 0u);
            static_for_agents<I + 1>(x, y, alive);
        }
    }
};

// CoordinateAgentPositionsFeaturizer (src/features/component.py:384-403): the row [x0, y0, x1, y1, ...] as floats, NOT zeroed for a dead
// agent -- the layout of the reference's `no_wall_coord_features` / `wall_coord_features` experiments (notebooks/experiment_1v1.ipynb).
// Not a 0/1 row, so it has no bit-mask writer here (the generic observation writer serves it); it exists for the Q-network kernel, whose
// first layer then gathers one LDS row per COORDINATE VALUE -- k times the coordinate's column of W1, multiplied on the host -- with
// the very instructions that gather the one-hot layouts' columns: `kOneHot` counts those rows, F is the network's input width.
template <int A_, int N_>
struct CoordRow {
    static constexpr int A = A_, N = N_, kOneHot = A * 2 * N, F = 2 * A, kTailBits = 0;
    static constexpr bool kDeadZero = false;
    uint32_t m[1];
    __device__ __forceinline__ void build(const uint32_t (&)[A], const uint32_t (&)[A], const uint32_t (&)[A]) { m[0] = 0u; }
};

// The wave's rows -> [nrows][F] float32 at byte offset `base` of the buffer `r` (rows of lanes >= nrows are not written).
// `lds`: 64 * MW words of staging.  All 64 lanes must call (inactive ones with any row).
template <class ROW>
__device__ __forceinline__ void flat_store_wave(const ROW &row, uint32_t *lds, int tid, int nrows, __amdgpu_buffer_rsrc_t r, uint32_t base) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int w = 0; w < ROW::MW; w++) lds[tid * ROW::MW + w] = row.m[w];
    wave_lds_publish();
    const uint32_t total = (uint32_t)nrows * (uint32_t)ROW::C;
#pragma unroll
    for (int p = 0; p < ROW::C; p++) { // 64 * C chunks, 64 lanes
        const uint32_t u = (uint32_t)tid + 64u * (uint32_t)p;
        if (u < total) {
            const uint32_t rr = u / (uint32_t)ROW::C, c4 = u - rr * (uint32_t)ROW::C;
            const uint32_t nib = lds[rr * ROW::MW + ((4u * c4) >> 5)] >> ((4u * c4) & 31u);
            // bit -> 0.0f / 1.0f: sign-extend the bit, keep the exponent pattern of 1.0f
            const u32x4 v = {(uint32_t)((int32_t)(nib << 31) >> 31) & 0x3f800000u, (uint32_t)((int32_t)(nib << 30) >> 31) & 0x3f800000u,
                             (uint32_t)((int32_t)(nib << 29) >> 31) & 0x3f800000u, (uint32_t)((int32_t)(nib << 28) >> 31) & 0x3f800000u};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, base + 16u * u, 0, 0);
        }
    }
    wave_lds_publish(); // the masks are rewritten next tick
}

// which compiled-in configurations have such a writer: the layout bench.py / the policy loop use on them
template <class S>
struct FlatFor { static constexpr bool kOk = false; static constexpr int kFeat = 0; using Row = FlatRow<FEAT_ONEHOT, 2, 9>; };
// BASELINE configs[1] (1v1 on the 9x9 grid): `onehot_pos`, 36 floats (notebooks/experiment_1v1.ipynb)
template <>
struct FlatFor<Spec<2, 0, SUSNET_VARIANT_ITG, 0, 0, 1>> { static constexpr bool kOk = true; static constexpr int kFeat = FEAT_ONEHOT; using Row = FlatRow<FEAT_ONEHOT, 2, 9>; };
// BASELINE configs[2] / [4] (1v2 on the 14x14 grid): `onehot_pos + alive_crew + closest_crew`, 88 floats (the policy's input)
template <>
struct FlatFor<Spec<3, 4, SUSNET_VARIANT_BASE, 1, -1, 1>> { static constexpr bool kOk = true; static constexpr int kFeat = FEAT_ONEHOT_ALIVE_CLOSEST; using Row = FlatRow<FEAT_ONEHOT_ALIVE_CLOSEST, 3, 14>; };

} // namespace susnet
