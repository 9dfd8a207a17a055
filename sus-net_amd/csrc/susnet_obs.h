// susnet_obs.h -- fused observation writers.
//
// Reference layouts reproduced (float32 unless the caller asks for compact bytes):
//   RAW     env.flatten_state            src/environment/base.py:234-235 over observation_space 211-228
//                                        (tagging.py:42-60 appends used, counts, time-left)
//   FLAT    FlatFeaturizer               src/features/model_ready.py:309-354 over the component
//                                        featurizers of src/features/component.py:221-482
//   PLANES  GlobalFeaturizer             src/features/model_ready.py:230-247 (+ component.py:83-131)
//
// Mechanism: the lane that owns an environment first writes that environment's observation ROW into an
// LDS staging area in a compact form (one bit per element for the 0/1 planes, one byte per element for
// everything else).  Then the whole wave streams the 64 rows -- which are contiguous in the [B][F] output
// -- to HBM with 16-byte stores, expanding bits/bytes to float32 on the way.  A lane therefore never
// writes a strided row of its own: every global store instruction covers 1 KiB of consecutive bytes.
#pragma once

#include "susnet_device.h"

namespace susnet {

struct ObsArgs {
    int32_t mode;   // SUSNET_OBS_*
    int32_t dtype;  // SUSNET_F32 or SUSNET_U8
    int32_t ncomp;
    int32_t comp[16];
    int32_t F, F2;          // elements per env of out / out2
    int32_t words1, words2; // LDS staging words per row (odd) for segment 1 / 2
    void *out, *out2;
    int64_t tick_stride, tick_stride2; // rollout: elements between consecutive ticks
};

__host__ __device__ inline int odd_words(int w) { return (w & 1) ? w : w + 1; }

__host__ __device__ inline int flat_component_size(int comp, int A, int N, int n_crew) {
    switch (comp) {
    case SUSNET_F_ONEHOT_POS: return A * 2 * N;
    case SUSNET_F_COORD_POS: return 2 * A;
    case SUSNET_F_ALIVE_CREW: return A - 1;
    case SUSNET_F_L1_CREW: return n_crew;
    case SUSNET_F_CLOSEST_CREW: return n_crew;
    case SUSNET_F_WALLS3X3: return 9;
    case SUSNET_F_DIST_TO_IMP: return (A - 1) * 2;
    case SUSNET_F_ROOM_LOC: return 8;
    default: return -1;
    }
}

// ---- row fill (owning lane) -----------------------------------------------------------------------
__device__ __forceinline__ void stage_zero(uint32_t *row, int words) {
    for (int w = 0; w < words; w++) row[w] = 0u;
}
__device__ __forceinline__ void stage_byte(uint32_t *row, int f, int v) {
    reinterpret_cast<uint8_t *>(row)[f] = (uint8_t)(int8_t)v;
}
__device__ __forceinline__ void stage_bit(uint32_t *row, int f) { row[f >> 5] |= 1u << (f & 31); }

__device__ __forceinline__ void fill_raw(const Consts &c, const Lds &L, int tid, const Env &e, uint32_t *row) {
    const int A = c.A, J = c.J;
    int k = 0;
    for (int i = 0; i < A; i++) {
        uint32_t w = L.xy[i * kBlock + tid];
        stage_byte(row, k++, w & 15u);
        stage_byte(row, k++, (w >> 4) & 15u);
    }
    for (int i = 0; i < A; i++) stage_byte(row, k++, (e.alive >> i) & 1u);
    const bool tagging = c.variant == SUSNET_VARIANT_TAGGING;
    if (J > 0 || tagging) {
        for (int j = 0; j < J; j++) {
            uint32_t w = L.job[j * kBlock + tid];
            stage_byte(row, k++, w & 15u);
            stage_byte(row, k++, (w >> 4) & 15u);
        }
        for (int j = 0; j < J; j++) stage_byte(row, k++, (e.jd >> j) & 1u);
    }
    if (tagging) {
        for (int i = 0; i < A; i++) stage_byte(row, k++, (e.used >> i) & 1u);
        for (int i = 0; i < A; i++) stage_byte(row, k++, (L.xy[i * kBlock + tid] >> 8) & 0xffu);
        stage_byte(row, k++, c.tag_interval - (int)e.timer);
    }
}

__device__ __forceinline__ void fill_flat(const Consts &c, const ObsArgs &o, const Lds &L, int tid, const Env &e, uint32_t *row) {
    const int A = c.A, N = c.N, NC = c.n_crew;
    const uint32_t w0 = L.xy[tid];
    const int ix = (int)(w0 & 15u), iy = (int)((w0 >> 4) & 15u); // "imposter" = agent 0 (component.py:262,289,...)
    int k = 0;
    for (int ci = 0; ci < o.ncomp; ci++) {
        const int comp = o.comp[ci];
        switch (comp) {
        case SUSNET_F_ONEHOT_POS: // component.py:226-240
            for (int i = 0; i < A; i++)
                if ((e.alive >> i) & 1u) {
                    uint32_t w = L.xy[i * kBlock + tid];
                    stage_byte(row, k + i * 2 * N + (int)(w & 15u), 1);
                    stage_byte(row, k + i * 2 * N + N + (int)((w >> 4) & 15u), 1);
                }
            break;
        case SUSNET_F_COORD_POS: // component.py:389-399
            for (int i = 0; i < A; i++) {
                uint32_t w = L.xy[i * kBlock + tid];
                stage_byte(row, k + 2 * i, w & 15u);
                stage_byte(row, k + 2 * i + 1, (w >> 4) & 15u);
            }
            break;
        case SUSNET_F_ALIVE_CREW: // component.py:411-421
            for (int i = 1; i < A; i++)
                if ((e.alive >> i) & 1u) stage_byte(row, k + i - 1, 1);
            break;
        case SUSNET_F_L1_CREW: // component.py:433-448
            for (int i = 1; i < A; i++) {
                uint32_t w = L.xy[i * kBlock + tid];
                int d = abs(ix - (int)(w & 15u)) + abs(iy - (int)((w >> 4) & 15u));
                stage_byte(row, k + i - 1, ((e.alive >> i) & 1u) ? d : -1);
            }
            break;
        case SUSNET_F_CLOSEST_CREW: { // component.py:460-478: argmin, first minimum, dead = N + N
            int best = 0, bestd = 1 << 20;
            for (int i = 1; i < A; i++) {
                uint32_t w = L.xy[i * kBlock + tid];
                int d = ((e.alive >> i) & 1u) ? abs(ix - (int)(w & 15u)) + abs(iy - (int)((w >> 4) & 15u)) : N + N;
                if (d < bestd) { bestd = d; best = i - 1; }
            }
            if (NC > 0) stage_byte(row, k + best, 1);
            break;
        }
        case SUSNET_F_WALLS3X3: // component.py:286-296: zero-padded grid[x, y] around agent 0
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    int gx = ix + a - 1, gy = iy + b - 1;
                    bool in = (unsigned)gx < (unsigned)N && (unsigned)gy < (unsigned)N;
                    uint32_t rw = L.grid[in ? gx : 0];
                    if (in && ((rw >> gy) & 1u)) stage_byte(row, k + a * 3 + b, 1);
                }
            break;
        case SUSNET_F_DIST_TO_IMP: { // component.py:255-273: alive non-0 agents packed left
            int p = 0;
            for (int i = 1; i < A; i++)
                if ((e.alive >> i) & 1u) {
                    uint32_t w = L.xy[i * kBlock + tid];
                    stage_byte(row, k + p, ix - (int)(w & 15u));
                    stage_byte(row, k + p + 1, iy - (int)((w >> 4) & 15u));
                    p += 2;
                }
            break;
        }
        case SUSNET_F_ROOM_LOC: { // component.py:8-17,308-329 (9x9 quadrants)
            uint32_t cnt = 0; // eight 4-bit counters (at most 15 agents per bucket)
            for (int i = 0; i < A; i++) {
                if (!((e.alive >> i) & 1u)) continue;
                uint32_t w = L.xy[i * kBlock + tid];
                int x = (int)(w & 15u), y = (int)((w >> 4) & 15u);
                int room = (x < 5) ? (y < 5 ? 0 : 1) : (y >= 5 ? 2 : 3);
                cnt += 1u << (4 * (((i == 0) ? 0 : 4) + room));
            }
            for (int r = 0; r < 8; r++) stage_byte(row, k + r, (cnt >> (4 * r)) & 15u);
            break;
        }
        default: break;
        }
        k += flat_component_size(comp, A, N, NC);
    }
}

__device__ __forceinline__ void fill_planes(const Consts &c, const Lds &L, int tid, const Env &e, uint32_t *bits, uint32_t *bytes) {
    const int A = c.A, J = c.J, N = c.N, NN = N * N;
    for (int i = 0; i < A; i++) // component.py:90-100: channel i, [x][y], only if alive
        if ((e.alive >> i) & 1u) {
            uint32_t w = L.xy[i * kBlock + tid];
            stage_bit(bits, i * NN + (int)(w & 15u) * N + (int)((w >> 4) & 15u));
        }
    for (int j = 0; j < J; j++) { // component.py:116-127: channel A + int(done)
        uint32_t w = L.job[j * kBlock + tid];
        stage_bit(bits, (A + (int)((e.jd >> j) & 1u)) * NN + (int)(w & 15u) * N + (int)((w >> 4) & 15u));
    }
    int k = 0; // model_ready.py:237-247: [alive, (tag_counts), job_status]
    for (int i = 0; i < A; i++) stage_byte(bytes, k++, (e.alive >> i) & 1u);
    if (c.variant == SUSNET_VARIANT_TAGGING)
        for (int i = 0; i < A; i++) stage_byte(bytes, k++, (L.xy[i * kBlock + tid] >> 8) & 0xffu);
    for (int j = 0; j < J; j++) stage_byte(bytes, k++, (e.jd >> j) & 1u);
}

// ---- cooperative wave store -------------------------------------------------------------------------
// rows [0, nrows) of `stage` (row stride `words`) -> out[(b0 + row) * F + f], 4 elements per lane per pass.
template <bool BITS>
__device__ __forceinline__ void store_rows(const uint32_t *stage, int words, int F, int nrows, int64_t b0, void *out,
                                           int dtype, bool is_signed, int lane) {
    if (!out || F <= 0) return;
    const int total = nrows * F;
    const int64_t base = b0 * (int64_t)F;
    for (int g = lane * 4; g < total; g += kWave * 4) {
        int row = g / F, f = g - row * F;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int r = row, ff = f + k;
            while (ff >= F) { ff -= F; r++; }
            int val = 0;
            if (g + k < total) {
                if (BITS) val = (stage[r * words + (ff >> 5)] >> (ff & 31)) & 1u;
                else {
                    uint8_t by = reinterpret_cast<const uint8_t *>(stage + r * words)[ff];
                    val = is_signed ? (int)(int8_t)by : (int)by;
                }
            }
            v[k] = val;
        }
        if (dtype == SUSNET_F32) {
            float *o = reinterpret_cast<float *>(out) + base + g;
            if (g + 3 < total) {
                float4 q = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                *reinterpret_cast<float4 *>(o) = q;
            } else {
                for (int k = 0; k < 4 && g + k < total; k++) o[k] = (float)v[k];
            }
        } else {
            uint8_t *o = reinterpret_cast<uint8_t *>(out) + base + g;
            if (g + 3 < total) {
                uint32_t q = (uint32_t)(v[0] & 0xff) | ((uint32_t)(v[1] & 0xff) << 8) | ((uint32_t)(v[2] & 0xff) << 16) |
                             ((uint32_t)(v[3] & 0xff) << 24);
                *reinterpret_cast<uint32_t *>(o) = q;
            } else {
                for (int k = 0; k < 4 && g + k < total; k++) o[k] = (uint8_t)v[k];
            }
        }
    }
}

// Fill this lane's row(s), then stream the wave's rows. `active` = this lane owns a real env.
// `tick` selects the trajectory slice in rollouts (0 otherwise).
__device__ __forceinline__ void write_obs(const Consts &c, const ObsArgs &o, const Lds &L, int tid, const Env &e, bool active,
                                          int64_t b0, int nrows, int64_t tick) {
    if (o.mode == SUSNET_OBS_NONE) return;
    uint32_t *seg1 = L.stage;
    uint32_t *seg2 = L.stage + kBlock * o.words1;
    uint32_t *row1 = seg1 + tid * o.words1;
    uint32_t *row2 = seg2 + tid * o.words2;
    if (active) {
        stage_zero(row1, o.words1);
        if (o.mode == SUSNET_OBS_RAW) fill_raw(c, L, tid, e, row1);
        else if (o.mode == SUSNET_OBS_FLAT) fill_flat(c, o, L, tid, e, row1);
        else {
            stage_zero(row2, o.words2);
            fill_planes(c, L, tid, e, row1, row2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int esz = (o.dtype == SUSNET_F32) ? 4 : 1;
    char *out1 = o.out ? reinterpret_cast<char *>(o.out) + tick * o.tick_stride * esz : nullptr;
    if (o.mode == SUSNET_OBS_PLANES) {
        store_rows<true>(seg1, o.words1, o.F, nrows, b0, out1, o.dtype, false, tid);
        char *out2 = o.out2 ? reinterpret_cast<char *>(o.out2) + tick * o.tick_stride2 * esz : nullptr;
        store_rows<false>(seg2, o.words2, o.F2, nrows, b0, out2, o.dtype, false, tid);
    } else {
        store_rows<false>(seg1, o.words1, o.F, nrows, b0, out1, o.dtype, o.mode == SUSNET_OBS_FLAT, tid);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

} // namespace susnet
