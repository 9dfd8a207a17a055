// susnet_obs.h -- fused observation writers.
//
// Reference layouts reproduced (float32 unless the caller asks for compact bytes):
//   RAW     env.flatten_state            src/environment/base.py:234-235 over observation_space 211-228
//                                        (tagging.py:42-60 appends used, counts, time-left)
//   FLAT    FlatFeaturizer               src/features/model_ready.py:309-354 over the component
//                                        featurizers of src/features/component.py:221-482
//   PLANES  GlobalFeaturizer             src/features/model_ready.py:230-247 (+ component.py:83-131)
//
// Mechanism: the 64 rows a wave owns are CONTIGUOUS in the [B][F] output, so the wave builds exactly that
// byte image in LDS -- one byte per element (small integers), or one bit per element for the 0/1 planes --
// each lane writing only its own environment's row, and then streams the image to HBM with 16-byte
// stores (bytes copied as they are for uint8 output, expanded to float32 otherwise).  No lane ever writes
// a strided row of its own: every global store instruction covers up to 1 KiB of consecutive bytes.
#pragma once

#include "susnet_device.h"

namespace susnet {

struct ObsArgs {
    int32_t mode;   // SUSNET_OBS_*
    int32_t dtype;  // SUSNET_F32 or SUSNET_U8
    int32_t ncomp;
    int32_t comp[16];
    int32_t F, F2;          // elements per env of out / out2
    int32_t Fi, F2i;        // elements per env of the LDS images (= F, F2 except PERSP: the planes image feeds A rotated copies)
    int32_t words1, words2; // LDS staging words of segment 1 / 2 (whole wave)
    int32_t flat_feat;      // FLAT float32 whose component list is one of the compiled-in layouts (susnet_flat.h FEAT_*), else 0
    int32_t pad_;
    void *out, *out2;
    int64_t tick_stride, tick_stride2; // rollout: elements between consecutive ticks
};

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

// A workgroup is ONE wave and the LDS executes a wave's instructions in issue order, so hand-offs between
// lanes need only a compiler-level ordering point -- in particular NOT a wait for outstanding global stores
// (a workgroup-scope fence would emit s_waitcnt vmcnt(0) every tick).
__device__ __forceinline__ void wave_lds_fence() { wave_lds_publish(); }

// ---- row fill (owning lane); `row` = this env's F bytes inside the packed image -----------------------
__device__ __forceinline__ void put_b(uint8_t *row, int f, int v) { row[f] = (uint8_t)(int8_t)v; }

template <class S, class Store>
__device__ __forceinline__ void fill_raw(const Consts &c, const Store &st, const Env &e, uint8_t *row) {
    const int A = S::A(c), J = S::J(c);
    int k = 0;
#pragma unroll
    for (int i = 0; i < A; i++) {
        uint32_t w = st.xy(i);
        put_b(row, k++, w & 15u);
        put_b(row, k++, (w >> 4) & 15u);
    }
#pragma unroll
    for (int i = 0; i < A; i++) put_b(row, k++, (e.alive >> i) & 1u);
    const bool tagging = S::tagging(c);
    if (J > 0 || tagging) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            uint32_t w = st.job(j);
            put_b(row, k++, w & 15u);
            put_b(row, k++, (w >> 4) & 15u);
        }
#pragma unroll
        for (int j = 0; j < J; j++) put_b(row, k++, (e.jd >> j) & 1u);
    }
    if (tagging) {
        for (int i = 0; i < A; i++) put_b(row, k++, (e.used >> i) & 1u);
        for (int i = 0; i < A; i++) put_b(row, k++, st.cnt(i));
        put_b(row, k++, c.tag_interval - (int)e.timer);
    }
}

template <class S, class Store>
__device__ __forceinline__ void fill_flat(const Consts &c, const ObsArgs &o, const Tables &T, const Store &st, const Env &e, uint8_t *row) {
    const int A = S::A(c), N = c.N, NC = c.n_crew;
    const uint32_t w0 = st.xy(0);
    const int ix = (int)(w0 & 15u), iy = (int)((w0 >> 4) & 15u); // "imposter" = agent 0 (component.py:262,289,...)
    int k = 0;
    for (int ci = 0; ci < o.ncomp; ci++) {
        const int comp = (int)T.comp[ci];
        switch (comp) {
        case SUSNET_F_ONEHOT_POS: // component.py:226-240
#pragma unroll
            for (int i = 0; i < A; i++)
                if ((e.alive >> i) & 1u) {
                    uint32_t w = st.xy(i);
                    if ((int)(w & 15u) < N) put_b(row, k + i * 2 * N + (int)(w & 15u), 1); // (cells of imported states are
                    if ((int)((w >> 4) & 15u) < N) put_b(row, k + i * 2 * N + N + (int)((w >> 4) & 15u), 1); //  not range-checked elsewhere)
                }
            break;
        case SUSNET_F_COORD_POS: // component.py:389-399
#pragma unroll
            for (int i = 0; i < A; i++) {
                uint32_t w = st.xy(i);
                put_b(row, k + 2 * i, w & 15u);
                put_b(row, k + 2 * i + 1, (w >> 4) & 15u);
            }
            break;
        case SUSNET_F_ALIVE_CREW: // component.py:411-421
#pragma unroll
            for (int i = 1; i < A; i++)
                if ((e.alive >> i) & 1u) put_b(row, k + i - 1, 1);
            break;
        case SUSNET_F_L1_CREW: // component.py:433-448
#pragma unroll
            for (int i = 1; i < A; i++) {
                uint32_t w = st.xy(i);
                int d = abs(ix - (int)(w & 15u)) + abs(iy - (int)((w >> 4) & 15u));
                put_b(row, k + i - 1, ((e.alive >> i) & 1u) ? d : -1);
            }
            break;
        case SUSNET_F_CLOSEST_CREW: { // component.py:460-478: argmin, first minimum, dead = N + N
            int best = 0, bestd = 1 << 20;
#pragma unroll
            for (int i = 1; i < A; i++) {
                uint32_t w = st.xy(i);
                int d = ((e.alive >> i) & 1u) ? abs(ix - (int)(w & 15u)) + abs(iy - (int)((w >> 4) & 15u)) : N + N;
                if (d < bestd) { bestd = d; best = i - 1; }
            }
            if (NC > 0) put_b(row, k + best, 1);
            break;
        }
        case SUSNET_F_WALLS3X3: // component.py:286-296: zero-padded grid[x, y] around agent 0
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    int gx = ix + a - 1, gy = iy + b - 1;
                    bool in = (unsigned)gx < (unsigned)N && (unsigned)gy < (unsigned)N;
                    uint32_t rw = T.grid[gx & 15];
                    if (in && ((rw >> (gy & 15)) & 1u)) put_b(row, k + a * 3 + b, 1);
                }
            break;
        case SUSNET_F_DIST_TO_IMP: { // component.py:255-273: alive non-0 agents packed left
            int p = 0;
#pragma unroll
            for (int i = 1; i < A; i++)
                if ((e.alive >> i) & 1u) {
                    uint32_t w = st.xy(i);
                    put_b(row, k + p, ix - (int)(w & 15u));
                    put_b(row, k + p + 1, iy - (int)((w >> 4) & 15u));
                    p += 2;
                }
            break;
        }
        case SUSNET_F_ROOM_LOC: { // component.py:8-17,308-329 (9x9 quadrants)
            uint32_t cnt = 0; // eight 4-bit counters (at most 15 agents per bucket)
#pragma unroll
            for (int i = 0; i < A; i++) {
                if (!((e.alive >> i) & 1u)) continue;
                uint32_t w = st.xy(i);
                int x = (int)(w & 15u), y = (int)((w >> 4) & 15u);
                int room = (x < 5) ? (y < 5 ? 0 : 1) : (y >= 5 ? 2 : 3);
                cnt += 1u << (4 * (((i == 0) ? 0 : 4) + room));
            }
            for (int r = 0; r < 8; r++) put_b(row, k + r, (cnt >> (4 * r)) & 15u);
            break;
        }
        default: break;
        }
        k += flat_component_size(comp, A, N, NC);
    }
}

// planes: bit image [64][F] packed; set bit (row * F + f) with an LDS atomic (lanes share words)
template <class S, class Store>
__device__ __forceinline__ void fill_planes(const Consts &c, const Store &st, const Env &e, uint32_t *bits, int rowbit0, uint8_t *bytes) {
    const int A = S::A(c), J = S::J(c), N = c.N, NN = N * N;
#pragma unroll
    for (int i = 0; i < A; i++) // component.py:90-100: channel i, [x][y], only if alive
        if ((e.alive >> i) & 1u) {
            uint32_t w = st.xy(i);
            if ((int)(w & 15u) >= N || (int)((w >> 4) & 15u) >= N) continue; // never leave this row's bit range
            int g = rowbit0 + i * NN + (int)(w & 15u) * N + (int)((w >> 4) & 15u);
            atomicOr(&bits[g >> 5], 1u << (g & 31));
        }
#pragma unroll
    for (int j = 0; j < J; j++) { // component.py:116-127: channel A + int(done)
        uint32_t w = st.job(j);
        if ((int)(w & 15u) >= N || (int)((w >> 4) & 15u) >= N) continue;
        int g = rowbit0 + (A + (int)((e.jd >> j) & 1u)) * NN + (int)(w & 15u) * N + (int)((w >> 4) & 15u);
        atomicOr(&bits[g >> 5], 1u << (g & 31));
    }
    int k = 0; // model_ready.py:237-247: [alive, (tag_counts), job_status]
#pragma unroll
    for (int i = 0; i < A; i++) put_b(bytes, k++, (e.alive >> i) & 1u);
    if (S::tagging(c))
        for (int i = 0; i < A; i++) put_b(bytes, k++, st.cnt(i));
#pragma unroll
    for (int j = 0; j < J; j++) put_b(bytes, k++, (e.jd >> j) & 1u);
}

// ---- cooperative wave store of a packed image of `total` elements -------------------------------------
// BYTES image, uint8 output: straight copy
__device__ __forceinline__ void copy_bytes(const uint32_t *img, int total, uint8_t *dst, int lane) {
    if (__builtin_expect((reinterpret_cast<uintptr_t>(dst) & 15u) == 0, 1)) {
        const int vec = total & ~15;
        for (int g = lane * 16; g < vec; g += kWave * 16)
            *reinterpret_cast<uint4 *>(dst + g) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(img) + g);
        for (int g = vec + lane; g < total; g += kWave) dst[g] = reinterpret_cast<const uint8_t *>(img)[g];
    } else {
        for (int g = lane; g < total; g += kWave) dst[g] = reinterpret_cast<const uint8_t *>(img)[g];
    }
}

// BYTES or BITS image, float32 output: 4 elements per lane per pass
template <bool BITS>
__device__ __forceinline__ void expand_f32(const uint32_t *img, int total, float *dst, bool is_signed, int lane) {
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
#pragma unroll 4
    for (int g = lane * 4; g < total; g += kWave * 4) {
        float v[4];
        if (BITS) {
            uint32_t q = (img[g >> 5] >> (g & 31)) & 15u;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (float)((q >> k) & 1u);
        } else {
            uint32_t q = img[g >> 2];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t by = (q >> (8 * k)) & 0xffu;
                v[k] = is_signed ? (float)(int)(int8_t)by : (float)by;
            }
        }
        if (vec_ok && g + 3 < total) *reinterpret_cast<float4 *>(dst + g) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int k = 0; k < 4 && g + k < total; k++) dst[g + k] = v[k];
    }
}

// PerspectiveFeaturizer (model_ready.py:175-216): agent i sees the agent channels in the order i, 0, .., i-1, i+1, .., A-1
__device__ __forceinline__ int persp_source(int i, int pos) { return pos == 0 ? i : (pos <= i ? pos - 1 : pos); }

// planes BIT image [rows][C][NN] -> [rows][A][C][NN] with every agent's channel order; float32 or uint8 output, 4 elements
// per lane per pass (their positions decoded once, then walked)
template <class OUT_T>
__device__ __forceinline__ void expand_persp_planes(const uint32_t *img, int rows, int A, int C, int NN, OUT_T *dst, int lane) {
    const int per_row = A * C * NN, total = rows * per_row, Fi = C * NN;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (4 * sizeof(OUT_T) - 1)) == 0;
    for (int g = lane * 4; g < total; g += kWave * 4) {
        int row = g / per_row, r = g - row * per_row;
        int i = r / Fi, r2 = r - i * Fi;
        int ch = r2 / NN, cell = r2 - ch * NN;
        OUT_T v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int src = ch < A ? persp_source(i, ch) : ch;
            const int bit = row * Fi + src * NN + cell;
            v[k] = (OUT_T)((img[bit >> 5] >> (bit & 31)) & 1u);
            if (++cell == NN) { // next channel / agent / row
                cell = 0;
                if (++ch == C) { ch = 0; if (++i == A) { i = 0; row++; } }
            }
        }
        if (vec_ok && g + 3 < total) {
            if constexpr (sizeof(OUT_T) == 4) *reinterpret_cast<float4 *>(dst + g) = make_float4(v[0], v[1], v[2], v[3]);
            else *reinterpret_cast<uint32_t *>(dst + g) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        } else {
            for (int k = 0; k < 4 && g + k < total; k++) dst[g + k] = v[k];
        }
    }
}
// non-spatial BYTE image [rows][nb * A + J] -> [rows][A][nb * A + J]: the per-agent blocks (alive, tag counts) in agent i's order
template <class OUT_T>
__device__ __forceinline__ void expand_persp_bytes(const uint32_t *img, int rows, int A, int F2i, int nb, OUT_T *dst, int lane) {
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(img);
    const int per_row = A * F2i, total = rows * per_row;
    for (int g = lane; g < total; g += kWave) {
        const int row = g / per_row, r = g - row * per_row;
        const int i = r / F2i, k = r - i * F2i;
        int src = k;
        if (k < nb * A) {
            const int blk = k / A, pos = k - blk * A;
            src = blk * A + persp_source(i, pos);
        }
        dst[g] = (OUT_T)bytes[row * F2i + src];
    }
}

// BITS image, uint8 output
__device__ __forceinline__ void expand_bits_u8(const uint32_t *img, int total, uint8_t *dst, int lane) {
    for (int g = lane * 4; g < total; g += kWave * 4) {
        uint32_t q = (img[g >> 5] >> (g & 31)) & 15u;
        uint32_t packed = (q & 1u) | ((q & 2u) << 7) | ((q & 4u) << 14) | ((q & 8u) << 21);
        if (g + 3 < total && (reinterpret_cast<uintptr_t>(dst + g) & 3u) == 0) *reinterpret_cast<uint32_t *>(dst + g) = packed;
        else
            for (int k = 0; k < 4 && g + k < total; k++) dst[g + k] = (uint8_t)((q >> k) & 1u);
    }
}

// Build this wave's image(s) and stream them. `active` = this lane owns a real env; `tick` selects the
// trajectory slice in rollouts (0 otherwise).
template <class S, class Store>
__device__ __forceinline__ void write_obs(const Consts &c, const ObsArgs &o, const Tables &T, const Store &st, int tid, const Env &e,
                                          bool active, int64_t b0, int nrows, int64_t tick) {
    if (__builtin_expect(o.mode == SUSNET_OBS_NONE, 0)) return;
    uint32_t *seg1 = T.stage;
    uint32_t *seg2 = T.stage + o.words1;
    const bool persp = o.mode == SUSNET_OBS_PERSP;
    const bool planes = o.mode == SUSNET_OBS_PLANES || persp;
    if (__builtin_expect(o.mode != SUSNET_OBS_RAW, 0)) { // RAW rows are fully overwritten; the others start from zeros
        for (int w = tid; w < o.words1 + o.words2; w += kWave) seg1[w] = 0u;
        wave_lds_fence();
    }
    if (active) {
        uint8_t *row = reinterpret_cast<uint8_t *>(seg1) + tid * o.Fi;
        if (__builtin_expect(o.mode == SUSNET_OBS_RAW, 1)) fill_raw<S>(c, st, e, row);
        else if (o.mode == SUSNET_OBS_FLAT) fill_flat<S>(c, o, T, st, e, row);
        else fill_planes<S>(c, st, e, seg1, tid * o.Fi, reinterpret_cast<uint8_t *>(seg2) + tid * o.F2i);
    }
    wave_lds_fence();
    if (__builtin_expect(persp, 0)) { // every agent's rotated copy of the planes / of the per-agent non-spatial blocks
        const int A = S::A(c), NN = c.N * c.N, nb = S::tagging(c) ? 2 : 1;
        if (o.dtype == SUSNET_F32) {
            expand_persp_planes(seg1, nrows, A, A + 2, NN, reinterpret_cast<float *>(o.out) + tick * o.tick_stride + b0 * o.F, tid);
            if (o.out2) expand_persp_bytes(seg2, nrows, A, o.F2i, nb, reinterpret_cast<float *>(o.out2) + tick * o.tick_stride2 + b0 * o.F2, tid);
        } else {
            expand_persp_planes(seg1, nrows, A, A + 2, NN, reinterpret_cast<uint8_t *>(o.out) + tick * o.tick_stride + b0 * o.F, tid);
            if (o.out2) expand_persp_bytes(seg2, nrows, A, o.F2i, nb, reinterpret_cast<uint8_t *>(o.out2) + tick * o.tick_stride2 + b0 * o.F2, tid);
        }
        wave_lds_fence();
        return;
    }
    const int total1 = nrows * o.F;
    if (__builtin_expect(o.dtype == SUSNET_F32, 0)) {
        float *d1 = reinterpret_cast<float *>(o.out) + tick * o.tick_stride + b0 * o.F;
        if (planes) {
            expand_f32<true>(seg1, total1, d1, false, tid);
            if (o.out2) expand_f32<false>(seg2, nrows * o.F2, reinterpret_cast<float *>(o.out2) + tick * o.tick_stride2 + b0 * o.F2, false, tid);
        } else {
            expand_f32<false>(seg1, total1, d1, o.mode == SUSNET_OBS_FLAT, tid);
        }
    } else {
        uint8_t *d1 = reinterpret_cast<uint8_t *>(o.out) + tick * o.tick_stride + b0 * o.F;
        if (planes) {
            expand_bits_u8(seg1, total1, d1, tid);
            if (o.out2) copy_bytes(seg2, nrows * o.F2, reinterpret_cast<uint8_t *>(o.out2) + tick * o.tick_stride2 + b0 * o.F2, tid);
        } else {
            copy_bytes(seg1, total1, d1, tid);
        }
    }
    wave_lds_fence(); // the image is rebuilt next tick
}

// compile-time flavour of write_obs for the raw uint8 observation (the rollout's populate()-shaped record)
template <class S, class Store>
__device__ __forceinline__ void write_obs_raw8(const Consts &c, const ObsArgs &o, const Tables &T, const Store &st, int tid, const Env &e,
                                               bool active, int64_t b0, int nrows, int64_t tick) {
    uint32_t *img = T.stage;
    if (active) fill_raw<S>(c, st, e, reinterpret_cast<uint8_t *>(img) + tid * o.F);
    wave_lds_fence();
    copy_bytes(img, nrows * o.F, reinterpret_cast<uint8_t *>(o.out) + tick * o.tick_stride + b0 * o.F, tid);
    wave_lds_fence();
}

} // namespace susnet
