// susnet_qnet.h -- the policy loop's Q-network forward pass as ONE kernel on the matrix cores (BASELINE config 5).
//
// Reference behaviour (paths relative to the reference repo root):
//   MLP.forward                       src/models/dqn.py:72-88   Linear + PReLU stack on the flat features, spatial input ignored
//   make_mlp                          src/models/dqn.py:322-329 (last activation dropped; nn.PReLU(): ONE slope per layer)
//   FlatFeaturizer features           src/features/model_ready.py:356-367, component.py:221-247, 406-482 (susnet_flat.h)
//   layer dims [F, 256, 128, 64, 16, n_actions]   notebooks/experiment_1v1.ipynb cell 1
//   caller: run_game / the acting loop            src/visualize.py:547-562, src/train.py:355-381
//
// What the stock path does per tick on 65 536 environments: write the [B][88] float observation, five hipBLASLt f32 GEMMs and four
// PReLU passes over [B][256 .. 16] activations that each go out to HBM and come back (158 of the tick's 172 us).  Here a wave owns
// 64 environments (two column tiles of 32) from the state words to the Q row:
//   * layer 1 is not a GEMM at all: a FlatFeaturizer row of these layouts is a 0/1 vector with at most 2A + A ones, so
//     h1 = b1 + sum of the <= 9 columns of W1 the set bits select -- gathered from an LDS image of W1^T (one padded row per feature,
//     a bias row and a zero row for absent bits), 16 bytes per lane and read, products exact;
//   * layers 2..5 run on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise a k-ordered fmaf chain) in TRANSPOSED form,
//     H_out^T[n][m] = W[n][k] . H_in^T[k][m]: the weights are the A operand, the activations the B operand, and the C/D layout of one
//     layer (lane = environment column m, register t of half h = row 8 (t / 4) + 4 h + t % 4 of the 32-row block) IS the B operand
//     layout of the next one when the k index of MFMA step t is taken in that same order -- the weights are packed on the host in that
//     order (susnet_qnet_pack), so activations never move between registers, LDS or memory from layer 1 to the Q row;
//   * layer 1 is produced 32 features at a time and consumed at once as a k block of layer 2 (the [256] activation never exists
//     as a whole); biases are the accumulators' initial values; PReLU is three VALU instructions per register in the MFMA shadow.
// Weights of layers 2..5 (170 KB) stream from L2 as 1 KiB coalesced blocks, each feeding 32 MFMAs of the wave.
// Cost per wave: 1 360 MFMAs x 64 cycles = 87 K cycles = 36 us at 2.4 GHz for 65 536 environments (one wave per SIMD) against
// 158 us of library GEMM + activation kernels; the f32 matrix peak (157 TFLOP/s) bounds it, layer 1 (35 % of the MACs) costs no MFMA.
#pragma once

#include "susnet_device.h"
#include "susnet_flat.h"

namespace susnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Shape of the compiled-in network family: five Linear layers, hidden widths PADDED to these (zero weights / biases: a padded unit
// outputs PReLU(0) = 0 and feeds nothing), so any [F, <=256, <=128, <=64, <=32, <=32] stack of the reference's MLP class runs on it.
template <class ROW>
struct QNet {
    static constexpr int F = ROW::F, H1 = 256, H2 = 128, H3 = 64, H4 = 32, NO = 32;
    static constexpr int kOnes = 2 * ROW::A + (ROW::F > ROW::kOneHot ? ROW::A : 0); // most set bits of a row (positions, alive flags, closest)
    static constexpr int kRows = F + 2;      // + the bias row (index F) + an all-zero row (index F + 1) for absent bits
    static constexpr int kRowStride = H1 + 4; // floats: consecutive feature rows start 4 banks apart
    static constexpr int kW1 = kRows * kRowStride;
    // packed image, in floats (susnet_qnet_pack writes it, the kernel reads it):
    static constexpr int oW1 = 0, oW2 = oW1 + kW1, oB2 = oW2 + H1 * H2, oW3 = oB2 + H2, oB3 = oW3 + H2 * H3, oW4 = oB3 + H3, oB4 = oW4 + H3 * H4,
                         oW5 = oB4 + H4, oB5 = oW5 + H4 * NO, oSlope = oB5 + NO, kPacked = oSlope + 4;
    static constexpr int kLdsBytes = kW1 * 4;
    static constexpr int kThreads = 256, kEnvsPerWave = 64, kEnvsPerBlock = 4 * kEnvsPerWave;
};

// positions of a row's set bits, ascending, padded with `fill`
template <class ROW, int MAXONES>
__device__ __forceinline__ void flat_row_ones(const ROW &row, uint32_t (&idx)[MAXONES], uint32_t fill) {
    uint32_t w[ROW::MW];
#pragma unroll
    for (int i = 0; i < ROW::MW; i++) w[i] = row.m[i];
#pragma unroll
    for (int q = 0; q < MAXONES; q++) {
        uint32_t f = fill;
        bool found = false;
#pragma unroll
        for (int i = 0; i < ROW::MW; i++) {
            const bool take = !found && w[i] != 0u;
            f = take ? (uint32_t)(32 * i) + (uint32_t)__builtin_ctz(w[i] | 0x80000000u) : f;
            w[i] = take ? (w[i] & (w[i] - 1u)) : w[i];
            found = found || take;
        }
        idx[q] = f;
    }
}

__device__ __forceinline__ void prelu16(f32x16 &v, float slope) { // torch.prelu: x > 0 ? x : slope * x
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = v[i] > 0.0f ? v[i] : slope * v[i];
}

// one 32 x 32 weight block (lane: output row n = lane % 32; its 16 k values in MFMA-step order) times T activation blocks
template <int T>
__device__ __forceinline__ void mfma_block(const f32x4 (&w)[4], const f32x16 (&in)[T], f32x16 (&acc)[T]) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int t = 0; t < T; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q][r], in[t][4 * q + r], acc[t], 0, 0, 0);
}
__device__ __forceinline__ void load_block(f32x4 (&w)[4], const f32x4 *blk, int lane) {
#pragma unroll
    for (int q = 0; q < 4; q++) w[q] = blk[q * 64 + lane];
}
// accumulators start as the layer's bias: register 4 q + r of half h is row 8 q + 4 h + r of the block
template <int T>
__device__ __forceinline__ void bias_block(const float *bias, int nb, int h, f32x16 (&acc)[T]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + nb * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int t = 0; t < T; t++) acc[t][4 * q + r] = b[r];
    }
}
// a register-resident dense layer: in[KB][T] (activated) -> out[NB][T] (bias + sum, not yet activated); weight blocks in [kb][nb] order
template <int KB, int NB, int T>
__device__ __forceinline__ void dense(const float *wp, const float *bias, int lane, const f32x16 (&in)[KB][T], f32x16 (&out)[NB][T]) {
    const int h = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) bias_block<T>(bias, nb, h, out[nb]);
    const f32x4 *blk = reinterpret_cast<const f32x4 *>(wp);
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            f32x4 w[4];
            load_block(w, blk + (kb * NB + nb) * 256, lane);
            mfma_block<T>(w, in[kb], out[nb]);
        }
}

// Q rows of the handle's CURRENT environments: q_out [B][n_out] float32.  256 threads = 4 waves (one per SIMD) share the LDS
// image of layer 1; wave w of block g owns environments (4 g + w) * 64 ..; lane = (column m = lane % 32, half h = lane / 32) of each
// of its two 32-environment tiles.
template <class ROW>
__global__ __launch_bounds__(256) void k_qnet(Consts c, State s, const float *pk, float *q_out, int n_out) {
    using Q = QNet<ROW>;
    constexpr int T = 2;
    extern __shared__ float w1[];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(pk + Q::oW1);
        f32x4 *dst = reinterpret_cast<f32x4 *>(w1);
        for (int i = threadIdx.x; i < Q::kW1 / 4; i += Q::kThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * Q::kEnvsPerWave;
    if (b0 >= c.B) return; // (after the only barrier)
    const float slope1 = pk[Q::oSlope + 0], slope2 = pk[Q::oSlope + 1], slope3 = pk[Q::oSlope + 2], slope4 = pk[Q::oSlope + 3];

    // the <= kOnes feature rows of each tile's environment, as LDS addresses of this half's 16-byte column slice
    const float *rowp[T][Q::kOnes];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int64_t b = b0 + 32 * t + m;
        ROW row;
        row.clear();
        if (b < c.B) {
            uint32_t fx[ROW::A], fy[ROW::A], fal[ROW::A];
#pragma unroll
            for (int i = 0; i < ROW::A; i++) {
                const uint32_t w = s.agent[(size_t)i * c.Bp + b];
                fx[i] = w & 15u;
                fy[i] = (w >> 4) & 15u;
                fal[i] = (w >> 8) & 1u;
            }
            row.build(fx, fy, fal);
        }
        uint32_t idx[Q::kOnes];
        flat_row_ones<ROW, Q::kOnes>(row, idx, (uint32_t)(Q::F + 1));
#pragma unroll
        for (int q = 0; q < Q::kOnes; q++) rowp[t][q] = w1 + idx[q] * Q::kRowStride + 4 * h;
    }
    const float *biasp = w1 + Q::F * Q::kRowStride + 4 * h;

    // layers 1 + 2: h1 block kb (32 features) by gather, then straight into layer 2 as its k block kb
    f32x16 a2[Q::H2 / 32][T];
#pragma unroll
    for (int nb = 0; nb < Q::H2 / 32; nb++) bias_block<T>(pk + Q::oB2, nb, h, a2[nb]);
    const f32x4 *w2 = reinterpret_cast<const f32x4 *>(pk + Q::oW2);
    f32x4 wa[4], wb[4];
    load_block(wa, w2, lane);
#pragma unroll 1
    for (int kb = 0; kb < Q::H1 / 32; kb++) {
        f32x16 hb[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int col = kb * 32 + 8 * j; // (+ 4 h inside the row pointers)
                f32x4 v = *reinterpret_cast<const f32x4 *>(biasp + col);
#pragma unroll
                for (int q = 0; q < Q::kOnes; q++) v += *reinterpret_cast<const f32x4 *>(rowp[t][q] + col);
#pragma unroll
                for (int r = 0; r < 4; r++) hb[t][4 * j + r] = v[r];
            }
            prelu16(hb[t], slope1);
        }
        const f32x4 *blk = w2 + (size_t)kb * (Q::H2 / 32) * 256;
        const f32x4 *nxt = w2 + (size_t)(kb + 1 < Q::H1 / 32 ? kb + 1 : kb) * (Q::H2 / 32) * 256;
        static_assert(Q::H2 / 32 == 4, "the weight double buffer below is written for four row blocks");
        load_block(wb, blk + 1 * 256, lane);
        mfma_block<T>(wa, hb, a2[0]);
        load_block(wa, blk + 2 * 256, lane);
        mfma_block<T>(wb, hb, a2[1]);
        load_block(wb, blk + 3 * 256, lane);
        mfma_block<T>(wa, hb, a2[2]);
        load_block(wa, nxt, lane);
        mfma_block<T>(wb, hb, a2[3]);
    }
#pragma unroll
    for (int nb = 0; nb < Q::H2 / 32; nb++)
#pragma unroll
        for (int t = 0; t < T; t++) prelu16(a2[nb][t], slope2);

    f32x16 a3[Q::H3 / 32][T];
    dense<Q::H2 / 32, Q::H3 / 32, T>(pk + Q::oW3, pk + Q::oB3, lane, a2, a3);
#pragma unroll
    for (int nb = 0; nb < Q::H3 / 32; nb++)
#pragma unroll
        for (int t = 0; t < T; t++) prelu16(a3[nb][t], slope3);

    f32x16 a4[Q::H4 / 32][T];
    dense<Q::H3 / 32, Q::H4 / 32, T>(pk + Q::oW4, pk + Q::oB4, lane, a3, a4);
#pragma unroll
    for (int t = 0; t < T; t++) prelu16(a4[0][t], slope4);

    f32x16 a5[Q::NO / 32][T];
    dense<Q::H4 / 32, Q::NO / 32, T>(pk + Q::oW5, pk + Q::oB5, lane, a4, a5); // dqn.py:328: no activation after the last Linear

#pragma unroll
    for (int t = 0; t < T; t++) {
        const int64_t b = b0 + 32 * t + m;
        if (b < c.B) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                if (n < n_out) q_out[b * n_out + n] = a5[0][t][i];
            }
        }
    }
}

} // namespace susnet
