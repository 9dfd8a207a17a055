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
#include "susnet_kernels.h" // static_for

namespace susnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Shape of the compiled-in network family: five Linear layers, hidden widths PADDED to these (zero weights / biases: a padded unit
// outputs PReLU(0) = 0 and feeds nothing), so any [F, <=256, <=128, <=64, <=32, <=32] stack of the reference's MLP class runs on it.
template <class ROW>
struct QNet {
    static constexpr int F = ROW::F, H1 = 256, H2 = 128, H3 = 64, H4 = 32, NO = 32;
    // LDS image of layer 1, one row of H1 floats per entry: rows 0 .. kOneHot - 1 = the W1 columns of the position one-hots; row kZero =
    // all zeros (a dead agent's positions, component.py:226-240); rows kTail + v = b1 + the W1 columns of the set bits of v, v = the
    // row's bits BEHIND the one-hots (alive flags, closest crew: 2^kTailBits combinations, summed on the host; no such bits: just b1).
    // h1 = sum of 2 A position rows + ONE tail row: 2 A + 1 LDS reads per 16-byte slice instead of one per set bit + bias.
    // (the coordinate layout, susnet_flat.h CoordRow: the "one-hot" rows are k x the coordinate's W1 column for every value k, no tail bits)
    static constexpr int kOneHot = ROW::kOneHot, kTailBits = ROW::kTailBits, kZero = kOneHot, kTail = kOneHot + 1;
    static constexpr int kGather = 2 * ROW::A + 1;
    static constexpr int kRows = kTail + (1 << kTailBits);
    static constexpr int kRowStride = H1 + 4; // floats: consecutive rows start 4 banks apart
    // (padded to a multiple of 4 KiB: the both-teams tick swaps this part of the image by 1 KiB global -> LDS transfers, the same number for
    // each of the four waves: qnet_swap_issue)
    static constexpr int kW1 = (kRows * kRowStride + 1023) / 1024 * 1024;
    static constexpr int kTailFloats = H2 + H3 + H4 + NO; // the biases of layers 2..5 behind it
    // packed image, in floats (susnet_qnet_pack writes it, the kernel reads it).  First the part every workgroup copies to LDS
    // (layer 1 transposed + the biases of layers 2..5), then ALL 32 x 32 weight blocks of layers 2..5 as one stream in the order the
    // kernel consumes them, then the four PReLU slopes.
    static constexpr int oW1 = 0, oB2 = oW1 + kW1, oB3 = oB2 + H2, oB4 = oB3 + H3, oB5 = oB4 + H4, kLdsFloats = oB5 + NO;
    static constexpr int oW2 = kLdsFloats, oW3 = oW2 + H1 * H2, oW4 = oW3 + H2 * H3, oW5 = oW4 + H3 * H4, oSlope = oW5 + H4 * NO, kPacked = oSlope + 4;
    static constexpr int kBlocks = (H1 * H2 + H2 * H3 + H3 * H4 + H4 * NO) / 1024; // of the stream
    static constexpr int kLdsBytes = kLdsFloats * 4;
    static constexpr int kLdsBytesTwo = kLdsBytes + kTailFloats * 4; // both teams: the second network's biases stay resident behind the image
    static constexpr int kThreads = 256, kEnvsPerWave = 64, kEnvsPerBlock = 4 * kEnvsPerWave;
    static_assert(kTailBits >= 0 && kTailBits <= 4 && kLdsFloats % 4 == 0 && H2 / 32 == 4 && (H1 / 32) % 2 == 0, "16-byte copies; the pipelined stage is written for this shape");
};

// torch.prelu: x > 0 ? x : slope * x.  UNIT (every slope of the network in [0, 1], the usual case: nn.PReLU() starts at 0.25): the same
// value as max(x, slope * x) -- two instructions per element (the multiply packs) instead of three.
template <bool UNIT>
__device__ __forceinline__ float prelu(float x, float slope) {
    if constexpr (UNIT) return __builtin_fmaxf(x, slope * x);
    else return x > 0.0f ? x : slope * x;
}
// four elements: the multiply as ONE vector expression (two v_pk_mul_f32 instead of four v_mul_f32; the same products)
template <bool UNIT>
__device__ __forceinline__ f32x4 prelu4(f32x4 x, float slope) {
    const f32x4 m = x * slope;
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = UNIT ? __builtin_fmaxf(x[i], m[i]) : (x[i] > 0.0f ? x[i] : m[i]);
    return r;
}
template <bool UNIT>
__device__ __forceinline__ void prelu16(f32x16 &v, float slope) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        const f32x4 r = prelu4<UNIT>(x, slope);
#pragma unroll
        for (int i = 0; i < 4; i++) v[4 * q + i] = r[i];
    }
}

// HALF of a 32 x 32 weight block (float4 2 HALF and 2 HALF + 1 of the lane's 16 k values, MFMA steps 8 HALF .. 8 HALF + 7) times
// T activation blocks: 8 T MFMAs on T independent accumulators
template <int HALF, int T>
__device__ __forceinline__ void mfma_half(const f32x4 (&w)[4], const f32x16 (&in)[T], f32x16 (&acc)[T]) {
#pragma unroll
    for (int q = 2 * HALF; q < 2 * HALF + 2; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int t = 0; t < T; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q][r], in[t][4 * q + r], acc[t], 0, 0, 0);
}
// accumulators start as the layer's bias (LDS): register 4 q + r of half h is row 8 q + 4 h + r of the block
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
// The weight stream: block b of the packed image lives in register buffer b % 4 and is requested two blocks (64 T MFMAs >= 4 000
// cycles) before its first use, so an L2 round trip never stalls the matrix core; sched_barrier keeps the compiler from sinking a
// request down to its use.
struct WeightStream {
    __amdgpu_buffer_rsrc_t r; // the stream as a raw buffer: the block offset rides in the scalar offset, the float4 index in the immediate
    uint32_t lane16;          // 16 * lane
    int lane;
    template <int BUF>
    __device__ __forceinline__ void request(f32x4 (&w)[4][4], int block, int last) const {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t so = (uint32_t)(block < last ? block : last) * 4096u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane16 + 1024u * q, so, 0);
            w[BUF][q] = __builtin_bit_cast(f32x4, v);
        }
    }
};
// a register-resident dense layer on stream blocks B0 .. B0 + KB NB - 1 ([kb][nb] order): in[KB][T] (activated) -> out[NB][T] (bias + sum)
// mid(i): called in the middle of block i's matrix instructions (the both-teams tick issues its image transfers there, a few per block)
struct NoMidBlock {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int B0, int KB, int NB, int T, int LAST, class MIDB = NoMidBlock>
__device__ __forceinline__ void dense(const WeightStream &ws, f32x4 (&w)[4][4], const float *bias, const f32x16 (&in)[KB][T], f32x16 (&out)[NB][T], MIDB &&mid = MIDB()) {
    const int h = ws.lane >> 5;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) bias_block<T>(bias, nb, h, out[nb]);
    static_for<0, KB * NB>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, kb = i / NB, nb = i % NB, b = B0 + i;
        if constexpr (b % 2 == 0) { // (two requests at a time: every MFMA -> load transition costs issue cycles)
            ws.template request<(b + 2) % 4>(w, b + 2, LAST);
            ws.template request<(b + 3) % 4>(w, b + 3, LAST);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_half<0, T>(w[b % 4], in[kb], out[nb]);
        __builtin_amdgcn_sched_barrier(0);
        mid(i);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half<1, T>(w[b % 4], in[kb], out[nb]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Q rows of the handle's CURRENT environments: q_out [B][n_out] float32.  256 threads = 4 waves (one per SIMD) share the LDS
// image of layer 1; wave w of block g owns environments (4 g + w) * 64 ..; lane = (column m = lane % 32, half h = lane / 32) of each
// of its two 32-environment tiles.
// Returns the greedy action of the environment lane `lane` of the wave owns (b0 + lane): argmax of its Q row, first maximum.
// btail: the biases of layers 2..5 in LDS (b2 | b3 | b4 | b5); after: synchronise() is called once the layer-1 image has been read for
// the last time, transfer(i, n) in the middle of block i of layer 3's n weight blocks (the both-teams tick fetches the other network's
// image there, a share per block: the rest of the pass covers the transfer)
struct NoAfter {
    __device__ __forceinline__ void synchronise() const {}
    __device__ __forceinline__ void transfer(int, int) const {}
};
template <class ROW, bool UNIT, class AFTER = NoAfter>
__device__ __forceinline__ uint32_t qnet_wave(const Consts &c, const uint32_t (&aw)[2][ROW::A], const float *pk, const float *w1, const float *btail, float *q_out,
                                              int n_out, int64_t b0, int lane, AFTER &&after = AFTER()) {
    using Q = QNet<ROW>;
    constexpr int T = 2, R = Q::kGather;
    const int h = lane >> 5;
    const float slope1 = pk[Q::oSlope + 0], slope2 = pk[Q::oSlope + 1], slope3 = pk[Q::oSlope + 2], slope4 = pk[Q::oSlope + 3];
    constexpr int kLast = Q::kBlocks - 1;
    const WeightStream ws = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pk + Q::oW2), 0, Q::kBlocks * 4096, 0x00020000), (uint32_t)lane * 16u, lane};
    f32x4 w[4][4]; // block b of the stream sits in w[b % 4]; blocks b + 2, b + 3 are requested while b, b + 1 are multiplied
    ws.template request<0>(w, 0, kLast);
    ws.template request<1>(w, 1, kLast);

    // the R rows of each tile's environment, as LDS addresses of this half's 16-byte column slice
    const float *rowp[T][R];
#pragma unroll
    for (int t = 0; t < T; t++) {
        uint32_t fx[ROW::A], fy[ROW::A], fal[ROW::A];
#pragma unroll
        for (int i = 0; i < ROW::A; i++) {
            fx[i] = aw[t][i] & 15u;
            fy[i] = (aw[t][i] >> 4) & 15u;
            fal[i] = (aw[t][i] >> 8) & 1u; // (an environment past the batch: all zero = every agent dead, nothing stored)
        }
        ROW row; // the feature row itself (susnet_flat.h): only its bits behind the one-hots are needed, but they come from build()
        row.build(fx, fy, fal);
        uint32_t tail = 0;
        if constexpr (Q::kTailBits > 0) {
            static_assert(Q::kOneHot / 32 == (Q::kOneHot + Q::kTailBits - 1) / 32, "the tail bits sit in one mask word");
            tail = (row.m[Q::kOneHot / 32] >> (Q::kOneHot & 31)) & ((1u << Q::kTailBits) - 1u);
        }
        rowp[t][0] = w1 + (Q::kTail + tail) * Q::kRowStride + 4 * h;
#pragma unroll
        for (int i = 0; i < ROW::A; i++) { // component.py:226-240: [onehot_N(x) | onehot_N(y)] per agent, zeros if dead
            const bool there = fal[i] != 0u || !ROW::kDeadZero; // (coordinates are not zeroed for a dead agent, component.py:389-399)
            rowp[t][1 + 2 * i] = w1 + (there ? (uint32_t)(i * 2 * ROW::N) + fx[i] : (uint32_t)Q::kZero) * Q::kRowStride + 4 * h;
            rowp[t][2 + 2 * i] = w1 + (there ? (uint32_t)(i * 2 * ROW::N + ROW::N) + fy[i] : (uint32_t)Q::kZero) * Q::kRowStride + 4 * h;
        }
    }

    // Layers 1 + 2.  h1 block kb (32 features x 64 environments: per tile and 8-column slice R 16-byte LDS reads, their sum, PReLU) is
    // gathered and consumed at once as k block kb of layer 2.  v_mfma_f32_32x32x2_f32 holds the SIMD's vector issue for all of its 64
    // cycles (tools/mfma_f32_fillers.hip: every VALU / LDS instruction placed beside it costs its full issue time, the first one after
    // an MFMA 12 cycles more), so nothing hides in its shadow: a stage is a gather CLUSTER followed by 128 back-to-back MFMAs, never an
    // interleaving, and the k loop is unrolled so that every LDS offset is an immediate.
    f32x16 a2[Q::H2 / 32][T];
#pragma unroll
    for (int nb = 0; nb < Q::H2 / 32; nb++) bias_block<T>(btail, nb, h, a2[nb]);
    static_for<0, Q::H1 / 32>([&](auto kc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kc)::value;
        ws.template request<2>(w, kb * 4 + 2, kLast);
        ws.template request<3>(w, kb * 4 + 3, kLast);
        f32x16 hb[T];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(rowp[t][0] + kb * 32 + 8 * j);
#pragma unroll
                for (int q = 1; q < R; q++) v += *reinterpret_cast<const f32x4 *>(rowp[t][q] + kb * 32 + 8 * j);
                const f32x4 pv = prelu4<UNIT>(v, slope1);
#pragma unroll
                for (int r = 0; r < 4; r++) hb[t][4 * j + r] = pv[r];
            }
        __builtin_amdgcn_sched_barrier(0);
        mfma_half<0, T>(w[0], hb, a2[0]);
        mfma_half<1, T>(w[0], hb, a2[0]);
        mfma_half<0, T>(w[1], hb, a2[1]);
        mfma_half<1, T>(w[1], hb, a2[1]);
        __builtin_amdgcn_sched_barrier(0);
        ws.template request<0>(w, kb * 4 + 4, kLast);
        ws.template request<1>(w, kb * 4 + 5, kLast);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half<0, T>(w[2], hb, a2[2]);
        mfma_half<1, T>(w[2], hb, a2[2]);
        mfma_half<0, T>(w[3], hb, a2[3]);
        mfma_half<1, T>(w[3], hb, a2[3]);
        __builtin_amdgcn_sched_barrier(0);
    });
    after.synchronise();
#pragma unroll
    for (int nb = 0; nb < Q::H2 / 32; nb++)
#pragma unroll
        for (int t = 0; t < T; t++) prelu16<UNIT>(a2[nb][t], slope2);

    constexpr int kB3 = Q::H1 / 32 * 4, kB4 = kB3 + (Q::H2 / 32) * (Q::H3 / 32), kB5 = kB4 + (Q::H3 / 32) * (Q::H4 / 32);
    f32x16 a3[Q::H3 / 32][T];
    dense<kB3, Q::H2 / 32, Q::H3 / 32, T, kLast>(ws, w, btail + Q::H2, a2, a3, [&](int i) __attribute__((always_inline)) { after.transfer(i, (Q::H2 / 32) * (Q::H3 / 32)); });
#pragma unroll
    for (int nb = 0; nb < Q::H3 / 32; nb++)
#pragma unroll
        for (int t = 0; t < T; t++) prelu16<UNIT>(a3[nb][t], slope3);

    f32x16 a4[Q::H4 / 32][T];
    dense<kB4, Q::H3 / 32, Q::H4 / 32, T, kLast>(ws, w, btail + Q::H2 + Q::H3, a3, a4);
#pragma unroll
    for (int t = 0; t < T; t++) prelu16<UNIT>(a4[0][t], slope4);

    f32x16 a5[Q::NO / 32][T];
    dense<kB5, Q::H4 / 32, Q::NO / 32, T, kLast>(ws, w, btail + Q::H2 + Q::H3 + Q::H4, a4, a5); // dqn.py:328: no activation after the last Linear

    // The Q rows go out through a raw buffer over this wave's rows: a row past the batch or an entry past n_out gets an offset the
    // hardware's range check drops -- NO divergent branch anywhere in this function.  (Every vector register is spoken for here and the
    // compiler parks values in accumulator registers around the matrix section; a parking move it places inside a divergent region runs
    // under that region's narrowed EXEC and loses the value in the other lanes -- seen with the row index across `if (b < B)`.)
    uint32_t best[T];
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e)); // (the epilogue's lane-derived values are recomputed here, not carried across the matrix section)
    const int me = lane_e & 31, he = lane_e >> 5;
    const int64_t rows_left = c.B - b0;
    const uint32_t n_rows = rows_left < 64 ? (uint32_t)rows_left : 64u;
    if (q_out != nullptr) { // (wave-uniform)
        // (base and size forced into scalar registers: a descriptor the compiler takes for lane-dependent is served by a loop over lanes)
        float *qb = reinterpret_cast<float *>(uniform64(reinterpret_cast<uint64_t>(q_out + b0 * n_out)));
        const int qbytes = __builtin_amdgcn_readfirstlane((int)(n_rows * (uint32_t)n_out * 4u));
        const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(qb, 0, qbytes, 0x00020000);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int n = 8 * (i >> 2) + 4 * he + (i & 3);
                const uint32_t off = n < n_out ? ((uint32_t)(32 * t + me) * (uint32_t)n_out + (uint32_t)n) * 4u : 0x80000000u;
                const float qv = a5[0][t][i]; // (a copy: __builtin_bit_cast applied to the vector element itself reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, qv), qr, off, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        // argmax over the row: this lane's entries in ascending n (first maximum), then against the other half's (lane ^ 32)
        float hv = -__builtin_inff();
        uint32_t hn = 0xffffu;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int n = 8 * (i >> 2) + 4 * he + (i & 3);
            const float v = a5[0][t][i];
            if (n < n_out && (hn == 0xffffu || v > hv)) { hv = v; hn = (uint32_t)n; }
        }
        const float pv = __shfl_xor(hv, 32, 64);
        const uint32_t pn = (uint32_t)__shfl_xor((int)hn, 32, 64);
        const bool theirs = pn != 0xffffu && (hn == 0xffffu || pv > hv || (pv == hv && pn < hn));
        best[t] = theirs ? pn : hn;
    }
    return lane_e < 32 ? best[0] : best[1]; // lane L = (m = L % 32, half L / 32): tile L / 32 holds environment b0 + L
}

template <class ROW>
__global__ __launch_bounds__(256) void k_qnet(Consts c, State s, const float *pk, float *q_out, int n_out) {
    using Q = QNet<ROW>;
    extern __shared__ float w1[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * Q::kEnvsPerWave;
    // the state words of this lane's two environments are requested first: they arrive while the LDS image is being filled
    uint32_t aw[2][ROW::A];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int64_t b = b0 + 32 * t + (lane & 31);
#pragma unroll
        for (int i = 0; i < ROW::A; i++) { // (no branch around the load -- see qnet_wave's epilogue; a wave past the batch reads row 0)
            const uint32_t wv = (uint32_t)s.agent[(size_t)i * c.Bp + (b < c.B ? b : 0)];
            aw[t][i] = b < c.B ? wv : 0u;
        }
    }
    { // the LDS image: every load in flight before the first write (one memory round trip, not kFill of them)
        constexpr int kFill = (Q::kLdsFloats / 4 + Q::kThreads - 1) / Q::kThreads;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(pk);
        f32x4 *dst = reinterpret_cast<f32x4 *>(w1);
        f32x4 tmp[kFill];
#pragma unroll
        for (int i = 0; i < kFill; i++) {
            const int k = (int)threadIdx.x + Q::kThreads * i;
            tmp[i] = src[k < Q::kLdsFloats / 4 ? k : 0];
        }
#pragma unroll
        for (int i = 0; i < kFill; i++) {
            const int k = (int)threadIdx.x + Q::kThreads * i;
            if (k < Q::kLdsFloats / 4) dst[k] = tmp[i];
        }
    }
    __syncthreads();
    if (b0 >= c.B) return; // (after the only barrier)
    bool unit = true; // wave-uniform: scalar loads and compares
#pragma unroll
    for (int l = 0; l < 4; l++) unit = unit && pk[Q::oSlope + l] >= 0.0f && pk[Q::oSlope + l] <= 1.0f;
    if (unit) qnet_wave<ROW, true>(c, aw, pk, w1, w1 + Q::oB2, q_out, n_out, b0, lane);
    else qnet_wave<ROW, false>(c, aw, pk, w1, w1 + Q::oB2, q_out, n_out, b0, lane);
}

// The whole policy tick in ONE kernel (visualize.py:547-582 with a reference MLP as the imposters' network and a random crew): the
// Q-network as above, the argmax taken where the Q row lives, then the wave steps its own 64 environments (k_step's body, susnet_kernels.h
// step_wave: the crew's draws from the action stream, the step, the in-step reset, the fused observation) -- no Q rows, no actions and no
// second launch in between.  Dynamic LDS: the network image, then one step region of step_lds_bytes per wave.  q_out may be NULL.
// n_ticks > 1 (susnet_qnet_policy_rollout): the kernel stays resident for a block of ticks -- the network image is copied to LDS once, a
// launch is paid once -- and the outputs of tick k go to slot k of [T][B] arrays (ts: bytes between consecutive ticks of each output).
// A wave only ever reads what it wrote itself (its own 64 environments): the stores of tick k are made visible to its loads of tick k + 1
// by a release / acquire fence pair at WAVEFRONT scope: the lanes of one wave share the CU's vector L1 and their memory operations are
// performed in issue order, so the pair costs no instruction -- it only keeps the compiler from moving a load of tick k + 1 above a store
// of tick k.  (Agent scope, the first version, is an L2 write-back plus an L1 invalidate per wave and tick: 59.9 against 55.1 us per tick
// on one box.)
#ifndef SUSNET_TICK_FENCE_SCOPE
#define SUSNET_TICK_FENCE_SCOPE "wavefront"
#endif
struct TickStrides {
    int64_t actions, rewards, done, trunc, term_obs, roles, q, q_crew; // bytes
};
// All the arguments as ONE by-value struct, so the kernel can re-read them from the kernel-argument segment inside the tick loop (scalar
// loads through a pointer the optimiser cannot see through) instead of carrying ~100 scalar registers across the Q-network, where every
// vector register is already spoken for.
struct QStepArgs {
    Consts c;
    State s;
    const float *pk;
    float *q_out;
    int n_out;
    StepArgs a;
    ObsArgs o;
    int step_lds_bytes, n_ticks;
    TickStrides ts;
    // the CREW's network (run_game drives both teams by their networks, visualize.py:547-562; train_crew, notebooks/experiment.ipynb cell 5):
    // a second packed image of the same feature layout, or NULL = the crew draws from the action stream
    const float *pk_crew;
    float *q_crew_out;
    int n_out_crew, pad_;
};
typedef __attribute__((address_space(4))) const char *KernargPtr;
template <class T>
__device__ __forceinline__ T kernarg_read(KernargPtr base, size_t off) {
    T v; // (the typed pointer carries T's alignment: the copy becomes scalar loads, not byte-aligned vector loads)
    __builtin_memcpy(&v, (const __attribute__((address_space(4))) T *)(base + off), sizeof(T));
    return v;
}
#define SUSNET_KARG(base, field) kernarg_read<decltype(QStepArgs::field)>(base, offsetof(QStepArgs, field))

// The network image of a packed network -> LDS, all 256 threads: every load in flight before the first write (one memory round trip)
// tid: the thread's index in the workgroup -- inside a tick loop an OPAQUE copy (the per-thread addresses and range tests below are loop
// invariants: hoisted, they would be held across the matrix section)
template <class ROW>
__device__ __forceinline__ void qnet_fill_image(const float *pk, float *w1, int tid) {
    using Q = QNet<ROW>;
    constexpr int kFill = (Q::kLdsFloats / 4 + Q::kThreads - 1) / Q::kThreads;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(pk);
    f32x4 *dst = reinterpret_cast<f32x4 *>(w1);
    f32x4 tmp[kFill];
#pragma unroll
    for (int i = 0; i < kFill; i++) {
        const int k = tid + Q::kThreads * i;
        tmp[i] = src[k < Q::kLdsFloats / 4 ? k : 0];
    }
#pragma unroll
    for (int i = 0; i < kFill; i++) {
        const int k = tid + Q::kThreads * i;
        if (k < Q::kLdsFloats / 4) dst[k] = tmp[i];
    }
}

// The both-teams tick's image swap: the layer-1 part of another network's image (kW1 floats, whole KiB) global -> LDS without passing
// through registers, 1 KiB per instruction and wave (global_load_lds_dwordx4: lane L's 16 bytes land at M0 + 16 L), piece p of the
// image by wave p % 4, rounds [r0, r1) of the image's kW1 / 1024.  Inline assembly on purpose: with the compiler's builtin every
// vector-memory wait behind a transfer becomes vmcnt(0) (two kinds of events pending on one counter: it assumes nothing about their order)
// and the next weight block's wait takes the whole transfer; as assembly the compiler does not see them -- the hardware counter is
// in order, so its waits for the weight stream stay correct but ALSO wait for every transfer issued before them: the caller issues a
// few per weight block, right behind that block's wait, so that they are a block's matrix instructions old when the next wait comes
// (all 26 at one point: 2 us gained of 6).  The caller waits (qnet_swap_wait) and synchronises the workgroup before anybody reads the image.
template <class ROW>
__device__ __forceinline__ void qnet_swap_issue(const float *pk_other, uint32_t w1_lds_addr, int wave, int lane, int r0, int r1) {
    using Q = QNet<ROW>;
    constexpr int kRounds = Q::kW1 * 4 / 4096;
    static_assert(Q::kW1 % 1024 == 0 && Q::oW1 == 0, "whole rounds of four KiB, at the start of the packed image");
    const uint32_t voff = (uint32_t)lane * 16u + (uint32_t)wave * 1024u;
    const uint64_t gbase = reinterpret_cast<uint64_t>(pk_other);
    const uint32_t lbase = w1_lds_addr + (uint32_t)wave * 1024u;
#pragma unroll
    for (int i = 0; i < kRounds; i++) {
        if (i >= r0 && i < r1) { // (compile-time once inlined: no run-time condition around the assembly -- inside a branch the compiler takes
                                 // for divergent it hands the scalar operands over in vector registers)
            const uint32_t la = lbase + (uint32_t)i * 4096u;
            const uint64_t gb = gbase + (uint64_t)i * 4096u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(la), "v"(voff), "s"(gb) : "memory");
        }
    }
}
template <class ROW>
constexpr int qnet_swap_rounds() { return QNet<ROW>::kW1 * 4 / 4096; }
__device__ __forceinline__ void qnet_swap_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// BOTH TEAMS by their networks (pk_crew != NULL): per tick the workgroup runs the imposters' network and the crew's on ONE LDS image
// slot.  A pass reads the layer-1 part of its image in its first stage only (layers 1 + 2); right behind that stage the workgroup
// synchronises (everybody is done with it) and starts the transfer of the OTHER network's layer-1 part into the slot (qnet_swap_issue),
// which the rest of the pass -- 320 matrix instructions -- covers; before the next pass's first stage: wait + barrier.  The biases of
// layers 2..5 are read throughout a pass, so both networks' stay resident (the second set behind the image).  Each wave steps its
// environments with both argmaxes.  Waves past the batch stay in the loop for the fills and barriers (they hold no environment:
// no network pass, no step).  With the crew's network and no exploration nothing is drawn from the action stream, so numpy-tape handles
// are served too (RNG = TapeRng: the reference's collection loop with two networks, tests/golden/collect_*.npz).
// TWO: both networks (a kernel of its own: the one-network tick keeps its register budget).
constexpr int kStashWords = 128; // per wave, in front of its step region: the two teams' greedy actions across the network passes
template <class ROW, class S, class RNG = PhiloxRng, bool TWO = false>
__global__ __launch_bounds__(256) void k_qnet_step(QStepArgs ka) {
    using Q = QNet<ROW>;
    // dynamic LDS: [the step's table image, at address 0: its readers use absolute addresses][the network image][4 wave regions]
    extern __shared__ uint32_t dyn[];
    float *w1 = reinterpret_cast<float *>(dyn + kTableWords);
    static_assert((kTableWords * 4) % 16 == 0, "the network image is copied in 16-byte pieces");
    const int lane0 = threadIdx.x & 63, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t b00 = ((int64_t)blockIdx.x * 4 + wave0) * Q::kEnvsPerWave;
    qnet_fill_image<ROW>(ka.pk, w1, (int)threadIdx.x);
    if constexpr (TWO) { // the crew network's biases, resident behind the image
        if (threadIdx.x < Q::kTailFloats / 4)
            reinterpret_cast<f32x4 *>(w1 + Q::kLdsFloats)[threadIdx.x] = reinterpret_cast<const f32x4 *>(ka.pk_crew + Q::oB2)[threadIdx.x];
    }
    constexpr int kImageFloats = Q::kLdsFloats + (TWO ? Q::kTailFloats : 0);
    // the step's table image (launch-constant): written ONCE, by the first wave, and published to the other three by the barrier -- no
    // wave writes LDS words another one reads after this point (but for the image swaps of TWO, which have barriers of their own)
    if (wave0 == 0) {
        TableLoad<true> tl;
        tl.issue(ka.c, ka.o.comp, lane0);
        tl.commit(dyn, lane0, true);
    }
    __syncthreads();
    const bool live = b00 < ka.c.B;
    if (!TWO && !live) return; // (one network: no barrier past this point; the step below synchronises inside the wave only)
    const int n_ticks = ka.n_ticks;
    // one network pass of my 64 environments (pass 0: the imposters' network, 1: the crew's); the greedy action goes to the wave's stash
    // in LDS -- not a register: every vector register is spoken for inside qnet_wave, and a value carried across it costs scratch
    auto net_pass = [&](int pass, int k) __attribute__((always_inline)) {
        // (the optimiser would otherwise hoist every lane-dependent invariant of the step -- table-fill addresses, LDS offsets, ~150
        // vector registers of them -- out of the tick loop and keep them across the Q-network; an opaque copy of the lane id per use
        // site makes them per-tick values again)
        int lane = lane0, wave = wave0;
        int64_t b0 = b00;
        KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp), "+v"(lane), "+s"(wave), "+s"(b0));
        const Consts &c = ka.c; // (the tables: read in place)
        const auto *agent = SUSNET_KARG(kp, s.agent);
        const float *pk = pass ? SUSNET_KARG(kp, pk_crew) : SUSNET_KARG(kp, pk);
        float *q_out = pass ? SUSNET_KARG(kp, q_crew_out) : SUSNET_KARG(kp, q_out);
        const int n_out = pass ? SUSNET_KARG(kp, n_out_crew) : SUSNET_KARG(kp, n_out);
        const int64_t qstride = pass ? SUSNET_KARG(kp, ts.q_crew) : SUSNET_KARG(kp, ts.q);
        const int step_lds_bytes = SUSNET_KARG(kp, step_lds_bytes);
        bool unit = true; // wave-uniform: scalar loads and compares
#pragma unroll
        for (int l = 0; l < 4; l++) unit = unit && pk[Q::oSlope + l] >= 0.0f && pk[Q::oSlope + l] <= 1.0f;
        uint32_t aw[2][ROW::A];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int64_t b = b0 + 32 * t + (lane & 31);
#pragma unroll
            for (int i = 0; i < ROW::A; i++) { // (rows are padded to Bp: no branch around the load -- see qnet_wave's epilogue)
                const uint32_t wv = (uint32_t)agent[(size_t)i * c.Bp + b];
                aw[t][i] = b < c.B ? wv : 0u;
            }
        }
        float *qk = q_out ? reinterpret_cast<float *>(reinterpret_cast<char *>(q_out) + (int64_t)k * qstride) : nullptr;
        const float *btail = TWO && pass ? w1 + Q::kLdsFloats : w1 + Q::oB2;
        // both teams: behind the first stage the other network's layer-1 image starts to arrive, a share per weight block of layer 3 (see above)
        struct After {
            int pass;
            int lane0, wave0;
            __device__ __forceinline__ void synchronise() const {
                if constexpr (TWO) __syncthreads();
            }
            __device__ __forceinline__ void transfer(int i, int n) const {
                if constexpr (TWO) {
                    KernargPtr kq = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
                    int lane_a = lane0, wave_a = wave0;
                    asm volatile("" : "+s"(kq), "+v"(lane_a), "+s"(wave_a));
                    const float *other = pass ? SUSNET_KARG(kq, pk) : SUSNET_KARG(kq, pk_crew);
                    const int per = (qnet_swap_rounds<ROW>() + n - 1) / n;
                    qnet_swap_issue<ROW>(other, (uint32_t)kTableWords * 4u, wave_a, lane_a, i * per, (i + 1) * per);
                }
            }
        } after{pass, lane0, wave0};
        const uint32_t am = unit ? qnet_wave<ROW, true>(c, aw, pk, w1, btail, qk, n_out, b0, lane, after) : qnet_wave<ROW, false>(c, aw, pk, w1, btail, qk, n_out, b0, lane, after);
        int lane_s = lane0; // (a fresh copy: nothing lane-derived crosses the matrix section)
        asm volatile("" : "+v"(lane_s));
        uint32_t *stash = dyn + kTableWords + kImageFloats + (size_t)wave * (size_t)(step_lds_bytes / 4);
        stash[(pass ? 64 : 0) + lane_s] = am;
    };
#pragma clang loop unroll(disable)
    for (int k = 0; k < n_ticks; k++) {
        if (k > 0) { // what this wave stored at tick k - 1 (its environments' state) is what it loads now
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, SUSNET_TICK_FENCE_SCOPE);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SUSNET_TICK_FENCE_SCOPE);
        }
        if constexpr (!TWO) {
            net_pass(0, k);
        } else {
#pragma clang loop unroll(disable)
            for (int pass = 0; pass < 2; pass++) {
                if (pass > 0 || k > 0) { // this pass's image was requested behind the previous pass's first stage: it is whole once every wave's transfers are
                    qnet_swap_wait();
                    __syncthreads();
                }
                if (live) net_pass(pass, k);
                else { // a wave without environments: its share of the barrier and of the transfer
                    KernargPtr kq = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
                    int lane_a = lane0, wave_a = wave0;
                    asm volatile("" : "+s"(kq), "+v"(lane_a), "+s"(wave_a));
                    const float *other = pass ? SUSNET_KARG(kq, pk) : SUSNET_KARG(kq, pk_crew);
                    __syncthreads();
                    qnet_swap_issue<ROW>(other, (uint32_t)kTableWords * 4u, wave_a, lane_a, 0, qnet_swap_rounds<ROW>());
                }
            }
        }
        if (live) {
            int lane = lane0, wave = wave0;
            int64_t b0 = b00;
            KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp), "+v"(lane), "+s"(wave), "+s"(b0));
            const Consts &c = ka.c;
            const State s = SUSNET_KARG(kp, s);
            StepArgs ak = SUSNET_KARG(kp, a);
            const ObsArgs &o = ka.o; // (a component list indexed at run time: read in place)
            const TickStrides ts = SUSNET_KARG(kp, ts);
            const int step_lds_bytes = SUSNET_KARG(kp, step_lds_bytes);
            uint32_t *stash = dyn + kTableWords + kImageFloats + (size_t)wave * (size_t)(step_lds_bytes / 4);
            wave_lds_publish();
            const int a_imp = (int)stash[lane], a_crew = TWO ? (int)stash[64 + lane] : -1;
            uint32_t *rest = stash + kStashWords;
            auto shift = [&](auto *p, int64_t bytes) __attribute__((always_inline)) { return p ? reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + (int64_t)k * bytes) : p; };
            ak.actions = ak.actions ? static_cast<const void *>(static_cast<const char *>(ak.actions) + (int64_t)k * ts.actions) : nullptr;
            ak.rewards.ptr = ak.rewards.ptr ? static_cast<void *>(static_cast<char *>(ak.rewards.ptr) + (int64_t)k * ts.rewards) : nullptr;
            ak.done = shift(ak.done, ts.done);
            ak.trunc = shift(ak.trunc, ts.trunc);
            ak.term_obs = shift(ak.term_obs, ts.term_obs);
            ak.roles = shift(ak.roles, ts.roles);
            ak.tick = ak.tick + (uint64_t)k;
            step_wave<RNG, S, false>(c, s, ak, o, dyn, rest, lane, b0, a_imp, (int64_t)k, a_crew);
        }
    }
    if constexpr (TWO) qnet_swap_wait(); // (the last pass's transfer: nothing reads it, but it targets this workgroup's LDS)
}
#undef SUSNET_KARG

// ---- launchers: one translation unit per feature layout (inst_qnet_*.hip: these kernels are the library's largest), declared to the host
// side (susnet_capi.hip) as extern templates
template <class ROW>
hipError_t qnet_launch_k(const Consts &c, const State &s, const float *packed, float *q_out, int n_out, hipStream_t st) {
    using Q = QNet<ROW>;
    // ~106 KB of dynamic LDS: above the 64 KB a kernel gets without asking.  The attribute belongs to the CURRENT device's function
    // object, so it is set before every launch (a host-side table write: cheap; a process-wide "done" flag would leave a second
    // device without it and race between host threads)
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_qnet<ROW>), hipFuncAttributeMaxDynamicSharedMemorySize, Q::kLdsBytes)) return e;
    const unsigned blocks = (unsigned)((c.B + Q::kEnvsPerBlock - 1) / Q::kEnvsPerBlock);
    hipLaunchKernelGGL(k_qnet<ROW>, dim3(blocks), dim3(Q::kThreads), Q::kLdsBytes, st, c, s, packed, q_out, n_out);
    return hipGetLastError();
}
template <class ROW, class S, class RNG, bool TWO>
hipError_t qnet_step_launch_k(const QStepArgs &ka, size_t lds_bytes, hipStream_t st) {
    using Q = QNet<ROW>;
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_qnet_step<ROW, S, RNG, TWO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
    const unsigned blocks = (unsigned)((ka.c.B + Q::kEnvsPerBlock - 1) / Q::kEnvsPerBlock);
    hipLaunchKernelGGL((k_qnet_step<ROW, S, RNG, TWO>), dim3(blocks), dim3(Q::kThreads), lds_bytes, st, ka);
    return hipGetLastError();
}
using QRow1 = FlatRow<FEAT_ONEHOT, 2, 9>;                  // `onehot_pos` on the 1v1 9x9 game
using QRow3 = FlatRow<FEAT_ONEHOT_ALIVE_CLOSEST, 3, 14>;   // `onehot_pos + alive_crew + closest_crew` on the 1v2 14x14 game
using QRowC = CoordRow<2, 9>;                              // `coord_pos` on the 1v1 9x9 game
using QSpec2 = Spec<2, 0, SUSNET_VARIANT_ITG, 0, 0, 1>;
using QSpec3 = Spec<3, 4, SUSNET_VARIANT_BASE, 1, -1, 1>;
#define SUSNET_QNET_FOR(X, ROW, SPEC)                                                                                  \
    X template hipError_t qnet_launch_k<ROW>(const Consts &, const State &, const float *, float *, int, hipStream_t); \
    X template hipError_t qnet_step_launch_k<ROW, SPEC, PhiloxRng, false>(const QStepArgs &, size_t, hipStream_t);    \
    X template hipError_t qnet_step_launch_k<ROW, SPEC, PhiloxRng, true>(const QStepArgs &, size_t, hipStream_t);     \
    X template hipError_t qnet_step_launch_k<ROW, SPEC, TapeRng, true>(const QStepArgs &, size_t, hipStream_t);

} // namespace susnet
