// tools/store_patterns.hip -- what the write path of an MI355X does with the store patterns of the fused rollouts.
//
// 1 024 single-wave workgroups (one per SIMD, the geometry of bench.py's headline launch: 65 536 envs x 512 ticks), every lane
// one "environment"; per tick a lane burns VALU instructions (the step's arithmetic stand-in) and stores R bytes of "record".
// Patterns (R = 20 unless stated):
//   aos20      st128 + st32 at lane * 20 within the tick's slab (the round-2 packed record of cfg2)
//   aos32      2 x st128 at lane * 32 (record padded to a 32-byte sector)
//   aos16      st128 at lane * 16 (a 16-byte record: every store instruction covers 1 024 contiguous bytes)
//   soa16_4    st128 at chunk0 + lane * 16, st32 at chunk1 + lane * 4 (wave-blocked: [tick][wave][16 B x 64 | 4 B x 64])
//   lds20      the 20-byte records of the wave transposed through LDS, written as 80 contiguous 16-byte pieces (64 + 16 lanes)
//   aos40/80   cfg3 / cfg4 sized records as 16-byte stores at lane * R (+ an 8-byte tail)
//   none       no stores (the arithmetic alone)
// build: hipcc -O3 --offload-arch=gfx950 -o store_patterns tools/store_patterns.hip ; run: ./store_patterns [valu_per_tick]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }

enum { P_NONE, P_AOS20, P_AOS32, P_AOS16, P_SOA16_4, P_LDS20, P_AOS40, P_AOS80, P_SOA40, P_SOA80, P_COUNT };
static const char *kNames[] = {"none", "aos20", "aos32", "aos16", "soa16_4", "lds20", "aos40", "aos80", "soa40", "soa80"};
static const int kBytes[] = {0, 20, 32, 16, 20, 20, 40, 80, 40, 80};

template <int P>
__global__ __launch_bounds__(64) void k_store(uint8_t *out, uint32_t slab, int ticks, int valu, uint32_t *sink) {
    __shared__ uint32_t lds[64 * 5 + 16];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    constexpr uint32_t R = P == P_NONE ? 0 : (P == P_AOS20 || P == P_SOA16_4 || P == P_LDS20) ? 20 : P == P_AOS32 ? 32 : P == P_AOS16 ? 16 : (P == P_AOS40 || P == P_SOA40) ? 40 : 80;
    const __amdgpu_buffer_rsrc_t r = rsrc(out, slab * (uint32_t)ticks);
    uint32_t x = lane * 2654435761u + wave, y = wave ^ 0x9e3779b9u;
    const uint32_t wbase = wave * 64u * R;
    for (int t = 0; t < ticks; t++) {
        for (int k = 0; k < valu; k++) { x = x * 5u + y; y ^= x >> 7; } // (2 VALU per k, dependent)
        const uint32_t so = (uint32_t)t * slab;
        const u32x4 v = {x, y, x ^ y, x + y};
        if (P == P_AOS20) {
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 20u + so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(x, r, wbase + lane * 20u + 16u, so, 0);
        } else if (P == P_AOS32) {
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 32u + so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 32u + 16u + so, 0, 0);
        } else if (P == P_AOS16) {
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 16u + so, 0, 0);
        } else if (P == P_SOA16_4) {
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 16u + so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(x, r, wbase + 1024u + lane * 4u, so, 0);
        } else if (P == P_LDS20) {
            // records into LDS as they lie in memory (lane * 20 bytes), read back as 16-byte pieces
            uint32_t *mine = lds + lane * 5;
            mine[0] = v.x; mine[1] = v.y; mine[2] = v.z; mine[3] = v.w; mine[4] = x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const u32x4 a = *reinterpret_cast<const u32x4 *>(lds + lane * 4);
            __builtin_amdgcn_raw_buffer_store_b128(a, r, wbase + lane * 16u + so, 0, 0);
            if (lane < 16) {
                const u32x4 b = *reinterpret_cast<const u32x4 *>(lds + 256 + lane * 4);
                __builtin_amdgcn_raw_buffer_store_b128(b, r, wbase + 1024u + lane * 16u + so, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        } else if (P == P_AOS40) {
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 40u + so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 40u + 16u + so, 0, 0);
            const u32x2 w = {x, y};
            __builtin_amdgcn_raw_buffer_store_b64(w, r, wbase + lane * 40u + 32u, so, 0);
        } else if (P == P_SOA40) {
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 16u + so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + 1024u + lane * 16u + so, 0, 0);
            const u32x2 w = {x, y};
            __builtin_amdgcn_raw_buffer_store_b64(w, r, wbase + 2048u + lane * 8u, so, 0);
        } else if (P == P_AOS80) {
#pragma unroll
            for (int k = 0; k < 5; k++) __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + lane * 80u + 16u * k + so, 0, 0);
        } else if (P == P_SOA80) {
#pragma unroll
            for (int k = 0; k < 5; k++) __builtin_amdgcn_raw_buffer_store_b128(v, r, wbase + 1024u * k + lane * 16u + so, 0, 0);
        }
    }
    if (x == 0x12345u && y == 0x54321u) sink[0] = x; // keep the arithmetic
}

template <int P>
static float run(uint8_t *out, size_t cap, int waves, int ticks, int valu, uint32_t *sink, int reps) {
    const uint32_t slab = (uint32_t)waves * 64u * (uint32_t)kBytes[P];
    if ((size_t)slab * ticks > cap) { printf("%s: buffer too small\n", kNames[P]); return -1.f; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_store<P>, dim3(waves), dim3(64), 0, 0, out, slab ? slab : 4u, ticks, valu, sink);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_store<P>, dim3(waves), dim3(64), 0, 0, out, slab ? slab : 4u, ticks, valu, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int waves = 1024, ticks = 512, reps = 10;
    std::vector<int> valus;
    for (int i = 1; i < argc; i++) valus.push_back(atoi(argv[i]));
    if (valus.empty()) valus = {0, 40, 120, 200};
    const size_t cap = (size_t)waves * 64 * 80 * ticks;
    uint8_t *out; uint32_t *sink;
    if (hipMalloc(&out, cap) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(out, 0, cap);
    printf("{\"waves\": %d, \"ticks\": %d, \"rows\": [\n", waves, ticks);
    bool first = true;
    for (int valu : valus) {
        float ms[P_COUNT];
        ms[P_NONE] = run<P_NONE>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_AOS20] = run<P_AOS20>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_AOS32] = run<P_AOS32>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_AOS16] = run<P_AOS16>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_SOA16_4] = run<P_SOA16_4>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_LDS20] = run<P_LDS20>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_AOS40] = run<P_AOS40>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_AOS80] = run<P_AOS80>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_SOA40] = run<P_SOA40>(out, cap, waves, ticks, valu, sink, reps);
        ms[P_SOA80] = run<P_SOA80>(out, cap, waves, ticks, valu, sink, reps);
        for (int p = 0; p < P_COUNT; p++) {
            const double bytes = (double)waves * 64 * kBytes[p] * ticks;
            printf("%s {\"valu_per_tick\": %d, \"pattern\": \"%s\", \"record_bytes\": %d, \"us\": %.1f, \"GBs\": %.0f, \"cycles_per_tick_at_2.4GHz\": %.0f}",
                   first ? " " : ",\n ", 2 * valu, kNames[p], kBytes[p], ms[p] * 1e3, ms[p] > 0 ? bytes / (ms[p] * 1e-3) / 1e9 : 0.0, ms[p] * 1e-3 / ticks * 2.4e9);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
