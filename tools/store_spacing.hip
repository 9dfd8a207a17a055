// tools/store_spacing.hip -- does a wave pay for its stores only when they are issued back to back?
// Same geometry as tools/store_patterns.hip (1 024 single-wave workgroups x 512 ticks, 64 lanes).  Per tick: N = `valu` cheap
// independent VALU instructions and the 20-byte record (st128 + st32 at lane * 20) or the 40-byte record (st128, st128, st64),
// with the stores either adjacent at the end of the tick or spread evenly between slices of the arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }

// `n` cheap VALU instructions on four independent chains (not foldable: each depends on the loop-carried values)
template <int N>
__device__ __forceinline__ void burn(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) {
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
        a = (a ^ b) + 0x9e3779b9u; asm volatile("" : "+v"(a));
        b = (b + c) ^ 0x7f4a7c15u; asm volatile("" : "+v"(b));
        c = (c ^ d) + 0x85ebca6bu; asm volatile("" : "+v"(c));
        d = (d + a) ^ 0xc2b2ae35u; asm volatile("" : "+v"(d));
    }
}

// MODE 0: no stores; 1: 20 B adjacent; 2: 20 B spread; 3: 40 B adjacent; 4: 40 B spread; 5: 80 B (5 x st128) adjacent; 6: 80 B spread
template <int MODE, int VALU>
__global__ __launch_bounds__(64) void k(uint8_t *out, uint32_t slab, int ticks, uint32_t *sink) {
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    constexpr uint32_t R = MODE == 0 ? 4 : MODE <= 2 ? 20 : MODE <= 4 ? 40 : 80;
    const __amdgpu_buffer_rsrc_t r = rsrc(out, slab * (uint32_t)ticks);
    uint32_t a = lane * 2654435761u + wave, b = wave ^ 0x9e3779b9u, c = lane, d = wave + 7u;
    const uint32_t base = wave * 64u * R + lane * R;
    for (int t = 0; t < ticks; t++) {
        const uint32_t so = (uint32_t)t * slab;
        const u32x4 v = {a, b, c, d};
        const u32x2 w = {a, b};
        if (MODE == 0) { burn<VALU>(a, b, c, d); }
        if (MODE == 1) { burn<VALU>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + so, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(a, r, base + 16u, so, 0); }
        if (MODE == 2) { burn<VALU / 2>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + so, 0, 0); burn<VALU / 2>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b32(a, r, base + 16u, so, 0); }
        if (MODE == 3) { burn<VALU>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + so, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + 16u + so, 0, 0); __builtin_amdgcn_raw_buffer_store_b64(w, r, base + 32u, so, 0); }
        if (MODE == 4) {
            burn<VALU / 3>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + so, 0, 0);
            burn<VALU / 3>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + 16u + so, 0, 0);
            burn<VALU / 3>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b64(w, r, base + 32u, so, 0);
        }
        if (MODE == 5) {
            burn<VALU>(a, b, c, d);
#pragma unroll
            for (int q = 0; q < 5; q++) __builtin_amdgcn_raw_buffer_store_b128(v, r, base + 16u * q + so, 0, 0);
        }
        if (MODE == 6) {
#pragma unroll
            for (int q = 0; q < 5; q++) { burn<VALU / 5>(a, b, c, d); __builtin_amdgcn_raw_buffer_store_b128(v, r, base + 16u * q + so, 0, 0); }
        }
    }
    if (a == 0x12345u && b == 0x54321u) sink[0] = a + c + d;
}

template <int MODE, int VALU>
static void run(uint8_t *out, uint32_t *sink, const char *name) {
    const int waves = 1024, ticks = 512, reps = 10;
    const uint32_t R = MODE == 0 ? 4 : MODE <= 2 ? 20 : MODE <= 4 ? 40 : 80, slab = waves * 64u * R;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k<MODE, VALU>), dim3(waves), dim3(64), 0, 0, out, slab, ticks, sink);
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<MODE, VALU>), dim3(waves), dim3(64), 0, 0, out, slab, ticks, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("{\"valu_per_tick\": %d, \"stores\": \"%s\", \"us\": %.1f, \"cycles_per_tick_at_2.4GHz\": %.0f}\n", VALU, name, ms * 1e3, ms * 1e-3 / ticks * 2.4e9);
}
template <int VALU>
static void sweep(uint8_t *out, uint32_t *sink) {
    run<0, VALU>(out, sink, "none");
    run<1, VALU>(out, sink, "20B adjacent (st128 st32)");
    run<2, VALU>(out, sink, "20B spread");
    run<3, VALU>(out, sink, "40B adjacent (st128 st128 st64)");
    run<4, VALU>(out, sink, "40B spread");
    run<5, VALU>(out, sink, "80B adjacent (5 x st128)");
    run<6, VALU>(out, sink, "80B spread");
}
int main() {
    uint8_t *out; uint32_t *sink;
    if (hipMalloc(&out, (size_t)1024 * 64 * 80 * 512) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    sweep<60>(out, sink);
    sweep<120>(out, sink);
    sweep<240>(out, sink);
    sweep<480>(out, sink);
    return 0;
}
