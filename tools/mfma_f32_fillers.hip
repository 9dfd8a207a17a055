// mfma_f32_fillers.hip -- how many VALU / LDS instructions hide behind one v_mfma_f32_32x32x2_f32 (one wave per SIMD)?
// The fused Q-network kernel (sus-net_amd/csrc/susnet_qnet.h) gathers layer 1 on the VALU / LDS while layer 2 runs on the matrix
// core; this measures what that costs.  Every CU runs 4 waves (one per SIMD); a wave issues ITER x 8 MFMAs on 4 rotating
// accumulators with K filler instructions after each MFMA, pinned in place with sched_barrier.  Output: cycles per MFMA (s_memtime)
// for K = 0 .. 16 and each filler kind, as JSON lines.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_fillers tools/mfma_f32_fillers.hip && ./mfma_f32_fillers
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FILL_ADD = 0, FILL_FMA = 1, FILL_PKADD = 2, FILL_DSREAD = 3, FILL_CNDMASK = 4 };

template <int KIND, int K>
__device__ __forceinline__ void fillers(float (&x)[16], float y, const float *lds, f32x4 &sink, int lane) {
#pragma unroll
    for (int i = 0; i < K; i++) {
        if constexpr (KIND == FILL_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(y));
        if constexpr (KIND == FILL_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i % 16]) : "v"(y));
        if constexpr (KIND == FILL_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i % 16]) : "v"(y));
        if constexpr (KIND == FILL_PKADD) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 v = {x[(2 * i) % 16], x[(2 * i + 1) % 16]};
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(v));
            x[(2 * i) % 16] = v[0];
            x[(2 * i + 1) % 16] = v[1];
        }
        if constexpr (KIND == FILL_DSREAD) sink += *reinterpret_cast<const f32x4 *>(lds + ((lane * 4 + i * 260) & 8191));
    }
}

template <int KIND, int K>
__global__ __launch_bounds__(256) void k(float *out, long long *cycles, int iters) {
    __shared__ float lds[8192 + 64];
    for (int i = threadIdx.x; i < 8192 + 64; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4] = {};
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = (float)(lane + i);
    f32x4 sink = {};
    const float a = (float)lane * 1e-3f, b = 1.0f + (float)lane * 1e-4f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc[u % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fillers<KIND, K>(x, b, lds, sink, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink[0] + sink[1] + sink[2] + sink[3];
    for (int i = 0; i < 16; i++) s += x[i] + acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int K>
static void run(const char *name, float *out, long long *cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    // s_memtime ticks at 100 MHz on gfx9: report ticks and let the wall clock below give real cycles
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("{\"filler\": \"%s\", \"per_mfma\": %d, \"ns_per_mfma\": %.2f, \"memtime_ticks_per_mfma\": %.3f}\n", name, K, ms * 1e6 / (iters * 8.0),
                sum / h.size() / (iters * 8.0));
    std::fflush(stdout);
}

template <int KIND>
static void sweep(const char *name, float *out, long long *cyc, int blocks) {
    run<KIND, 0>(name, out, cyc, blocks);
    run<KIND, 2>(name, out, cyc, blocks);
    run<KIND, 4>(name, out, cyc, blocks);
    run<KIND, 8>(name, out, cyc, blocks);
    run<KIND, 12>(name, out, cyc, blocks);
    run<KIND, 16>(name, out, cyc, blocks);
}

int main() {
    const int blocks = 256;
    float *out;
    long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    sweep<FILL_ADD>("v_add_f32", out, cyc, blocks);
    sweep<FILL_FMA>("v_fma_f32", out, cyc, blocks);
    sweep<FILL_CNDMASK>("v_cndmask_b32", out, cyc, blocks);
    sweep<FILL_PKADD>("v_pk_add_f32", out, cyc, blocks);
    run<FILL_DSREAD, 0>("ds_read_b128", out, cyc, blocks);
    run<FILL_DSREAD, 1>("ds_read_b128", out, cyc, blocks);
    run<FILL_DSREAD, 2>("ds_read_b128", out, cyc, blocks);
    run<FILL_DSREAD, 4>("ds_read_b128", out, cyc, blocks);
    return 0;
}
