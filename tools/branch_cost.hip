// branch_cost.hip -- what a branch costs a wave that has its SIMD to itself (one wave per SIMD, every CU busy: the occupancy the fused
// rollouts run at).  Each test: 16 copies of (one v_add_u32 + one branch pattern) in one asm block, looped between two s_memtime reads;
// output = wall ns and s_memtime ticks per copy, JSON lines, the plain v_add_u32 first (subtract it).
//   hipcc -O3 --offload-arch=gfx950 -o branch_cost tools/branch_cost.hip && ./branch_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CLOB "v90", "v91", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v131", "vcc", "scc", "s20", "s21", "s22", "s23", "memory"

// one copy: BODY may use \\r for the copy's register number
#define REP16(P) P(100) P(101) P(102) P(103) P(104) P(105) P(106) P(107) P(108) P(109) P(110) P(111) P(112) P(113) P(114) P(115)
#define S_(x) #x
#define S(x) S_(x)

#define P_PLAIN(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\n"
#define P_NOT_TAKEN(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cbranch_scc0 1f\n1:\n"
#define P_TAKEN_SHORT(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_TAKEN_256B(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cbranch_scc1 1f\n.rept 32\nv_add_u32 v131, v90, v131\n.endr\n1:\n"
#define P_TAKEN_2KB(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cbranch_scc1 1f\n.rept 256\nv_add_u32 v131, v90, v131\n.endr\n1:\n"
#define P_UNCOND(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_branch 1f\nv_mov_b32 v131, v131\n1:\n"
// the ballot pattern of the step kernels: v_cmp -> vcc, branch on vcc zero (taken: vcc == 0)
#define P_VCCZ_TAKEN(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\nv_cmp_eq_u32 vcc, v91, v" S(r) "\ns_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_VCCZ_NOT_TAKEN(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\nv_cmp_ne_u32 vcc, v91, v" S(r) "\ns_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"
// ... and the same test without the branch (what the compare alone costs)
#define P_CMP_ONLY(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\nv_cmp_eq_u32 vcc, v91, v" S(r) "\n"
// ... with the result taken through s_cmp on the mask instead (ballot == 0 as a scalar compare)
#define P_SCMP_TAKEN(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\nv_cmp_eq_u32 s[22:23], v91, v" S(r) "\ns_cmp_eq_u64 s[22:23], 0\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"
// a taken branch whose target block ends in a branch back (an out-of-line rare block that IS executed: two taken branches)
#define P_OUT_OF_LINE(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cbranch_scc1 2f\n3:\n"

// dependent chains (every copy reads what the previous one wrote) against the independent form above
#define P_CHAIN_ADD(r) "v_add_u32 v100, v90, v100\n"
#define P_CHAIN2_ADD(r) "v_add_u32 v100, v90, v100\nv_add_u32 v101, v90, v101\n"
#define P_CHAIN_PERM(r) "v_perm_b32 v100, v90, v100, v91\n"
#define P_CHAIN_BITOP(r) "v_bitop3_b32 v100, v90, v91, v100 bitop3:0x96\n"
#define P_CHAIN_MAD64(r) "v_mad_u64_u32 v[100:101], vcc, v90, v100, v[100:101]\n"
#define P_IND_MAD64(r) "v_mad_u64_u32 v[100:101], vcc, v90, v91, v[102:103]\n"
#define P_CNDMASK_S(r) "v_cndmask_b32 v" S(r) ", v90, v" S(r) ", s[22:23]\n"
#define P_CHAIN_CNDMASK_S(r) "v_cndmask_b32 v100, v90, v100, s[22:23]\n"
#define P_CMP_CNDMASK(r) "v_cmp_eq_u32 vcc, v91, v" S(r) "\nv_cndmask_b32 v" S(r) ", v90, v" S(r) ", vcc\n"
#define P_CMPS_CNDMASK(r) "v_cmp_eq_u32 s[22:23], v91, v" S(r) "\nv_cndmask_b32 v" S(r) ", v90, v" S(r) ", s[22:23]\n"
#define P_SWAP(r) "v_permlane32_swap_b32 v" S(r) ", v131\n"
#define P_CHAIN_SWAP_USE(r) "v_permlane32_swap_b32 v100, v101\nv_add_u32 v100, v101, v100\n"
#define P_RFL_USE(r) "v_readfirstlane_b32 s20, v" S(r) "\nv_add_u32 v" S(r) ", s20, v" S(r) "\n"
#define P_SALU_CHAIN(r) "s_add_u32 s20, s20, 3\n"
#define P_VALU_SGPR_WRITE_READ(r) "s_add_u32 s20, s20, 3\nv_add_u32 v" S(r) ", s20, v" S(r) "\n"
#define P_LDS_RT(r) "ds_read_b32 v100, v131\ns_waitcnt lgkmcnt(0)\nv_and_b32 v131, 0xfc, v100\n"

// what makes a lone wave's vector instructions issue faster or slower: register patterns, operand kinds, instruction mixes
#define P_INLINE_CONST(r) "v_add_u32 v" S(r) ", 1, v" S(r) "\n"
#define P_ROT4(r) "v_add_u32 v100, v90, v100\nv_add_u32 v101, v90, v101\nv_add_u32 v102, v90, v102\nv_add_u32 v103, v90, v103\n"
#define P_ROT8(r) "v_add_u32 v100, v90, v100\nv_add_u32 v101, v90, v101\nv_add_u32 v102, v90, v102\nv_add_u32 v103, v90, v103\nv_add_u32 v104, v90, v104\nv_add_u32 v105, v90, v105\nv_add_u32 v106, v90, v106\nv_add_u32 v107, v90, v107\n"
#define P_MOV(r) "v_mov_b32 v" S(r) ", v90\n"
#define P_ADD_ADD_NOP(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\nv_add_u32 v131, v90, v131\ns_nop 0\n"
#define P_ADD_PERM(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\nv_perm_b32 v131, v90, v131, v91\n"
#define P_SWAR1(r) "v_add_u32 v100, 0x7b7b7b7b, v100\nv_and_b32 v100, 0x80808080, v100\nv_lshrrev_b32 v100, 7, v100\nv_perm_b32 v100, v90, v100, v91\nv_bitop3_b32 v100, v90, v91, v100 bitop3:0x96\nv_sub_u32 v100, v100, v90\n"
#define P_SWAR2(r) "v_add_u32 v100, 0x7b7b7b7b, v100\nv_add_u32 v101, 0x7b7b7b7b, v101\nv_and_b32 v100, 0x80808080, v100\nv_and_b32 v101, 0x80808080, v101\nv_lshrrev_b32 v100, 7, v100\nv_lshrrev_b32 v101, 7, v101\nv_perm_b32 v100, v90, v100, v91\nv_perm_b32 v101, v90, v101, v91\nv_bitop3_b32 v100, v90, v91, v100 bitop3:0x96\nv_bitop3_b32 v101, v90, v91, v101 bitop3:0x96\nv_sub_u32 v100, v100, v90\nv_sub_u32 v101, v101, v90\n"
#define P_SWAR3(r) "v_add_u32 v100, 0x7b7b7b7b, v100\nv_add_u32 v101, 0x7b7b7b7b, v101\nv_add_u32 v102, 0x7b7b7b7b, v102\nv_and_b32 v100, 0x80808080, v100\nv_and_b32 v101, 0x80808080, v101\nv_and_b32 v102, 0x80808080, v102\nv_lshrrev_b32 v100, 7, v100\nv_lshrrev_b32 v101, 7, v101\nv_lshrrev_b32 v102, 7, v102\nv_perm_b32 v100, v90, v100, v91\nv_perm_b32 v101, v90, v101, v91\nv_perm_b32 v102, v90, v102, v91\nv_bitop3_b32 v100, v90, v91, v100 bitop3:0x96\nv_bitop3_b32 v101, v90, v91, v101 bitop3:0x96\nv_bitop3_b32 v102, v90, v91, v102 bitop3:0x96\nv_sub_u32 v100, v100, v90\nv_sub_u32 v101, v101, v90\nv_sub_u32 v102, v102, v90\n"
#define P_SWAR1_NOPS(r) "v_add_u32 v100, 0x7b7b7b7b, v100\ns_nop 0\nv_and_b32 v100, 0x80808080, v100\ns_nop 0\nv_lshrrev_b32 v100, 7, v100\ns_nop 0\nv_perm_b32 v100, v90, v100, v91\ns_nop 0\nv_bitop3_b32 v100, v90, v91, v100 bitop3:0x96\ns_nop 0\nv_sub_u32 v100, v100, v90\ns_nop 0\n"

// the ballot branch with independent work between the compare and the branch (does the compare's way to the scalar side overlap?)
#define FILL8 "v_add_u32 v101, v90, v101\nv_add_u32 v102, v90, v102\nv_add_u32 v103, v90, v103\nv_add_u32 v104, v90, v104\nv_add_u32 v105, v90, v105\nv_add_u32 v106, v90, v106\nv_add_u32 v107, v90, v107\nv_add_u32 v108, v90, v108\n"
#define FILL4 "v_add_u32 v101, v90, v101\nv_add_u32 v102, v90, v102\nv_add_u32 v103, v90, v103\nv_add_u32 v104, v90, v104\n"
#define P_FILL8_ONLY(r) "v_add_u32 v100, v90, v100\n" FILL8
#define P_VCCZ_FILL8_NT(r) "v_add_u32 v100, v90, v100\nv_cmp_ne_u32 vcc, v91, v100\n" FILL8 "s_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_VCCZ_FILL8_T(r) "v_add_u32 v100, v90, v100\nv_cmp_eq_u32 vcc, v91, v100\n" FILL8 "s_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_VCCZ_FILL4_NT(r) "v_add_u32 v100, v90, v100\nv_cmp_ne_u32 vcc, v91, v100\n" FILL4 "s_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_SCMP_FILL8_NT(r) "v_add_u32 v100, v90, v100\nv_cmp_ne_u32 s[22:23], v91, v100\n" FILL8 "s_cmp_eq_u64 s[22:23], 0\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_SCMP_FILL8_T(r) "v_add_u32 v100, v90, v100\nv_cmp_eq_u32 s[22:23], v91, v100\n" FILL8 "s_cmp_eq_u64 s[22:23], 0\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_VCCZ_FILL8_AFTER_NT(r) "v_add_u32 v100, v90, v100\n" FILL8 "v_cmp_ne_u32 vcc, v91, v100\ns_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"

// gates on values that are already on the scalar side (a wave-uniform flag kept as a lane mask / as a 32-bit scalar)
#define P_SMASK_VCCNZ_NT(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_andn2_b64 vcc, exec, s[22:23]\ns_cbranch_vccz 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_SMASK_VCCNZ_T(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_andn2_b64 vcc, exec, s[22:23]\ns_cbranch_vccnz 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_SCMP64_NT(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cmp_lg_u64 s[22:23], 0\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_SCMP32_NT(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cmp_lg_u32 s21, 0\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"
#define P_SCMP32_T(r) "v_add_u32 v" S(r) ", v90, v" S(r) "\ns_cmp_eq_u32 s21, 0\ns_cbranch_scc1 1f\nv_mov_b32 v131, v131\n1:\n"

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, long long *cycles, int iters) {
    __shared__ uint32_t lds[64]; // (the LDS round-trip test walks it: every word holds a byte offset inside it)
    if (threadIdx.x < 64) lds[threadIdx.x] = (threadIdx.x * 52u + 4u) & 0xfcu;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t a = lane * 2654435761u + 12345u;
    asm volatile("v_mov_b32 v90, %0\nv_mov_b32 v91, 0x7fffffff\ns_mov_b32 s20, 77\nv_mov_b32 v131, 0" : : "v"(a) : CLOB);
    asm volatile(REP16(P_PLAIN) : : : CLOB);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) asm volatile(REP16(P_PLAIN) : : : CLOB);
        if constexpr (KIND == 1) asm volatile("s_cmp_eq_u32 s20, s20\n" REP16(P_NOT_TAKEN) : : : CLOB);
        if constexpr (KIND == 2) asm volatile("s_cmp_eq_u32 s20, s20\n" REP16(P_TAKEN_SHORT) : : : CLOB);
        if constexpr (KIND == 3) asm volatile("s_cmp_eq_u32 s20, s20\n" REP16(P_TAKEN_256B) : : : CLOB);
        if constexpr (KIND == 4) asm volatile("s_cmp_eq_u32 s20, s20\n" REP16(P_TAKEN_2KB) : : : CLOB);
        if constexpr (KIND == 5) asm volatile(REP16(P_UNCOND) : : : CLOB);
        if constexpr (KIND == 6) asm volatile(REP16(P_VCCZ_TAKEN) : : : CLOB);
        if constexpr (KIND == 7) asm volatile(REP16(P_VCCZ_NOT_TAKEN) : : : CLOB);
        if constexpr (KIND == 8) asm volatile(REP16(P_CMP_ONLY) : : : CLOB);
        if constexpr (KIND == 9) asm volatile(REP16(P_SCMP_TAKEN) : : : CLOB);
        if constexpr (KIND == 20) asm volatile(REP16(P_CHAIN_ADD) : : : CLOB);
        if constexpr (KIND == 21) asm volatile(REP16(P_CHAIN2_ADD) : : : CLOB);
        if constexpr (KIND == 22) asm volatile(REP16(P_CHAIN_PERM) : : : CLOB);
        if constexpr (KIND == 23) asm volatile(REP16(P_CHAIN_BITOP) : : : CLOB);
        if constexpr (KIND == 24) asm volatile(REP16(P_CHAIN_MAD64) : : : CLOB);
        if constexpr (KIND == 25) asm volatile(REP16(P_IND_MAD64) : : : CLOB, "v102", "v103");
        if constexpr (KIND == 26) asm volatile("s_mov_b64 s[22:23], 0x5555\n" REP16(P_CNDMASK_S) : : : CLOB);
        if constexpr (KIND == 27) asm volatile("s_mov_b64 s[22:23], 0x5555\n" REP16(P_CHAIN_CNDMASK_S) : : : CLOB);
        if constexpr (KIND == 28) asm volatile(REP16(P_CMP_CNDMASK) : : : CLOB);
        if constexpr (KIND == 29) asm volatile(REP16(P_CMPS_CNDMASK) : : : CLOB);
        if constexpr (KIND == 30) asm volatile(REP16(P_SWAP) : : : CLOB);
        if constexpr (KIND == 31) asm volatile(REP16(P_CHAIN_SWAP_USE) : : : CLOB);
        if constexpr (KIND == 32) asm volatile(REP16(P_RFL_USE) : : : CLOB);
        if constexpr (KIND == 33) asm volatile(REP16(P_SALU_CHAIN) : : : CLOB);
        if constexpr (KIND == 34) asm volatile(REP16(P_VALU_SGPR_WRITE_READ) : : : CLOB);
        if constexpr (KIND == 35) asm volatile(REP16(P_LDS_RT) : : : CLOB);
        if constexpr (KIND == 50) asm volatile(REP16(P_FILL8_ONLY) : : : CLOB);
        if constexpr (KIND == 51) asm volatile(REP16(P_VCCZ_FILL8_NT) : : : CLOB);
        if constexpr (KIND == 52) asm volatile(REP16(P_VCCZ_FILL8_T) : : : CLOB);
        if constexpr (KIND == 53) asm volatile(REP16(P_VCCZ_FILL4_NT) : : : CLOB);
        if constexpr (KIND == 54) asm volatile(REP16(P_SCMP_FILL8_NT) : : : CLOB);
        if constexpr (KIND == 55) asm volatile(REP16(P_SCMP_FILL8_T) : : : CLOB);
        if constexpr (KIND == 56) asm volatile(REP16(P_VCCZ_FILL8_AFTER_NT) : : : CLOB);
        if constexpr (KIND == 60) asm volatile("s_mov_b64 s[22:23], 0\n" REP16(P_SMASK_VCCNZ_NT) : : : CLOB);
        if constexpr (KIND == 61) asm volatile("s_mov_b64 s[22:23], 0\n" REP16(P_SMASK_VCCNZ_T) : : : CLOB);
        if constexpr (KIND == 62) asm volatile("s_mov_b64 s[22:23], 0\n" REP16(P_SCMP64_NT) : : : CLOB);
        if constexpr (KIND == 63) asm volatile("s_mov_b32 s21, 0\n" REP16(P_SCMP32_NT) : : : CLOB);
        if constexpr (KIND == 64) asm volatile("s_mov_b32 s21, 0\n" REP16(P_SCMP32_T) : : : CLOB);
        if constexpr (KIND == 40) asm volatile(REP16(P_INLINE_CONST) : : : CLOB);
        if constexpr (KIND == 41) asm volatile(REP16(P_ROT4) : : : CLOB);
        if constexpr (KIND == 42) asm volatile(REP16(P_ROT8) : : : CLOB);
        if constexpr (KIND == 43) asm volatile(REP16(P_MOV) : : : CLOB);
        if constexpr (KIND == 44) asm volatile(REP16(P_ADD_ADD_NOP) : : : CLOB);
        if constexpr (KIND == 45) asm volatile(REP16(P_ADD_PERM) : : : CLOB);
        if constexpr (KIND == 46) asm volatile(REP16(P_SWAR1) : : : CLOB);
        if constexpr (KIND == 47) asm volatile(REP16(P_SWAR2) : : : CLOB);
        if constexpr (KIND == 48) asm volatile(REP16(P_SWAR3) : : : CLOB);
        if constexpr (KIND == 49) asm volatile(REP16(P_SWAR1_NOPS) : : : CLOB);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s;
    asm volatile("v_add_u32 %0, v100, v115\nv_add_u32 %0, %0, v131" : "=v"(s) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[lane];
    if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char *name, uint32_t *out, long long *cyc) {
    const int blocks = 256, iters = 8000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters * 4);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int timed_iters = iters * 4;
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    std::printf("{\"pattern\": \"%s\", \"memtime_ticks_per_copy\": %.2f, \"ns_per_copy_wall\": %.3f}\n", name, sum / h.size() / (timed_iters * 16.0), ms * 1e6 / (timed_iters * 16.0));
    std::fflush(stdout);
}

int main() {
    uint32_t *out;
    long long *cyc;
    (void)hipMalloc(&out, sizeof(uint32_t) * 256 * 256);
    (void)hipMalloc(&cyc, sizeof(long long) * 256 * 4);
    for (int i = 0; i < 400; i++) hipLaunchKernelGGL((k<0>), dim3(256), dim3(256), 0, 0, out, cyc, 8000); // ~0.3 s: ramp the clocks first
    (void)hipDeviceSynchronize();
    run<0>("v_add_u32 alone", out, cyc);
    run<0>("v_add_u32 alone (again)", out, cyc);
    run<1>("+ s_cbranch_scc0, not taken", out, cyc);
    run<2>("+ s_cbranch_scc1, taken over one instruction", out, cyc);
    run<3>("+ s_cbranch_scc1, taken over 256 bytes", out, cyc);
    run<4>("+ s_cbranch_scc1, taken over 2 KB", out, cyc);
    run<5>("+ s_branch over one instruction", out, cyc);
    run<8>("+ v_cmp -> vcc (no branch)", out, cyc);
    run<6>("+ v_cmp -> vcc, s_cbranch_vccz taken", out, cyc);
    run<7>("+ v_cmp -> vcc, s_cbranch_vccz not taken", out, cyc);
    run<9>("+ v_cmp -> sgpr pair, s_cmp_eq_u64, s_cbranch_scc1 taken", out, cyc);
    run<20>("DEPENDENT chain of v_add_u32", out, cyc);
    run<21>("two interleaved chains of v_add_u32 (per PAIR)", out, cyc);
    run<22>("dependent chain of v_perm_b32", out, cyc);
    run<23>("dependent chain of v_bitop3_b32", out, cyc);
    run<25>("independent v_mad_u64_u32", out, cyc);
    run<24>("dependent chain of v_mad_u64_u32", out, cyc);
    run<26>("independent v_cndmask_b32 on an SGPR-pair mask", out, cyc);
    run<27>("dependent chain of v_cndmask_b32 on an SGPR-pair mask", out, cyc);
    run<28>("v_cmp -> vcc, v_cndmask vcc (per pair)", out, cyc);
    run<29>("v_cmp -> s[22:23], v_cndmask s[22:23] (per pair)", out, cyc);
    run<30>("independent v_permlane32_swap_b32", out, cyc);
    run<31>("v_permlane32_swap_b32 + dependent v_add_u32 (per pair)", out, cyc);
    run<32>("v_readfirstlane_b32 -> v_add_u32 reading the SGPR (per pair)", out, cyc);
    run<33>("dependent chain of s_add_u32", out, cyc);
    run<34>("s_add_u32 -> v_add_u32 reading the SGPR (per pair)", out, cyc);
    run<35>("ds_read_b32 round trip (address from the previous read) + v_and", out, cyc);
    run<50>("nine v_add_u32 (the filler of the next tests; per NINE)", out, cyc);
    run<56>("nine v_add_u32, THEN v_cmp -> vcc, s_cbranch_vccz not taken", out, cyc);
    run<51>("v_add, v_cmp -> vcc, EIGHT v_add, s_cbranch_vccz not taken", out, cyc);
    run<52>("v_add, v_cmp -> vcc, eight v_add, s_cbranch_vccz taken", out, cyc);
    run<53>("v_add, v_cmp -> vcc, FOUR v_add, s_cbranch_vccz not taken (per five + branch)", out, cyc);
    run<54>("v_add, v_cmp -> sgpr pair, eight v_add, s_cmp_eq_u64, s_cbranch_scc1 not taken", out, cyc);
    run<55>("v_add, v_cmp -> sgpr pair, eight v_add, s_cmp_eq_u64, s_cbranch_scc1 taken", out, cyc);
    run<60>("v_add, s_andn2_b64 vcc, exec, mask (set long before), s_cbranch_vccz not taken", out, cyc);
    run<61>("v_add, s_andn2_b64 vcc, exec, mask, s_cbranch_vccnz taken", out, cyc);
    run<62>("v_add, s_cmp_lg_u64 mask, 0, s_cbranch_scc1 not taken", out, cyc);
    run<63>("v_add, s_cmp_lg_u32 flag, 0, s_cbranch_scc1 not taken", out, cyc);
    run<64>("v_add, s_cmp_eq_u32 flag, 0, s_cbranch_scc1 taken", out, cyc);
    run<40>("v_add_u32 with an inline constant", out, cyc);
    run<41>("four rotating chains of v_add_u32 (per FOUR)", out, cyc);
    run<42>("eight rotating chains of v_add_u32 (per EIGHT)", out, cyc);
    run<43>("v_mov_b32", out, cyc);
    run<44>("v_add_u32, v_add_u32, s_nop 0 (per triple)", out, cyc);
    run<45>("v_add_u32 + v_perm_b32 (per pair)", out, cyc);
    run<46>("byte-parallel chain of six dependent instructions (per SIX)", out, cyc);
    run<47>("two such chains interleaved (per TWELVE)", out, cyc);
    run<48>("three such chains interleaved (per EIGHTEEN)", out, cyc);
    run<49>("the one chain with s_nop 0 between its instructions (per six + six)", out, cyc);
    return 0;
}
