// What do K1's exit tests cost a lone wavefront on gfx950?  v_cmp -> s_cbranch_vccnz (never taken), s_cmp -> s_cbranch_scc,
// taken branches, VALU -> SGPR -> SALU, each between plain VALU instructions.  Cycles per body, one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define X8(S) S S S S S S S S
#define X16(S) X8(S) X8(S)
#define KERNEL(name, BODY)                                                                              \
    __global__ void name(unsigned long long *out, int *sink, int seed) {                                 \
        int a = threadIdx.x + seed, b = (threadIdx.x & 7) + 1, c = seed;                                  \
        unsigned long long c0 = __builtin_readcyclecounter();                                             \
        for (int i = 0; i < REP; i++) asm volatile(X16(BODY) "s_branch Lskip%=\n Lnever%=:\n s_nop 0\n Lskip%=:\n" : "+v"(a), "+v"(c) : "v"(b) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
        unsigned long long c1 = __builtin_readcyclecounter();                                             \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = c1 - c0;                                     \
        sink[threadIdx.x] = a + c;                                                                        \
    }
#define V4 "v_add_u32 %0, %0, %2\n v_xor_b32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_xor_b32 %1, %1, %2\n"
KERNEL(k_base, V4)
KERNEL(k_cmp_only, V4 "v_cmp_gt_i32 vcc, 0, %2\n")
KERNEL(k_cmp_br, V4 "v_cmp_gt_i32 vcc, 0, %2\n s_cbranch_vccnz Lnever%=\n")
KERNEL(k_cmp_v_br, V4 "v_cmp_gt_i32 vcc, 0, %2\n v_add_u32 %0, %0, %2\n v_xor_b32 %1, %1, %2\n s_cbranch_vccnz Lnever%=\n")
KERNEL(k_cmp_br2, V4 "v_cmp_gt_i32 vcc, 0, %2\n s_cbranch_vccnz Lnever%=\n" V4 "v_cmp_gt_i32 vcc, 0, %2\n s_cbranch_vccnz Lnever%=\n")
KERNEL(k_scmp_br, V4 "s_cmp_lt_u32 s20, s20\n s_cbranch_scc1 Lnever%=\n")
KERNEL(k_taken, V4 "s_branch 0\n")
KERNEL(k_cmp_sgpr_or, V4 "v_cmp_gt_i32 vcc, 0, %2\n v_cmp_gt_i32 s[20:21], 0, %2\n s_or_b64 vcc, vcc, s[20:21]\n s_cbranch_vccnz Lnever%=\n")
KERNEL(k_salu2, V4 "s_add_u32 s22, s22, 1\n s_cmp_lt_u32 s22, s23\n")
int main() {
    unsigned long long *out, h[16]; int *sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096 * 4);
    struct T { const char *n; void (*k)(unsigned long long *, int *, int); } t[] = {
        {"4 VALU", k_base}, {"4 VALU + v_cmp", k_cmp_only}, {"4 VALU + v_cmp + s_cbranch_vccnz (not taken)", k_cmp_br},
        {"4 VALU + v_cmp + 2 VALU + s_cbranch_vccnz", k_cmp_v_br}, {"2 x (4 VALU + v_cmp + branch) [per half]", k_cmp_br2},
        {"4 VALU + s_cmp + s_cbranch_scc1 (not taken)", k_scmp_br}, {"4 VALU + s_branch (taken, next instr)", k_taken},
        {"4 VALU + 2 v_cmp + s_or_b64 + branch", k_cmp_sgpr_or}, {"4 VALU + s_add + s_cmp", k_salu2}};
    for (auto &x : t) {
        x.k<<<1, 256>>>(out, sink, 1); hipDeviceSynchronize();
        x.k<<<1, 256>>>(out, sink, 1); hipDeviceSynchronize();
        hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("%-52s %.2f cycles per body\n", x.n, (double)h[0] / (REP * 16));
    }
    return 0;
}
