// Is a lone wavefront's issue rate on gfx950 limited by instruction BYTES?  Bodies of independent v_add_u32 in the 4-byte
// (e32) and 8-byte (e64) encoding, in the size pattern of K1's cheap step (12 x 8 B + 10 x 4 B per 22 instructions) and others.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#define REP 512
#define A4 "v_add_u32_e32 %0, %0, %1\n"
#define A8 "v_add_u32_e64 %0, %0, %1\n"
#define B4 "v_add_u32_e32 %2, %2, %1\n"
#define B8 "v_add_u32_e64 %2, %2, %1\n"
// K1 cheap step order: 8,8,8,4,4,8,8,4,4,8,4,4,8,8,8,4,4,4,4,8,8,8
#define STEP_K1 A8 B8 A8 B4 A4 B8 A8 B4 A4 B8 A4 B4 A8 B8 A8 B4 A4 B4 A4 B8 A8 B8
#define STEP_ALT A8 B4 A8 B4 A8 B4 A8 B4 A8 B4 A8 B4 A8 B4 A8 B4 A8 B4 A8 B4 A8 B8
#define STEP_ALL4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4 A4 B4
#define STEP_ALL8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8
#define STEP_8_OF_22 A8 B4 A4 B8 A4 B4 A8 B4 A4 B8 A4 B4 A8 B4 A4 B8 A4 B4 A8 B4 A8 B4
#define STEP_10_OF_22 A8 B4 A8 B4 A4 B8 A4 B8 A4 B4 A8 B4 A8 B4 A4 B8 A4 B8 A4 B4 A8 B4
#define X4(S) S S S S
#define KERNEL(name, BODY)                                                                              \
    __global__ void name(unsigned long long *out, int *sink, int seed) {                                 \
        int a = threadIdx.x + seed, b = threadIdx.x * 3 + 1, c = seed;                                    \
        unsigned long long c0 = __builtin_readcyclecounter();                                             \
        for (int i = 0; i < REP; i++) asm volatile(X4(BODY) : "+v"(a), "+v"(c) : "v"(b));                 \
        unsigned long long c1 = __builtin_readcyclecounter();                                             \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = c1 - c0;                                     \
        sink[threadIdx.x] = a + c;                                                                        \
    }
// operands: %0 = a, %1 = b(input) -> careful: "+v"(a)=%0, "+v"(c)=%1, "v"(b)=%2; remap through the macros below
#undef A4
#undef A8
#undef B4
#undef B8
#define A4 "v_add_u32_e32 %0, %0, %2\n"
#define A8 "v_add_u32_e64 %0, %0, %2\n"
#define B4 "v_add_u32_e32 %1, %1, %2\n"
#define B8 "v_add_u32_e64 %1, %1, %2\n"
KERNEL(k_k1, STEP_K1)
KERNEL(k_alt, STEP_ALT)
KERNEL(k_all4, STEP_ALL4)
KERNEL(k_all8, STEP_ALL8)
KERNEL(k_8of22, STEP_8_OF_22)
KERNEL(k_10of22, STEP_10_OF_22)
int main() {
    unsigned long long *out, h[16]; int *sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096 * 4);
    struct T { const char *n; void (*k)(unsigned long long *, int *, int); int b8; } t[] = {
        {"K1 step order (12 x 8 B of 22)", k_k1, 12}, {"alternating (11+1 x 8 B of 22)", k_alt, 12}, {"all 4 B", k_all4, 0},
        {"all 8 B", k_all8, 22}, {"8 x 8 B of 22", k_8of22, 8}, {"10 x 8 B of 22", k_10of22, 10}};
    for (auto &x : t)
        for (int w : {1, 4, 8}) {
            x.k<<<1, 64 * w>>>(out, sink, 1); hipDeviceSynchronize();
            x.k<<<1, 64 * w>>>(out, sink, 1); hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(unsigned long long) * w, hipMemcpyDeviceToHost);
            printf("%-34s %.2f B/instr  %d waves in the CU: %.3f cycles/instr (wave 0), last wave %.3f\n", x.n, 4.0 + 4.0 * x.b8 / 22, w,
                   (double)h[0] / (REP * 88), (double)h[w - 1] / (REP * 88));
        }
    return 0;
}
