// Does a wavefront whose EXEC mask has only 16 (or 32) lanes set issue VALU instructions faster on gfx950?
// (K1's lone-wavefront floor is one instruction per ~4.14 cycles; a one-chain wavefront has 16 live lanes.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define BODY8(X) X X X X X X X X
#define BODY32(X) BODY8(X) BODY8(X) BODY8(X) BODY8(X)
#define STEPMIX "v_mad_i32_i24 %0, %0, %1, %2\n v_ashrrev_i32 %0, 1, %0\n v_dot2_i32_i16 %0, %1, %2, %0\n v_sub_u32 %0, %1, %0\n"
template <int LIVE>
__global__ void k(unsigned long long *out, int *sink, int seed) {
    int a = threadIdx.x + seed, b = threadIdx.x * 3 + 1, c = seed;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < LIVE) {
        t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long c0 = __builtin_readcyclecounter();
        for (int i = 0; i < REP; i++) asm volatile(BODY32(STEPMIX) : "+v"(a) : "v"(b), "v"(c));
        unsigned long long c1 = __builtin_readcyclecounter();
        t1 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((threadIdx.x & 63) == 0) { out[3 * (threadIdx.x >> 6)] = c1 - c0; out[3 * (threadIdx.x >> 6) + 1] = t1 - t0; out[3 * (threadIdx.x >> 6) + 2] = hwid; }
    }
    sink[threadIdx.x] = a;
}
int main() {
    unsigned long long *out, h[48]; int *sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096 * 4);
    auto run = [&](auto kern, const char *name) {
        for (int w : {4, 8, 16}) {
            kern<<<1, 64 * w>>>(out, sink, 1); hipDeviceSynchronize();
            kern<<<1, 64 * w>>>(out, sink, 1); hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
            printf("%-10s %2d waves/block: %.3f cycles/instr (s_memtime), %.3f x10ns/instr; simd ids:", name, w, (double)h[0] / (REP * 128), (double)h[1] / (REP * 128));
            for (int i = 0; i < w; i++) printf(" %llu/cu%llu", (h[3 * i + 2] >> 4) & 3, (h[3 * i + 2] >> 8) & 15);
            printf("  wave cycles:");
            for (int i = 0; i < w; i++) printf(" %.2f", (double)h[3 * i] / (REP * 128));
            printf("\n");
        }
    };
    run(k<64>, "64 live"); run(k<32>, "32 live"); run(k<16>, "16 live"); run(k<1>, "1 live");
    return 0;
}
