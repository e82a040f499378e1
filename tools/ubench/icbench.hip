// Does a large straight-line VALU body (instruction-cache footprint) still overlap with a saturating store stream?
// (development aid for K2: its hot loop is ~28 KB of code inside a 57-KB kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u4 *gptr;

// BODY x 8-byte VOP3 instructions straight-line, repeated REP times per chunk: 3072 VALU per chunk in every variant
template <int BODY, int REP, bool STORES>
__global__ void __launch_bounds__(256) k(unsigned char *out, unsigned n_frames, unsigned frame_bytes) {
    extern __shared__ unsigned s_pad[];
    if (n_frames == 0xffffffffu) s_pad[threadIdx.x] = 1;
    unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned piece_bytes = 320, ppf = 20, fpw = 32, np = 10, chunks = frame_bytes / piece_bytes;
    size_t base = (size_t)wave * fpw * frame_bytes;
    if ((wave + 1) * fpw > n_frames) return;
    u4 v = {1, 2, 3, lane};
    for (unsigned c = 0; c < chunks; c++) {
        for (int r = 0; r < REP; r++) {
            asm volatile(".rept %4\n v_add3_u32 %0, %0, %1, 1\n v_add3_u32 %1, %1, %2, 1\n v_add3_u32 %2, %2, %3, 1\n v_add3_u32 %3, %3, %0, 1\n .endr"
                         : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "n"(BODY / 4));
        }
        for (unsigned t = 0; t < np; t++) {
            unsigned p = lane + 64 * t, g = p / ppf, kk = p % ppf;
            gptr dst = (gptr)(out + base + (size_t)g * frame_bytes + (size_t)(STORES ? c : 0) * piece_bytes + kk * 16);
            __builtin_nontemporal_store(v, dst);
        }
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    }
}
int main() {
    size_t bytes = 5368709120ull;
    unsigned char *buf; (void)hipMalloc(&buf, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    unsigned n_frames = 262144, fb = 20480;
    auto time = [&](const char *name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %.3f ms\n", name, ms);
    };
#define RUN(B, R) \
    time("body " #B " x8B x" #R " + stores", [&] { hipLaunchKernelGGL((k<B, R, true>), dim3(n_frames / 128), dim3(256), 43000, 0, buf, n_frames, fb); }); \
    time("body " #B " x8B x" #R " compute only", [&] { hipLaunchKernelGGL((k<B, R, false>), dim3(n_frames / 128), dim3(256), 43000, 0, buf, n_frames, fb); });
    RUN(192, 8) RUN(768, 2) RUN(1536, 1)   // 1536 VALU per chunk: 1.5 / 6 / 12 KB bodies
    RUN(3072, 1) RUN(6144, 1)              // 24 KB, 48 KB bodies (3072 / 6144 VALU per chunk)
    RUN(768, 4) RUN(768, 8)                // same VALU counts from a 6-KB body
    return 0;
}
