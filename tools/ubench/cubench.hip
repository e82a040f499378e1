// Is the streaming-store rate bound per CU (L1 -> L2 request path) or chip-wide (HBM)?  One workgroup per CU (96 KB of
// dynamic LDS keeps a second one out), N workgroups, every workgroup fills its own contiguous region with full-line
// 16-byte-per-lane stores; W waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u4 *gptr;
template <bool NT>
__global__ void fill(u4 *out, size_t per_block16) {
    extern __shared__ unsigned s_pad[];
    if (per_block16 == 0) s_pad[threadIdx.x] = 1;
    u4 v = {1, 2, 3, threadIdx.x};
    u4 *base = out + (size_t)blockIdx.x * per_block16;
    for (size_t i = threadIdx.x; i < per_block16; i += blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, (gptr)(base + i)); else base[i] = v;
    }
}
__global__ void copy(const u4 *in, u4 *out, size_t per_block16) {
    extern __shared__ unsigned s_pad[];
    if (per_block16 == 0) s_pad[threadIdx.x] = 1;
    const u4 *src = in + (size_t)blockIdx.x * per_block16;
    u4 *dst = out + (size_t)blockIdx.x * per_block16;
    for (size_t i = threadIdx.x; i < per_block16; i += blockDim.x) dst[i] = src[i];
}
int main() {
    size_t bytes = 4ull << 30;
    u4 *buf, *buf2; (void)hipMalloc(&buf, bytes); (void)hipMalloc(&buf2, bytes);
    (void)hipMemset(buf2, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)fill<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)hipFuncSetAttribute((const void *)fill<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)hipFuncSetAttribute((const void *)copy, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int waves : {4, 8, 16}) {
        for (int nblk : {16, 32, 64, 128, 192, 256}) {
            size_t per_block = (16ull << 20) / 16; // 16 MB per workgroup (256 x 16 MB = the buffer)
            for (int mode = 0; mode < 3; mode++) {
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    (void)hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(fill<false>, dim3(nblk), dim3(64 * waves), 98304, 0, buf, per_block);
                    else if (mode == 1) hipLaunchKernelGGL(fill<true>, dim3(nblk), dim3(64 * waves), 98304, 0, buf, per_block);
                    else hipLaunchKernelGGL(copy, dim3(nblk / 2 ? nblk / 2 : 1), dim3(64 * waves), 98304, 0, buf2, buf, per_block);
                    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                int nb = mode == 2 ? (nblk / 2 ? nblk / 2 : 1) : nblk;
                double gb = (double)nb * per_block * 16 / 1e9 * (mode == 2 ? 2 : 1);
                printf("waves/wg %2d  workgroups %3d  %-10s %.3f ms  %.1f GB/s  %.1f GB/s per CU\n", waves, nb,
                       mode == 0 ? "fill" : mode == 1 ? "fill nt" : "copy r+w", ms, gb / ms * 1e3, gb / ms * 1e3 / nb);
            }
        }
    }
    return 0;
}
