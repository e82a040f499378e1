// Write-bandwidth probes for the decode output pattern (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u4 *gptr;

template <bool NT>
__global__ void fill_seq(u4 *out, size_t n16) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    u4 v = {1, 2, 3, (unsigned)i};
    for (; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(v, (gptr)(out + i)); else out[i] = v;
    }
}
// decode-like: wave handles 32 frames of FRAME bytes; per chunk each lane stores 10 pieces of 16 B:
// piece p = lane + 64 t -> frame p/20, offset (p%20)*16 within the frame's 320-byte chunk.
template <bool NT>
__global__ void fill_decode_like(unsigned char *out, unsigned n_frames, unsigned frame_bytes, unsigned piece_bytes, int spin) {
    unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned ppf = piece_bytes / 16;             // 16-B pieces per frame per chunk
    unsigned fpw = 32;
    unsigned np = fpw * ppf / 64;
    unsigned chunks = frame_bytes / piece_bytes;
    size_t base = (size_t)wave * fpw * frame_bytes;
    if ((wave + 1) * fpw > n_frames) return;
    u4 v = {1, 2, 3, lane};
    for (unsigned c = 0; c < chunks; c++) {
        for (int s = 0; s < spin; s++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v.x));
        for (unsigned t = 0; t < np; t++) {
            unsigned p = lane + 64 * t, g = p / ppf, k = p % ppf;
            gptr dst = (gptr)(out + base + (size_t)g * frame_bytes + (size_t)c * piece_bytes + k * 16);
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}
int main() {
    size_t bytes = 5368709120ull;
    unsigned char *buf; hipMalloc(&buf, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    time("seq fill plain", [&] { hipLaunchKernelGGL(fill_seq<false>, dim3(8192), dim3(256), 0, 0, (u4 *)buf, bytes / 16); });
    time("seq fill nt", [&] { hipLaunchKernelGGL(fill_seq<true>, dim3(8192), dim3(256), 0, 0, (u4 *)buf, bytes / 16); });
    unsigned n_frames = 262144, fb = 20480;
    for (unsigned piece : {320u, 640u, 1280u, 5120u, 20480u}) {
        for (int spin : {0, 1500}) {
            char name[96];
            snprintf(name, 96, "decode-like piece=%u spin=%d plain", piece, spin);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<false>, dim3(n_frames / 32 / 4), dim3(256), 0, 0, buf, n_frames, fb, piece, spin); });
            snprintf(name, 96, "decode-like piece=%u spin=%d nt", piece, spin);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 0, 0, buf, n_frames, fb, piece, spin); });
        }
    }
    return 0;
}
