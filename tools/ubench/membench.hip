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
__global__ void fill_decode_like(unsigned char *out, unsigned n_frames, unsigned frame_bytes, unsigned piece_bytes, int spin, int inflight = 0, int lds_mix = 0) {
    extern __shared__ unsigned s_pad[]; // dynamic LDS only limits the occupancy (decode kernel: 12 waves per CU)
    if (n_frames == 0xffffffffu) s_pad[threadIdx.x] = 1;
    unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned ppf = piece_bytes / 16;             // 16-B pieces per frame per chunk
    unsigned fpw = 32;
    unsigned np = fpw * ppf / 64;
    unsigned chunks = frame_bytes / piece_bytes;
    size_t base = (size_t)wave * fpw * frame_bytes;
    if ((wave + 1) * fpw > n_frames) return;
    u4 v = {1, 2, 3, lane};
    for (unsigned c = 0; c < chunks; c++) {
        for (int s = 0; s < spin; s += 16) { // 16 VALU per trip on 4 independent registers
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1"
                         : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            if (lds_mix) { // the decode kernel's LDS share: ~1 ds_write_b16 + 1 ds_read_b32 per 13 VALU
                unsigned a = (threadIdx.x * 84u + (s & 1023)) & 0x7ffcu, r0;
                asm volatile("ds_write_b16 %1, %2\n ds_read_b32 %0, %3\n s_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(a), "v"(v.x), "v"((threadIdx.x & 127u) * 4u) : "memory");
                v.y ^= r0 & 1;
            }
        }
        for (unsigned t = 0; t < np; t++) {
            unsigned p = lane + 64 * t, g = p / ppf, k = p % ppf;
            gptr dst = (gptr)(out + base + (size_t)g * frame_bytes + (size_t)(inflight == 2 ? 0 : c) * piece_bytes + k * 16);
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
        // the decode kernel lets at most two flushes of stores be in flight per wave (counted vmcnt waits)
        if (inflight == 1) { if (np == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (np == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else if (np == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (np == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
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
    for (unsigned piece : {128u, 256u, 320u, 640u, 1280u, 5120u}) {
        for (int spin : {0}) {
            char name[96];
            snprintf(name, 96, "decode-like piece=%u spin=%d plain", piece, spin);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<false>, dim3(n_frames / 32 / 4), dim3(256), 0, 0, buf, n_frames, fb, piece, spin); });
            snprintf(name, 96, "decode-like piece=%u spin=%d nt", piece, spin);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 0, 0, buf, n_frames, fb, piece, spin); });
        }
    }
    // the same patterns at the decode kernel's occupancy (12 waves per CU: 42 KB of LDS per 4-wave block) and in-flight limit
    for (unsigned piece : {128u, 256u, 320u, 640u}) {
        for (int infl : {0, 1}) {
            char name[96];
            snprintf(name, 96, "occ12 piece=%u inflight-limit=%d nt", piece, infl);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 43000, 0, buf, n_frames, fb, piece, 0, infl); });
            snprintf(name, 96, "occ12 piece=%u inflight-limit=%d plain", piece, infl);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<false>, dim3(n_frames / 32 / 4), dim3(256), 43000, 0, buf, n_frames, fb, piece, 0, infl); });
        }
    }
    // does VALU work overlap with the stores?  occ12, in-flight limit, spin VALU instructions per chunk per wave
    // (3 waves per SIMD x 64 chunks x spin x 4 cycles: spin 3072 = 2.36 M cycles = ~1 ms of pure issue)
    for (unsigned piece : {256u, 320u}) {
        for (int spin : {0, 768, 1536, 3072, 4608}) {
            char name[96];
            snprintf(name, 96, "occ12 piece=%u valu-per-chunk=%d nt", piece, spin);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 43000, 0, buf, n_frames, fb, piece, spin, 1); });
        }
    }
    for (int spin : {1536, 3072}) {
        char name[96];
        snprintf(name, 96, "occ12 piece=320 valu-per-chunk=%d + LDS mix nt", spin);
        time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 43000, 0, buf, n_frames, fb, 320u, spin, 1, 1); });
        snprintf(name, 96, "occ12 valu-per-chunk=%d + LDS mix, stores to one chunk only", spin);
        time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 43000, 0, buf, n_frames, fb, 320u, spin, 2, 1); });
    }
    // compute only: the same VALU work, one chunk's stores per wave instead of 64 or 80
    {
        char name[96];
        for (int spin : {3072, 4608}) {
            snprintf(name, 96, "occ12 valu-per-chunk=%d, stores to one chunk only", spin);
            time(name, [&] { hipLaunchKernelGGL(fill_decode_like<true>, dim3(n_frames / 32 / 4), dim3(256), 43000, 0, buf, n_frames, fb, 320u, spin, 2); });
        }
    }
    return 0;
}
