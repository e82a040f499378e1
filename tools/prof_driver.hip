// Standalone profiling driver (no Python, no torch): config-3-shaped batch through the C ABI, for rocprofv3
// PMC passes (rocprofv3 --pmc crashes inside torch's elementwise launches on this image).
//   prof_driver [clips=1024] [frames=256] [channels=2] [enc_runs=2] [dec_runs=2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/qoaudio_amd.h"

__global__ void fill_pcm(int16_t *pcm, size_t n_clips, size_t samples, int ch) {
    size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t total = n_clips * samples;
    for (; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t clip = idx / samples, t = idx % samples;
        unsigned h = (unsigned)(clip * 2654435761u);
        for (int c = 0; c < ch; c++) {
            float f1 = 80.0f + (h % 977) * 3.0f, f2 = 300.0f + ((h >> 8) % 991) * 5.0f;
            float x = 6000.0f * __sinf(t * f1 * (6.2831853f / 44100.0f) + c) + 2500.0f * __sinf(t * f2 * (6.2831853f / 44100.0f) + 2 * c);
            unsigned r = (unsigned)(idx * 2 + c) * 747796405u + 2891336453u;
            r = ((r >> ((r >> 28) + 4)) ^ r) * 277803737u;
            x += ((int)(r >> 22) - 512) * 0.9f;
            pcm[(clip * samples + t) * ch + c] = (int16_t)fminf(fmaxf(x, -32768.f), 32767.f);
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

int main(int argc, char **argv) {
    size_t n_clips = argc > 1 ? atol(argv[1]) : 1024;
    size_t frames = argc > 2 ? atol(argv[2]) : 256;
    int ch = argc > 3 ? atoi(argv[3]) : 2;
    int enc_runs = argc > 4 ? atoi(argv[4]) : 2, dec_runs = argc > 5 ? atoi(argv[5]) : 2;
    unsigned flags = argc > 6 ? (unsigned)atoi(argv[6]) : 0;
    size_t samples = frames * 5120;
    size_t per_pcm = samples * ch, per_enc = qoa_encoded_size(ch, (uint32_t)samples);
    int16_t *pcm, *back; uint8_t *enc;
    CK(hipMalloc(&pcm, n_clips * per_pcm * 2));
    CK(hipMalloc(&back, n_clips * per_pcm * 2));
    CK(hipMalloc(&enc, n_clips * per_enc));
    hipLaunchKernelGGL(fill_pcm, dim3(4096), dim3(256), 0, 0, pcm, n_clips, samples, ch);
    CK(hipDeviceSynchronize());
    std::vector<qoa_clip> clips(n_clips);
    for (size_t i = 0; i < n_clips; i++) { clips[i].pcm_offset = i * per_pcm; clips[i].out_offset = i * per_enc; clips[i].desc = {(uint8_t)ch, 44100, (uint32_t)samples}; }
    qoa_encode_plan *ep; qoa_decode_plan *dp;
    if (qoa_encode_plan_create(clips.data(), n_clips, flags, &ep)) { printf("plan: %s\n", qoa_last_error_detail()); return 1; }
    if (qoa_decode_plan_from_clips(clips.data(), n_clips, &dp)) { printf("dplan: %s\n", qoa_last_error_detail()); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < enc_runs; i++) {
        hipEventRecord(e0, 0); qoa_encode_plan_run(ep, pcm, enc, nullptr); hipEventRecord(e1, 0); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("encode %zu clips x %zu frames x %d ch: %.3f ms  %.2f Gsamples/s\n", n_clips, frames, ch, ms, n_clips * per_pcm / ms / 1e6);
    }
    for (int i = 0; i < dec_runs; i++) {
        hipEventRecord(e0, 0); qoa_decode_plan_run(dp, enc, back, nullptr); hipEventRecord(e1, 0); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("decode: %.3f ms  %.2f Gsamples/s  %.1f GB/s\n", ms, n_clips * per_pcm / ms / 1e6, (n_clips * (per_pcm * 2 + per_enc)) / ms / 1e6);
    }
    uint32_t mm = 0; qoa_decode_plan_mismatches(dp, &mm);
    printf("header mismatches: %u\n", mm);
    return 0;
}
