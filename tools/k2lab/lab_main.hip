// K2 laboratory driver (development aid): times one build variant of the decode kernel on a config-3/2-shaped batch
// of random slice bytes (any bytes are valid slices) and prints a checksum of the PCM so variants can be compared.
//   lab [nch=2] [clips=1024] [frames=256] [runs=5]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT_FILE='"k2_variant.hip"' [-D...] lab_main.hip -o lab_x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include VARIANT_FILE
#undef qoa
#include "../../include/qoaudio_amd.h"
#define qoa qoa_lab

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

__global__ void fill_random(unsigned *p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned r = (unsigned)i * 747796405u + 2891336453u;
        r = ((r >> ((r >> 28) + 4)) ^ r) * 277803737u;
        // keep scalefactors moderate (top nibble of every 8-byte slice's first byte) so weights stay in the packed range
        p[i] = (r ^ (r >> 22));
    }
}
__global__ void fix_sf(unsigned char *p, size_t n8) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) p[i * 8] = (p[i * 8] & 0x7f); // sf 0..7
}
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
__global__ void checksum(const unsigned *p, size_t n, unsigned long long *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * (i % 1000003) + 1);
    atomicAdd(out, s);
}

__global__ void compare(const int16_t *a, const int16_t *b, size_t n, unsigned long long *res) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        if (a[i] != b[i]) { atomicAdd(&res[0], 1ull); atomicMin(&res[1], (unsigned long long)i); }
}

int main(int argc, char **argv) {
    int nch = argc > 1 ? atoi(argv[1]) : 2;
    size_t n_clips = argc > 2 ? atol(argv[2]) : 1024;
    size_t frames = argc > 3 ? atol(argv[3]) : 256;
    int runs = argc > 4 ? atoi(argv[4]) : 5;
    // experiment knob: every clip's frames are read from the encoded bytes of clip (c % in_wrap) -- in_wrap = 2 keeps the whole
    // input resident in each XCD's L2 (the PCM then differs from the product's decode: timing only)
    size_t in_wrap = argc > 5 ? atol(argv[5]) : 0;
    // experiment knob: frames re-laid out at a stride of `pad_stride` bytes (multiple of 128) so that every frame's slice
    // area (frame + 8 + 16 * nch) starts on a 128-byte line: what would whole-line slice loads buy?
    size_t pad_stride = argc > 6 ? atol(argv[6]) : 0;
    // experiment knob: output frames `out_pad` bytes apart instead of contiguous (5120 * nch * 2): is the power-of-two-ish
    // frame stride of the PCM (20480 B for stereo) a channel/bank camping problem?  (timing only)
    size_t out_pad = argc > 7 ? atol(argv[7]) : 0;
    const size_t fsize = 8 + 16 * nch + 8 * 256 * nch;
    const size_t out_pad_early = argc > 7 ? atol(argv[7]) : 0;
    const size_t per_enc = 8 + frames * fsize, per_pcm = frames * (5120 * nch + out_pad_early / 2);
    unsigned char *enc; int16_t *pcm; unsigned *mm; unsigned char *scratch; unsigned long long *cs;
    size_t enc_bytes = (n_clips * per_enc + 15) / 16 * 16;
    CK(hipMalloc(&enc, enc_bytes)); CK(hipMalloc(&pcm, n_clips * per_pcm * 2)); CK(hipMalloc(&mm, 4));
    CK(hipMalloc(&scratch, 65536)); CK(hipMalloc(&cs, 8));
    {   // realistic slices: encode synthetic PCM with the product library (K1), as tools/prof_driver does
        hipLaunchKernelGGL(fill_pcm, dim3(4096), dim3(256), 0, 0, pcm, n_clips, frames * 5120, nch);
        CK(hipDeviceSynchronize());
        std::vector<qoa_clip> clips(n_clips);
        for (size_t i = 0; i < n_clips; i++) { clips[i].pcm_offset = i * per_pcm; clips[i].out_offset = i * per_enc; clips[i].desc = {(uint8_t)nch, 44100, (uint32_t)(frames * 5120)}; }
        qoa_encode_plan *ep;
        if (qoa_encode_plan_create(clips.data(), n_clips, 0, &ep)) { printf("plan: %s\n", qoa_last_error_detail()); return 1; }
        if (qoa_encode_plan_run(ep, pcm, enc, nullptr)) { printf("run: %s\n", qoa_last_error_detail()); return 1; }
        CK(hipDeviceSynchronize());
        qoa_encode_plan_free(ep);
        CK(hipMemset(pcm, 0, n_clips * per_pcm * 2));
    }
    unsigned char *enc_pad = nullptr;
    const size_t lead = 128 - (8 + 16 * nch) % 128; // frame starts `lead` bytes into a line
    if (pad_stride) {
        CK(hipMalloc(&enc_pad, n_clips * frames * pad_stride + 256));
        for (size_t c = 0; c < n_clips; c++)
            CK(hipMemcpy2DAsync(enc_pad + lead + c * frames * pad_stride, pad_stride, enc + c * per_enc + 8, fsize, fsize, frames,
                                hipMemcpyDeviceToDevice, 0));
        CK(hipDeviceSynchronize());
    }
    std::vector<qoa::DecChain> tab(n_clips * frames);
    for (size_t c = 0; c < n_clips; c++)
        for (size_t f = 0; f < frames; f++) {
            qoa::DecChain &d = tab[c * frames + f];
            d.frame_off = (in_wrap ? c % in_wrap : c) * per_enc + 8 + f * fsize;
            if (pad_stride) d.frame_off = lead + ((in_wrap ? c % in_wrap : c) * frames + f) * pad_stride; d.out_off = c * per_pcm + f * (5120 * nch + out_pad / 2); d.n_out = 5120; d.rows = 256;
            d.channels = (uint8_t)nch; d.channel = 0; d.expect_hi = d.expect_lo = 0;
        }
    qoa::DecChain *dtab; CK(hipMalloc(&dtab, tab.size() * sizeof(qoa::DecChain)));
    CK(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(qoa::DecChain), hipMemcpyHostToDevice));
    CK(hipMemset(mm, 0, 4));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    for (int i = 0; i < runs + 1; i++) {
        (void)hipEventRecord(e0, 0);
#ifdef LAB_UNIFORM
        CK(qoa::launch_decode_uniform(nch, pad_stride ? enc_pad : enc, pcm, dtab, (unsigned)tab.size(), mm, 0));
#else
        CK(qoa::launch_decode_frames(nch, true, enc, pcm, dtab, (unsigned)tab.size(), mm, scratch, 0));
#endif
        (void)hipEventRecord(e1, 0); CK(hipDeviceSynchronize());
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i) { sum += ms; if (ms < best) best = ms; }
    }
    CK(hipMemset(cs, 0, 8));
    hipLaunchKernelGGL(checksum, dim3(4096), dim3(256), 0, 0, (const unsigned *)pcm, n_clips * per_pcm / 2, cs);
    unsigned long long h; CK(hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost));
    {   // element-wise check against the product library's decode of the same bytes
        std::vector<qoa_clip> clips(n_clips);
        for (size_t i = 0; i < n_clips; i++) { clips[i].pcm_offset = i * per_pcm; clips[i].out_offset = i * per_enc; clips[i].desc = {(uint8_t)nch, 44100, (uint32_t)(frames * 5120)}; }
        qoa_decode_plan *dp; int16_t *ref; unsigned long long *res;
        CK(hipMalloc(&ref, n_clips * per_pcm * 2)); CK(hipMalloc(&res, 16));
        unsigned long long init[2] = {0, ~0ull}; CK(hipMemcpy(res, init, 16, hipMemcpyHostToDevice));
        if (qoa_decode_plan_from_clips(clips.data(), n_clips, &dp)) { printf("dplan: %s\n", qoa_last_error_detail()); return 1; }
        if (qoa_decode_plan_run(dp, enc, ref, nullptr)) { printf("drun: %s\n", qoa_last_error_detail()); return 1; }
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(compare, dim3(4096), dim3(256), 0, 0, pcm, ref, n_clips * per_pcm, res);
        unsigned long long r[2]; CK(hipMemcpy(r, res, 16, hipMemcpyDeviceToHost));
        if (r[0]) {
            size_t i = r[1], clip = i / per_pcm, rem = i % per_pcm, fr = rem / (5120 * nch), s = (rem % (5120 * nch)) / nch, ch = rem % nch;
            printf("  MISMATCH vs product decode: %llu samples differ, first at clip %zu frame %zu sample %zu (row %zu, i %zu) ch %zu\n", r[0], clip, fr, s, s / 20, s % 20, ch);
            std::vector<int16_t> a(64), b(64); size_t base = i - i % (20 * nch);
            CK(hipMemcpy(a.data(), pcm + base, 128, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref + base, 128, hipMemcpyDeviceToHost));
            printf("  got :"); for (int k = 0; k < 44; k++) printf(" %d", a[k]); printf("\n  want:"); for (int k = 0; k < 44; k++) printf(" %d", b[k]); printf("\n");
        } else printf("  matches the product decode\n");
    }
    double bytes = (double)n_clips * (per_pcm * 2 + per_enc);
    printf("%-28s nch=%d avg %.3f ms  best %.3f ms  %.2f TB/s (best)  frac %.3f  checksum %016llx\n", LABEL, nch, sum / runs, best,
           bytes / best / 1e9, bytes / best / 1e9 / 8.0, h);
    return 0;
}
