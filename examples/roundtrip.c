/* Plain-C use of the C ABI (include/qoaudio_amd.h): encode_all + decode_all on host buffers, the way the crate's
 * examples/encode.rs and examples/decode.rs use the Rust API (there with WAV files; `python -m qoaudio_amd.cli` is the
 * WAV front end here).
 *   gcc -std=c99 -Iinclude examples/roundtrip.c -Lqoaudio_amd -lqoaudio_amd -Wl,-rpath,$PWD/qoaudio_amd -lm -o roundtrip */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qoaudio_amd.h"

int main(void) {
    const qoa_desc desc = {2, 44100, 44100 * 2}; /* channels, sample_rate, samples per channel */
    const size_t n = (size_t)desc.samples * desc.channels;
    int16_t *pcm = malloc(n * sizeof *pcm), *back = malloc(n * sizeof *back);
    uint8_t *enc = malloc(qoa_encoded_size(desc.channels, desc.samples));
    if (!pcm || !back || !enc) return 1;
    for (uint32_t t = 0; t < desc.samples; t++) {
        pcm[2 * t] = (int16_t)(9000.0 * sin(t * 0.031) + 2000.0 * sin(t * 0.4));
        pcm[2 * t + 1] = (int16_t)(7000.0 * sin(t * 0.047 + 1.0));
    }
    size_t written = 0, n_out = 0;
    int rc = qoa_encode_all(pcm, n, &desc, enc, qoa_encoded_size(desc.channels, desc.samples), &written);
    if (rc) {
        fprintf(stderr, "encode: %s (%s)\n", qoa_encode_strerror(rc), qoa_last_error_detail());
        return 2;
    }
    uint8_t channels = 0;
    uint32_t rate = 0;
    rc = qoa_decode_all(enc, written, &channels, &rate, back, n, &n_out);
    if (rc) {
        fprintf(stderr, "decode: %s (%s)\n", qoa_decode_strerror(rc), qoa_last_error_detail());
        return 3;
    }
    double err2 = 0;
    int max = 0;
    for (size_t i = 0; i < n; i++) {
        int d = abs(pcm[i] - back[i]);
        err2 += (double)d * d;
        if (d > max) max = d;
    }
    printf("%s: %zu samples -> %zu bytes (%.2f:1) -> %zu samples, %u ch, %u Hz; rms error %.1f, max %d\n", qoa_version(), n,
           written, (double)(n * 2) / (double)written, n_out, channels, rate, sqrt(err2 / (double)n), max);
    free(pcm); free(back); free(enc);
    return n_out == n && channels == desc.channels && rate == desc.sample_rate ? 0 : 4;
}
