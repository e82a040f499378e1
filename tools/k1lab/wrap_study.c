// Study for K1: how often does a scalefactor-0/1 candidate hit a qoa_div whose wrapped quotient has the OPPOSITE sign of
// the residual (the only case in which "level from |n|, sign from the residual" differs from the reference's
// n + sgn(v) - sgn(n))?  Self-contained QOA encoder (exhaustive search), input: a .qoa file (decoded first) or synthetic.
//   gcc -O2 -o wrap_study wrap_study.c -lm && ./wrap_study file.qoa
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const int RECIP[16] = {65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32};
static const int QT[17] = {7, 7, 7, 5, 5, 3, 3, 1, 0, 0, 2, 2, 4, 4, 6, 6, 6};
static const int SFT[16] = {1, 7, 21, 45, 84, 138, 211, 304, 421, 562, 731, 928, 1157, 1419, 1715, 2048};
static int DQ[16][8];
typedef struct { int h[4], w[4]; } lms_t;

static int predict(const lms_t *l) {
    int32_t p = 0;
    for (int i = 0; i < 4; i++) p = (int32_t)((uint32_t)p + (uint32_t)(l->w[i] * l->h[i]));
    return p >> 13;
}
static void update(lms_t *l, int s, int r) {
    int d = r >> 4;
    for (int i = 0; i < 4; i++) l->w[i] += l->h[i] < 0 ? -d : d;
    for (int i = 0; i < 3; i++) l->h[i] = l->h[i + 1];
    l->h[3] = s;
}
static int clamp16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

static long n_slices, flip_slices[2], wrap_slices[2], flip_and_clamp[2], winner_flipped, any_flip;

static void encode_chain(const int16_t *pcm, size_t n, lms_t *st) {
    for (size_t s0 = 0; s0 + 20 <= n; s0 += 20) {
        uint64_t best = ~0ull; lms_t best_l = *st; int best_sf = -1, best_flip = 0;
        int slice_any = 0;
        for (int sf = 0; sf < 16; sf++) {
            lms_t l = *st; uint64_t rank = 0; int flip = 0, wrap = 0, fc = 0;
            for (int i = 0; i < 20; i++) {
                int smp = pcm[s0 + i], pr = predict(&l), v = smp - pr;
                int64_t wsum = 0; for (int k = 0; k < 4; k++) wsum += (int64_t)l.w[k] * l.w[k];
                int64_t pen = (wsum >> 18) - 0x8ff; if (pen < 0) pen = 0;
                int32_t x = (int32_t)((uint32_t)v * (uint32_t)RECIP[sf] + (1u << 15));
                int nq = x >> 16;
                int64_t exact = (int64_t)v * RECIP[sf] + (1 << 15);
                int a = (v > 0) - (v < 0), sn = (nq > 0) - (nq < 0);
                int scaled = nq + a - sn;
                if (exact != (int64_t)x) wrap = 1;
                int clamped = scaled < -8 ? -8 : scaled > 8 ? 8 : scaled;
                int q = QT[clamped + 8], dq = DQ[sf][q];
                int rc = clamp16(pr + dq);
                if (sn == -a && a != 0) { flip = 1; if (rc != pr + dq) fc = 1; }
                int64_t err = smp - rc;
                rank += (uint64_t)(err * err) + (uint64_t)(pen * pen);
                update(&l, rc, dq);
            }
            if (sf < 2) { flip_slices[sf] += flip; wrap_slices[sf] += wrap; flip_and_clamp[sf] += fc; }
            slice_any |= flip;
            if (rank < best) { best = rank; best_l = l; best_sf = sf; best_flip = flip; }
        }
        (void)best_sf;
        winner_flipped += best_flip; any_flip += slice_any;
        n_slices++;
        *st = best_l;
    }
}

static int16_t *decode_qoa(const uint8_t *d, size_t len, int *ch, size_t *samples) {
    uint32_t total = (d[4] << 24) | (d[5] << 16) | (d[6] << 8) | d[7];
    size_t p = 8; *ch = d[8]; int16_t *out = malloc((size_t)total * *ch * 2); size_t done = 0;
    while (p + 8 <= len && done < total) {
        int c = d[p]; unsigned fs = (d[p + 4] << 8) | d[p + 5]; p += 8;
        lms_t l[8];
        for (int k = 0; k < c; k++) {
            for (int i = 0; i < 4; i++) l[k].h[i] = (int16_t)((d[p + 2 * i] << 8) | d[p + 2 * i + 1]);
            for (int i = 0; i < 4; i++) l[k].w[i] = (int16_t)((d[p + 8 + 2 * i] << 8) | d[p + 8 + 2 * i + 1]);
            p += 16;
        }
        for (unsigned s = 0; s < fs; s += 20) for (int k = 0; k < c; k++) {
            uint64_t sl = 0; for (int i = 0; i < 8; i++) sl = (sl << 8) | d[p + i]; p += 8;
            int sf = sl >> 60; sl <<= 4;
            for (unsigned i = s; i < s + 20 && i < fs; i++) {
                int pr = predict(&l[k]), q = (sl >> 61) & 7, dq = DQ[sf][q]; sl <<= 3;
                int rc = clamp16(pr + dq); out[(done + i) * c + k] = rc; update(&l[k], rc, dq);
            }
        }
        done += fs;
    }
    *samples = done; return out;
}

int main(int argc, char **argv) {
    for (int s = 0; s < 16; s++) {
        static const double B[4] = {0.75, 2.5, 4.5, 7.0};
        for (int k = 0; k < 4; k++) { int m = (int)floor(SFT[s] * B[k] + 0.5); DQ[s][2 * k] = m; DQ[s][2 * k + 1] = -m; }
    }
    if (argc > 1 && strcmp(argv[1], "synth")) {
        FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long len = ftell(f); rewind(f);
        uint8_t *d = malloc(len); if (fread(d, 1, len, f) != (size_t)len) return 1; fclose(f);
        int ch; size_t n; int16_t *pcm = decode_qoa(d, len, &ch, &n);
        double gain = argc > 2 ? atof(argv[2]) : 1.0;
        for (int k = 0; k < ch; k++) {
            int16_t *c = malloc(n * 2); for (size_t i = 0; i < n; i++) c[i] = clamp16((int)lrint(pcm[i * ch + k] * gain));
            lms_t st = {{0, 0, 0, 0}, {0, 0, -(1 << 13), 1 << 14}};
            encode_chain(c, n, &st); free(c);
        }
    } else {
        size_t n = 44100 * 20; int16_t *c = malloc(n * 2);
        double amp = argc > 2 ? atof(argv[2]) : 0.5; unsigned r = 12345;
        for (size_t i = 0; i < n; i++) {
            double t = i / 44100.0, v = 0;
            for (int h = 1; h <= 6; h++) v += sin(2 * M_PI * 220.0 * h * t * (1 + 0.01 * h)) / h;
            r = r * 1664525u + 1013904223u; v += ((r >> 8) / 16777216.0 - 0.5) * 0.3;
            c[i] = clamp16((int)lrint(v * amp * 16000));
        }
        lms_t st = {{0, 0, 0, 0}, {0, 0, -(1 << 13), 1 << 14}};
        encode_chain(c, n, &st);
    }
    printf("slices %ld\n", n_slices);
    for (int sf = 0; sf < 2; sf++)
        printf("  sf %d: slices with a 32-bit wrap %ld (%.3f %%), with a sign-flipped quotient %ld (%.3f %%), flip at a clamped reconstruction %ld\n",
               sf, wrap_slices[sf], 100.0 * wrap_slices[sf] / n_slices, flip_slices[sf], 100.0 * flip_slices[sf] / n_slices, flip_and_clamp[sf]);
    printf("  slices with a flip in any candidate %ld (%.3f %%); winner itself flipped %ld\n", any_flip, 100.0 * any_flip / n_slices, winner_flipped);
    return 0;
}
