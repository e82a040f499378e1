// Study for K1's hard classes (round 4): in how many slices does some candidate's weight energy leave the i16-packed forms'
// window (S >= 3875 << 18 at some step: its arithmetic is no longer exact there, "leaver"), and could such a candidate be
// proven a loser instead of sending the whole slice to the 32-bit step?  What IS exact for a leaver is everything up to and
// including the first step whose energy is outside the window -- that step's penalty is >= 1572 -- so its true rank is at
// least 1572^2 (kPkPenLimit).  Also: how large are the winners' ranks (the speculative forms' saturating 32-bit rank is exact
// below 32767^2), and how often is a scalefactor-0/1 candidate marked (wrapped quotient of the opposite sign).
// Self-contained exhaustive QOA encoder.  Input: raw interleaved int16 PCM.
//   gcc -O2 -o leaver_study leaver_study.c && ./leaver_study pcm.raw channels [label]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const int RECIP[16] = {65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32};
static const int QT[17] = {7, 7, 7, 5, 5, 3, 3, 1, 0, 0, 2, 2, 4, 4, 6, 6, 6};
static const int SFT[16] = {1, 7, 21, 45, 84, 138, 211, 304, 421, 562, 731, 928, 1157, 1419, 1715, 2048};
static int DQ[16][8];
typedef struct { int h[4], w[4]; } lms_t;
static int clamp16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    int ch = atoi(argv[2]);
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    int16_t *pcm = malloc(bytes); if (fread(pcm, 1, bytes, f) != (size_t)bytes) return 2;
    size_t n = bytes / 2 / ch;
    for (int s = 0; s < 16; s++) {
        const double m[4] = {0.75, 2.5, 4.5, 7.0};
        for (int k = 0; k < 4; k++) { int v = (int)(SFT[s] * m[k] + 0.5); DQ[s][2 * k] = v; DQ[s][2 * k + 1] = -v; }
    }
    const int64_t LIMIT = 3875ll << 18, PENLIM = 1572ll * 1572, EXACT = 32767ll * 32767;
    long slices = 0, leaver_slices = 0, leaver_provable = 0, leaver_wins = 0, leaver_provable_prefix = 0;
    long win_sat = 0, marked_slices = 0, marked_unprovable = 0, win_pen = 0, state_out = 0;
    long leaver_by_sf[16] = {0};
    for (int c = 0; c < ch; c++) {
        lms_t st = {{0, 0, 0, 0}, {0, 0, -8192, 16384}};
        for (size_t s0 = 0; s0 + 20 <= n; s0 += 20) {
            uint64_t best = ~0ull, best_first = 0; lms_t best_l = st; int best_sf = -1, best_leaver = 0; uint64_t best_pen = 0;
            int any_leaver = 0, any_marked = 0; uint64_t min_prefix = ~0ull;
            int smax = 0;
            for (int i = 0; i < 20; i++) { int a = abs(pcm[(s0 + i) * ch + c]); if (a > smax) smax = a; }
            { int64_t e = 0; for (int k = 0; k < 4; k++) e += (int64_t)st.w[k] * st.w[k]; if (e >= LIMIT) state_out++; }
            for (int sf = 0; sf < 16; sf++) {
                lms_t l = st; uint64_t rank = 0, first = 0, pens = 0, prefix = 0; int left = 0, marked = 0;
                for (int i = 0; i < 20; i++) {
                    int smp = pcm[(s0 + i) * ch + c];
                    int32_t p = 0; for (int k = 0; k < 4; k++) p = (int32_t)((uint32_t)p + (uint32_t)(l.w[k] * l.h[k]));
                    int pr = p >> 13, v = smp - pr;
                    int64_t wsum = 0; for (int k = 0; k < 4; k++) wsum += (int64_t)l.w[k] * l.w[k];
                    int32_t ws32 = (int32_t)(uint32_t)wsum;
                    int64_t pen = (ws32 >> 18) - 0x8ff; if (pen < 0) pen = 0;
                    int32_t x = (int32_t)((uint32_t)v * (uint32_t)RECIP[sf] + (1u << 15));
                    int nq = x >> 16, a = (v > 0) - (v < 0), sn = (nq > 0) - (nq < 0);
                    if (sn == -a && a != 0) marked = 1;
                    int scaled = nq + a - sn, cl = scaled < -8 ? -8 : scaled > 8 ? 8 : scaled;
                    int dq = DQ[sf][QT[cl + 8]], rc = clamp16(pr + dq);
                    int64_t err = smp - rc;
                    uint64_t t = (uint64_t)(err * err) + (uint64_t)(pen * pen);
                    rank += t; pens += (uint64_t)(pen * pen);
                    if (i == 0) first = t;
                    if (!left) { prefix = rank; if (wsum >= LIMIT) left = 1; }   // exact up to and including the first step outside
                    int d = dq >> 4;
                    for (int k = 0; k < 4; k++) l.w[k] += l.h[k] < 0 ? -d : d;
                    l.h[0] = l.h[1]; l.h[1] = l.h[2]; l.h[2] = l.h[3]; l.h[3] = rc;
                }
                if (left) { any_leaver = 1; leaver_by_sf[sf]++; if (prefix < min_prefix) min_prefix = prefix; }
                if (marked) any_marked = 1;
                if (rank < best || (rank == best && first < best_first)) { best = rank; best_first = first; best_l = l; best_sf = sf; best_leaver = left; best_pen = pens; }
            }
            slices++;
            if (any_leaver) { leaver_slices++; if (best < (uint64_t)PENLIM) leaver_provable++; if (best < min_prefix) leaver_provable_prefix++; if (best_leaver) leaver_wins++; }
            if (best >= (uint64_t)EXACT) win_sat++;
            if (best_pen) win_pen++;
            if (any_marked) { marked_slices++; int lb = 32767 - smax; if (lb > 32719) lb = 32719; if (lb < 0) lb = 0; if (best + 1 > (uint64_t)lb * lb) marked_unprovable++; }
            st = best_l; (void)best_sf;
        }
    }
    printf("%-10s slices %ld: winner rank >= 32767^2 %.3f %%, winner carries a penalty %.2f %%, entry state outside the window %.3f %%\n",
           argc > 3 ? argv[3] : "", slices, 100.0 * win_sat / slices, 100.0 * win_pen / slices, 100.0 * state_out / slices);
    printf("           some candidate leaves the i16 window: %.3f %% of the slices; of those the winner is provably better by the\n"
           "           1572^2 bound in %.1f %%, by the leaver's exact prefix in %.1f %%; a leaver IS the winner in %.2f %%\n",
           100.0 * leaver_slices / slices, leaver_slices ? 100.0 * leaver_provable / leaver_slices : 0.0,
           leaver_slices ? 100.0 * leaver_provable_prefix / leaver_slices : 0.0, leaver_slices ? 100.0 * leaver_wins / leaver_slices : 0.0);
    printf("           leavers by scalefactor:");
    for (int s = 0; s < 16; s++) printf(" %ld", leaver_by_sf[s]);
    printf("\n           a scalefactor-0/1 candidate is marked in %.2f %% of the slices; not settled by the sample bound: %.2f %% of all slices\n",
           100.0 * marked_slices / slices, 100.0 * marked_unprovable / slices);
    return 0;
}
