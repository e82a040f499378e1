/*
 * qoa_oracle.c -- CPU restatement of rafaelcaricio/qoaudio's QOA encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see qoa_oracle.h).  Plain C11, no dependencies, all
 * arithmetic that the reference does with wrapping_* is done here in uint32_t so
 * it is defined behaviour and UBSan-clean.
 *
 * Pinning: oracle/README.md.  In short: encode(decode(fixture)) reproduces the
 * reference fixture byte-for-byte for its first 1,073,768 bytes (259 frames),
 * tests/test_oracle_golden.py.
 */
#include "qoa_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- tables */

/* src/lib.rs:22 */
static const int32_t QUANT_TAB[17] = {7, 7, 7, 5, 5, 3, 3, 1, 0, 0, 2, 2, 4, 4, 6, 6, 6};

/* src/lib.rs:25-27 */
static const int32_t RECIPROCAL_TAB[16] = {65536, 9363, 3121, 1457, 781, 475, 311, 216,
                                           156,   117,  90,   71,   57,  47,  39,  32};

/* src/lib.rs:847-864 */
static const int32_t DEQUANT_TAB[16][8] = {
    {1, -1, 3, -3, 5, -5, 7, -7},
    {5, -5, 18, -18, 32, -32, 49, -49},
    {16, -16, 53, -53, 95, -95, 147, -147},
    {34, -34, 113, -113, 203, -203, 315, -315},
    {63, -63, 210, -210, 378, -378, 588, -588},
    {104, -104, 345, -345, 621, -621, 966, -966},
    {158, -158, 528, -528, 950, -950, 1477, -1477},
    {228, -228, 760, -760, 1368, -1368, 2128, -2128},
    {316, -316, 1053, -1053, 1895, -1895, 2947, -2947},
    {422, -422, 1405, -1405, 2529, -2529, 3934, -3934},
    {548, -548, 1828, -1828, 3290, -3290, 5117, -5117},
    {696, -696, 2320, -2320, 4176, -4176, 6496, -6496},
    {868, -868, 2893, -2893, 5207, -5207, 8099, -8099},
    {1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933},
    {1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005},
    {1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336},
};

/* ---------------------------------------------------------------- helpers */

static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
/* arithmetic shift right of a possibly negative i32 (gcc/clang: sign-propagating) */
static inline int32_t asr(int32_t a, int n) { return a >> n; }
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int32_t sgn(int32_t v) { return (v > 0) - (v < 0); }

static inline void put_be16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static inline void put_be32(uint8_t *p, uint32_t v) { put_be16(p, (uint16_t)(v >> 16)); put_be16(p + 2, (uint16_t)v); }
static inline void put_be64(uint8_t *p, uint64_t v) { put_be32(p, (uint32_t)(v >> 32)); put_be32(p + 4, (uint32_t)v); }
static inline uint64_t get_be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}

/* src/lib.rs:613-617 */
int32_t oq_div(int32_t v, int sf) {
    int32_t n = asr(wadd(wmul(v, RECIPROCAL_TAB[sf]), 1 << 15), 16);
    return n + sgn(v) - sgn(n);
}

/* src/lib.rs:799-805 */
static inline int32_t lms_predict(const oq_lms *l) {
    int32_t p01 = wadd(wmul(l->weights[0], l->history[0]), wmul(l->weights[1], l->history[1]));
    int32_t p23 = wadd(wmul(l->weights[2], l->history[2]), wmul(l->weights[3], l->history[3]));
    return asr(wadd(p01, p23), 13);
}

/* src/lib.rs:808-818 -- returns the prediction, stores penalty^2 */
static inline int32_t lms_predict_and_penalty_sq(const oq_lms *l, uint64_t *penalty_sq) {
    const int32_t *w = l->weights;
    int32_t s01 = wadd(wmul(w[0], w[0]), wmul(w[1], w[1]));
    int32_t s23 = wadd(wmul(w[2], w[2]), wmul(w[3], w[3]));
    int64_t penalty = (int64_t)asr(wadd(s01, s23), 18) - 0x8ff;
    if (penalty < 0) penalty = 0;
    *penalty_sq = (uint64_t)(penalty * penalty);
    return lms_predict(l);
}

/* src/lib.rs:821-828 */
static inline void lms_update(oq_lms *l, int16_t sample, int32_t residual) {
    int32_t delta = asr(residual, 4);
    for (int i = 0; i < OQ_LMS_LEN; i++)
        l->weights[i] = wadd(l->weights[i], l->history[i] < 0 ? -delta : delta);
    l->history[0] = l->history[1];
    l->history[1] = l->history[2];
    l->history[2] = l->history[3];
    l->history[3] = (int32_t)sample;
}

/* src/lib.rs:602-604 */
uint16_t oq_frame_size(uint32_t channels, uint32_t slices) {
    return (uint16_t)(8 + OQ_LMS_LEN * 4 * channels + 8 * slices * channels);
}

/* src/lib.rs:374-378 */
size_t oq_encoded_size(uint32_t channels, uint32_t samples) {
    size_t total = samples;
    size_t num_slices = (total + OQ_SLICE_LEN - 1) / OQ_SLICE_LEN;
    size_t num_frames = (total + OQ_FRAME_LEN - 1) / OQ_FRAME_LEN;
    return OQ_HEADER_SIZE + num_frames * (8 + OQ_LMS_LEN * 4 * (size_t)channels) + num_slices * 8 * (size_t)channels;
}

/* ---------------------------------------------------------------- encoder */

/* One candidate-sample step shared by both search forms: src/lib.rs:566-584 (and :525-533 for i==0).
 * Returns err^2 + penalty^2; advances lms; appends the 3-bit code to *slice. */
static inline uint64_t candidate_step(oq_lms *lms, int32_t sample, int sf, uint64_t *slice) {
    uint64_t penalty_sq;
    int32_t predicted = lms_predict_and_penalty_sq(lms, &penalty_sq);
    int32_t residual = sample - predicted;
    int32_t scaled = oq_div(residual, sf);
    int32_t clamped = clampi(scaled, -8, 8);
    int32_t quantized = QUANT_TAB[clamped + 8];
    int32_t dequantized = DEQUANT_TAB[sf][quantized];
    int32_t reconstructed = clampi(predicted + dequantized, -32768, 32767);
    int64_t error = (int64_t)sample - reconstructed;
    lms_update(lms, (int16_t)reconstructed, dequantized);
    *slice = (*slice << 3) | (uint64_t)quantized;
    return (uint64_t)(error * error) + penalty_sq;
}

/* src/lib.rs:495-596 */
uint64_t oq_encode_slice(const oq_lms *entry, const int16_t *pcm, size_t start, size_t end,
                         size_t stride, int search, int *best_sf_out, oq_lms *best_lms_out,
                         uint64_t *total_rank_out, uint64_t *steps) {
    int32_t samples[OQ_SLICE_LEN];
    int slice_len = 0;
    for (size_t si = start; si < end && slice_len < OQ_SLICE_LEN; si += stride) /* :502-508 */
        samples[slice_len++] = pcm[si];

    uint64_t best_rank = UINT64_MAX, best_slice = 0, nsteps = 0;
    int best_sf = 0;
    oq_lms best_lms;
    memset(&best_lms, 0, sizeof best_lms);

    if (total_rank_out) { /* exhaustive per-candidate totals for known-answer tests */
        for (int sf = 0; sf < 16; sf++) {
            oq_lms l = *entry;
            uint64_t s = (uint64_t)sf, r = 0;
            for (int i = 0; i < slice_len; i++) r += candidate_step(&l, samples[i], sf, &s);
            total_rank_out[sf] = r;
        }
    }

    if (search == OQ_SEARCH_EXHAUSTIVE) {
        /* Closed form of :516-593: winner = argmin (total rank, first-sample rank, sf). */
        uint64_t best_first = UINT64_MAX;
        for (int sf = 0; sf < 16; sf++) {
            oq_lms l = *entry;
            uint64_t s = (uint64_t)sf, total = 0, first = 0;
            for (int i = 0; i < slice_len; i++) {
                uint64_t r = candidate_step(&l, samples[i], sf, &s);
                if (i == 0) first = r;
                total += r;
                nsteps++;
            }
            if (total < best_rank || (total == best_rank && first < best_first)) {
                best_rank = total; best_first = first; best_slice = s; best_sf = sf; best_lms = l;
            }
        }
    } else {
        /* :516-542 first-sample pass over all 16 scalefactors + stable insertion sort by rank */
        uint64_t first_rank[16], first_slice[16];
        oq_lms first_lms[16];
        uint8_t order[16];
        int count = 0;
        for (int sf = 0; sf < 16; sf++) {
            first_lms[sf] = *entry;
            first_slice[sf] = (uint64_t)sf;
            first_rank[sf] = candidate_step(&first_lms[sf], samples[0], sf, &first_slice[sf]);
            nsteps++;
            int pos = count;
            while (pos > 0 && first_rank[order[pos - 1]] > first_rank[sf]) {
                order[pos] = order[pos - 1];
                pos--;
            }
            order[pos] = (uint8_t)sf;
            count++;
        }
        /* :544-593 candidates in that order, with both early exits */
        for (int k = 0; k < 16; k++) {
            int sf = order[k];
            uint64_t rank = first_rank[sf];
            if (rank > best_rank) break; /* :555-557 */
            oq_lms l = first_lms[sf];
            uint64_t s = first_slice[sf];
            int valid = 1;
            for (int i = 1; i < slice_len; i++) {
                /* the reference adds the rank and tests before it updates; the update of a
                 * rejected candidate is never observed, so step-then-test is equivalent */
                rank += candidate_step(&l, samples[i], sf, &s);
                nsteps++;
                if (rank > best_rank) { valid = 0; break; } /* :578-581 */
            }
            if (valid && rank < best_rank) { /* :587-592 */
                best_rank = rank; best_slice = s; best_sf = sf; best_lms = l;
            }
        }
    }
    if (best_sf_out) *best_sf_out = best_sf;
    if (best_lms_out) *best_lms_out = best_lms;
    if (steps) *steps += nsteps;
    return best_slice;
}

/* Instrumentation for the pin report (oracle/README.md): every candidate of one slice run to the end, with the two
 * terms of its rank (src/lib.rs:576) kept apart -- sum of err^2, sum of penalty^2 -- and the same for its first sample. */
void oq_slice_rank_terms(const oq_lms *entry, const int16_t *pcm, size_t start, size_t end, size_t stride,
                         uint64_t err_sq[16], uint64_t pen_sq[16], uint64_t first_err_sq[16], uint64_t first_pen_sq[16]) {
    int32_t samples[OQ_SLICE_LEN];
    int slice_len = 0;
    for (size_t si = start; si < end && slice_len < OQ_SLICE_LEN; si += stride) samples[slice_len++] = pcm[si];
    for (int sf = 0; sf < 16; sf++) {
        oq_lms l = *entry;
        uint64_t s = (uint64_t)sf;
        err_sq[sf] = pen_sq[sf] = first_err_sq[sf] = first_pen_sq[sf] = 0;
        for (int i = 0; i < slice_len; i++) {
            uint64_t pen;
            (void)lms_predict_and_penalty_sq(&l, &pen);
            uint64_t r = candidate_step(&l, samples[i], sf, &s);
            err_sq[sf] += r - pen;
            pen_sq[sf] += pen;
            if (i == 0) { first_err_sq[sf] = r - pen; first_pen_sq[sf] = pen; }
        }
    }
}

/* Instrumentation for the whole-fixture pins: the slice word every candidate produces when run to the end (sf in the top
 * 4 bits after slice_len shifts, src/lib.rs:549, :584; NOT yet left-aligned for a short tail, :485-487). */
void oq_slice_candidates(const oq_lms *entry, const int16_t *pcm, size_t start, size_t end, size_t stride,
                         uint64_t words[16]) {
    int32_t samples[OQ_SLICE_LEN];
    int slice_len = 0;
    for (size_t si = start; si < end && slice_len < OQ_SLICE_LEN; si += stride) samples[slice_len++] = pcm[si];
    for (int sf = 0; sf < 16; sf++) {
        oq_lms l = *entry;
        uint64_t s = (uint64_t)sf;
        for (int i = 0; i < slice_len; i++) (void)candidate_step(&l, samples[i], sf, &s);
        words[sf] = s;
    }
}

/* src/lib.rs:335-361 */
int oq_encoder_init(oq_encoder *e, const oq_desc *d, int search) {
    if (d->channels == 0 || d->channels > OQ_MAX_CHANNELS) return OQ_ENC_INVALID_CHANNELS;
    if (d->sample_rate == 0) return OQ_ENC_INVALID_SAMPLE_RATE;
    if (d->samples == 0) return OQ_ENC_INVALID_SAMPLES;
    memset(e, 0, sizeof *e);
    e->desc = *d;
    e->search = search;
    for (int c = 0; c < d->channels; c++) {
        e->lms[c].weights[2] = -(1 << 13);
        e->lms[c].weights[3] = 1 << 14;
    }
    return OQ_OK;
}

/* src/lib.rs:441-493 */
static size_t encode_frame_to_buf(oq_encoder *e, const int16_t *pcm, size_t n_i16, uint8_t *buf) {
    size_t channels = e->desc.channels;
    size_t frame_len = n_i16 / channels;
    size_t slices = (frame_len + OQ_SLICE_LEN - 1) / OQ_SLICE_LEN;
    uint16_t frame_size = oq_frame_size((uint32_t)channels, (uint32_t)slices);
    size_t p = 0;

    uint64_t header = ((uint64_t)e->desc.channels << 56) | ((uint64_t)e->desc.sample_rate << 32) |
                      ((uint64_t)frame_len << 16) | (uint64_t)frame_size; /* :448-451, rate unmasked */
    put_be64(buf + p, header);
    p += 8;

    for (size_t c = 0; c < channels; c++) { /* :455-466 */
        uint64_t history = 0, weights = 0;
        for (int i = 0; i < OQ_LMS_LEN; i++) {
            history = (history << 16) | (uint64_t)(uint16_t)e->lms[c].history[i];
            weights = (weights << 16) | (uint64_t)(uint16_t)e->lms[c].weights[i];
        }
        put_be64(buf + p, history);
        put_be64(buf + p + 8, weights);
        p += 16;
    }

    for (size_t sample_index = 0; sample_index < frame_len; sample_index += OQ_SLICE_LEN) { /* :468 */
        for (size_t c = 0; c < channels; c++) {
            size_t slice_len = frame_len - sample_index;
            if (slice_len > OQ_SLICE_LEN) slice_len = OQ_SLICE_LEN;
            size_t start = sample_index * channels + c;
            size_t end = (sample_index + slice_len) * channels + c;
            oq_lms best;
            uint64_t slice = oq_encode_slice(&e->lms[c], pcm, start, end, channels, e->search,
                                             NULL, &best, NULL, &e->candidate_steps);
            e->slices++;
            e->lms[c] = best; /* :482 full i32 state carried, never the 16-bit header image */
            if (slice_len < OQ_SLICE_LEN) slice <<= (OQ_SLICE_LEN - slice_len) * 3; /* :485-487 */
            put_be64(buf + p, slice);
            p += 8;
        }
    }
    return p;
}

/* src/lib.rs:404-408 */
void oq_encoder_write_header(const oq_encoder *e, uint8_t out[8]) {
    put_be32(out, OQ_MAGIC);
    put_be32(out + 4, e->desc.samples);
}

/* src/lib.rs:367-398 */
int oq_encoder_encode(oq_encoder *e, const int16_t *pcm, size_t n_i16, uint8_t *out, size_t cap,
                      size_t *written) {
    size_t channels = e->desc.channels, total = e->desc.samples;
    if (n_i16 != total * channels) return OQ_ENC_INVALID_SAMPLES;
    size_t need = oq_encoded_size((uint32_t)channels, (uint32_t)total);
    if (cap < need) return OQ_ENC_IO_ERROR; /* C boundary only: caller-owned buffer too small */
    size_t p = 0;
    oq_encoder_write_header(e, out);
    p += 8;
    size_t sample_index = 0;
    while (sample_index < total) {
        size_t frame_len = total - sample_index;
        if (frame_len > OQ_FRAME_LEN) frame_len = OQ_FRAME_LEN;
        p += encode_frame_to_buf(e, pcm + sample_index * channels, frame_len * channels, out + p);
        sample_index += frame_len;
    }
    if (written) *written = p;
    return OQ_OK;
}

/* src/lib.rs:418-439 */
int oq_encoder_encode_frame(oq_encoder *e, const int16_t *pcm, size_t n_i16, uint8_t *out,
                            size_t cap, size_t *written, size_t *frame_len_out) {
    size_t channels = e->desc.channels;
    if (n_i16 == 0 || n_i16 % channels != 0) return OQ_ENC_INVALID_SAMPLES;
    size_t frame_len = n_i16 / channels;
    if (frame_len > OQ_FRAME_LEN) return OQ_ENC_INVALID_SAMPLES;
    size_t slices = (frame_len + OQ_SLICE_LEN - 1) / OQ_SLICE_LEN;
    size_t frame_size = oq_frame_size((uint32_t)channels, (uint32_t)slices);
    if (cap < frame_size) return OQ_ENC_IO_ERROR;
    size_t w = encode_frame_to_buf(e, pcm, n_i16, out);
    if (written) *written = w;
    if (frame_len_out) *frame_len_out = frame_len;
    return OQ_OK;
}

/* src/lib.rs:760-763 */
int oq_encode_all(const int16_t *pcm, size_t n_i16, const oq_desc *d, int search, uint8_t *out,
                  size_t cap, size_t *written) {
    oq_encoder e;
    int rc = oq_encoder_init(&e, d, search);
    if (rc) return rc;
    return oq_encoder_encode(&e, pcm, n_i16, out, cap, written);
}

/* ---------------------------------------------------------------- decoder */

struct oq_decoder {
    /* reader: Cursor over a growable byte vector (src/lib.rs:627, io::Cursor<Vec<u8>>) */
    uint8_t *data;
    size_t len, cap, pos;
    /* src/lib.rs:59-68 */
    int mode;
    uint8_t fixed_channels;
    uint32_t fixed_rate, fixed_samples;
    oq_lms lms[256];
    size_t lms_len;
    oq_frame_header cur_header;
    uint16_t remaining; /* CurrentFrame.num_samples_per_channel_remaining, :766-770 */
    int16_t pending[OQ_SLICE_LEN * 255];
    size_t pending_len, pending_end, next_pending;
    int returned_first_frame_header;
};

/* Cursor::read_exact: on a short read the cursor is left at EOF and nothing is returned. */
static int rd(oq_decoder *d, uint8_t *dst, size_t n) {
    if (d->len - d->pos < n) {
        d->pos = d->len;
        return -1;
    }
    memcpy(dst, d->data + d->pos, n);
    d->pos += n;
    return 0;
}

static int rd_u64(oq_decoder *d, uint64_t *v) {
    uint8_t b[8];
    if (rd(d, b, 8)) return -1;
    *v = get_be64(b);
    return 0;
}

/* src/lib.rs:205-289.  Returns 1 frame read, 0 clean EOF, <0 = -(OQ_DEC_*) */
static int decode_frame_header_and_lms(oq_decoder *d, int first) {
    uint64_t fh;
    if (rd_u64(d, &fh)) return 0; /* :206-216 UnexpectedEof is "expected" here */
    uint8_t num_channels = (uint8_t)((fh >> 56) & 0xff);
    uint32_t sample_rate = (uint32_t)((fh >> 32) & 0xffffff);
    uint16_t nspc = (uint16_t)((fh >> 16) & 0xffff);
    uint16_t frame_size = (uint16_t)(fh & 0xffff);

    if (num_channels == 0 || sample_rate == 0) return -OQ_DEC_INVALID_FRAME_HEADER; /* :227-229 */
    size_t non_sample = OQ_HEADER_SIZE + OQ_LMS_LEN * 4 * (size_t)num_channels; /* :231-235 */
    if ((size_t)frame_size <= non_sample) return -OQ_DEC_INVALID_FRAME_HEADER;
    size_t num_slices = ((size_t)frame_size - non_sample) / 8;
    if (num_slices % num_channels != 0) return -OQ_DEC_INVALID_FRAME_HEADER;                  /* :239-241 */
    if (num_slices / num_channels > OQ_MAX_SLICES_PER_CHANNEL_PER_FRAME) return -OQ_DEC_INVALID_FRAME_HEADER;

    if (d->mode == OQ_MODE_FIXED) { /* :246-259 */
        if (first) {
            d->fixed_channels = num_channels;
            d->fixed_rate = sample_rate;
        } else if (num_channels != d->fixed_channels || sample_rate != d->fixed_rate) {
            return -OQ_DEC_INCOMPATIBLE_FRAME;
        }
    }
    if (d->lms_len != num_channels) { /* :263-268 */
        for (size_t c = d->lms_len; c < num_channels; c++) memset(&d->lms[c], 0, sizeof(oq_lms));
        d->lms_len = num_channels;
    }
    for (size_t c = 0; c < num_channels; c++) { /* :271-281 */
        uint64_t history, weights;
        if (rd_u64(d, &history)) return -OQ_DEC_IO_ERROR;
        if (rd_u64(d, &weights)) return -OQ_DEC_IO_ERROR;
        for (int i = 0; i < OQ_LMS_LEN; i++) {
            d->lms[c].history[i] = (int32_t)(int16_t)(history >> 48);
            history <<= 16;
            d->lms[c].weights[i] = (int32_t)(int16_t)(weights >> 48);
            weights <<= 16;
        }
    }
    d->cur_header.num_channels = num_channels; /* :283-286 */
    d->cur_header.sample_rate = sample_rate;
    d->cur_header.num_samples_per_channel = nspc;
    d->remaining = nspc;
    return 1;
}

/* One channel's slice: the body of the per-channel loop of src/lib.rs:300-318 (scalefactor from the top 4 bits, 20 x
 * predict / dequantise / clamp / update; the state is copied out, stepped, and stored back as at :302 and :316).
 * Exposed for the whole-fixture pins (tests/test_oracle_golden.py): out[i * stride] receives sample i. */
void oq_decode_slice(oq_lms *state, uint64_t slice, int16_t *out, size_t stride) {
    int sf = (int)((slice >> 60) & 0xf);
    oq_lms lms = *state;
    size_t idx = 0;
    for (int i = 0; i < OQ_SLICE_LEN; i++) {
        int32_t prediction = lms_predict(&lms);
        int quantized = (int)((slice >> 57) & 0x7);
        int32_t dequantized = DEQUANT_TAB[sf][quantized];
        int16_t reconstructed = (int16_t)clampi(prediction + dequantized, -32768, 32767);
        out[idx] = reconstructed;
        idx += stride;
        slice <<= 3;
        lms_update(&lms, reconstructed, dequantized);
    }
    *state = lms;
}

/* src/lib.rs:291-330.  Returns 0 or -(OQ_DEC_IO_ERROR) */
static int decode_one_slice_per_channel(oq_decoder *d) {
    size_t channels = d->cur_header.num_channels;
    size_t full = OQ_SLICE_LEN * channels;
    if (d->pending_len != full) {
        d->pending_len = full;
        memset(d->pending, 0, full * sizeof(int16_t));
    }
    d->next_pending = 0;
    for (size_t c = 0; c < channels; c++) {
        uint64_t slice;
        if (rd_u64(d, &slice)) return -OQ_DEC_IO_ERROR;
        oq_decode_slice(&d->lms[c], slice, d->pending + c, channels);
    }
    uint16_t n = d->remaining; /* :320-328 */
    if (n < OQ_SLICE_LEN) {
        d->pending_end = (size_t)n * channels;
        d->remaining = 0;
    } else {
        d->pending_end = full;
        d->remaining = (uint16_t)(n - OQ_SLICE_LEN);
    }
    return 0;
}

static oq_decoder *decoder_alloc(const uint8_t *bytes, size_t len) {
    oq_decoder *d = (oq_decoder *)calloc(1, sizeof *d);
    if (!d) return NULL;
    d->cap = len > 64 ? len : 64;
    d->data = (uint8_t *)malloc(d->cap);
    if (len) memcpy(d->data, bytes, len);
    d->len = len;
    return d;
}

void oq_decoder_free(oq_decoder *d) {
    if (!d) return;
    free(d->data);
    free(d);
}

/* src/lib.rs:127-169 */
int oq_decoder_new(const uint8_t *bytes, size_t len, oq_decoder **out) {
    *out = NULL;
    oq_decoder *d = decoder_alloc(bytes, len);
    if (!d) return OQ_DEC_IO_ERROR;
    uint8_t b[4];
    if (rd(d, b, 4)) { oq_decoder_free(d); return OQ_DEC_IO_ERROR; }
    uint32_t magic = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    if (magic != OQ_MAGIC) { oq_decoder_free(d); return OQ_DEC_NOT_QOA_FILE; }
    if (rd(d, b, 4)) { oq_decoder_free(d); return OQ_DEC_IO_ERROR; }
    uint32_t samples = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    if (samples == 0) {
        d->mode = OQ_MODE_STREAMING;
    } else {
        d->mode = OQ_MODE_FIXED;
        d->fixed_samples = samples;
        int r = decode_frame_header_and_lms(d, 1);
        if (r < 0) { oq_decoder_free(d); return -r; }
        if (r == 0) { oq_decoder_free(d); return OQ_DEC_NO_SAMPLES; }
    }
    *out = d;
    return OQ_OK;
}

/* src/lib.rs:632-638 */
int oq_decoder_new_streaming(oq_decoder **out) {
    static const uint8_t hdr[8] = {'q', 'o', 'a', 'f', 0, 0, 0, 0};
    return oq_decoder_new(hdr, 8, out);
}

int oq_decoder_mode(const oq_decoder *d, uint8_t *channels, uint32_t *rate, uint32_t *samples) {
    if (channels) *channels = d->fixed_channels;
    if (rate) *rate = d->fixed_rate;
    if (samples) *samples = d->fixed_samples;
    return d->mode;
}

void oq_decoder_current_frame_header(const oq_decoder *d, oq_frame_header *h) { *h = d->cur_header; }

/* test accessor: the decoder's per-channel predictor state (the `lms` field, src/lib.rs:63; written at :271-281, :316) */
int oq_decoder_lms(const oq_decoder *d, size_t channel, oq_lms *out) {
    if (channel >= d->lms_len) return -1;
    *out = d->lms[channel];
    return 0;
}

/* src/lib.rs:668-692 (the tail recursion there is a loop here) */
void oq_decoder_next(oq_decoder *d, oq_item *item) {
    memset(item, 0, sizeof *item);
    for (;;) {
        if (d->next_pending < d->pending_end) {
            item->kind = OQ_ITEM_SAMPLE;
            item->sample = d->pending[d->next_pending++];
            return;
        }
        if (!d->returned_first_frame_header) {
            d->returned_first_frame_header = 1;
            item->kind = OQ_ITEM_FRAME_HEADER;
            item->header = d->cur_header;
            return;
        }
        if (d->remaining > 0) {
            int r = decode_one_slice_per_channel(d);
            if (r < 0) { item->kind = OQ_ITEM_ERROR; item->error = -r; return; }
            continue;
        }
        int r = decode_frame_header_and_lms(d, 0);
        if (r > 0) { item->kind = OQ_ITEM_FRAME_HEADER; item->header = d->cur_header; return; }
        if (r == 0) { item->kind = OQ_ITEM_END; return; }
        item->kind = OQ_ITEM_ERROR;
        item->error = -r;
        return;
    }
}

/* src/lib.rs:641-651 */
int oq_decoder_decode_frame(oq_decoder *d, const uint8_t *frame, size_t len, int16_t *out,
                            size_t cap, size_t *n_out) {
    if (d->len + len > d->cap) {
        size_t ncap = (d->len + len) * 2;
        d->data = (uint8_t *)realloc(d->data, ncap);
        d->cap = ncap;
    }
    memcpy(d->data + d->len, frame, len);
    d->len += len;
    size_t n = 0;
    oq_item it;
    for (;;) {
        oq_decoder_next(d, &it);
        if (it.kind == OQ_ITEM_END) break;
        if (it.kind == OQ_ITEM_ERROR) { if (n_out) *n_out = n; return it.error; }
        if (it.kind == OQ_ITEM_SAMPLE) {
            if (out && n < cap) out[n] = it.sample;
            n++;
        }
    }
    if (n_out) *n_out = n;
    return OQ_OK;
}

/* src/lib.rs:713-747 */
int oq_decode_all(const uint8_t *bytes, size_t len, uint8_t *channels, uint32_t *rate,
                  int16_t *out, size_t cap, size_t *n_out) {
    oq_decoder *d;
    int rc = oq_decoder_new(bytes, len, &d);
    if (rc) return rc;
    oq_item it;
    oq_decoder_next(d, &it); /* :724-731: always the first frame header */
    if (it.kind == OQ_ITEM_ERROR) { oq_decoder_free(d); return it.error; }
    uint8_t nch = it.header.num_channels;
    uint32_t sr = it.header.sample_rate;
    size_t n = 0;
    for (;;) {
        oq_decoder_next(d, &it);
        if (it.kind == OQ_ITEM_END) break;
        if (it.kind == OQ_ITEM_ERROR) { oq_decoder_free(d); return it.error; }
        if (it.kind == OQ_ITEM_SAMPLE) {
            if (out && n < cap) out[n] = it.sample;
            n++;
        } else if (it.header.num_channels != nch || it.header.sample_rate != sr) { /* :735-738 */
            oq_decoder_free(d);
            return OQ_DEC_INCOMPATIBLE_FRAME;
        }
    }
    oq_decoder_free(d);
    if (channels) *channels = nch;
    if (rate) *rate = sr;
    if (n_out) *n_out = n;
    return OQ_OK;
}
