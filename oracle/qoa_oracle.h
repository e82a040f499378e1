/*
 * qoa_oracle.h -- CPU restatement of rafaelcaricio/qoaudio's QOA encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the MI355X engine in
 * qoaudio_amd/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may import, link or execute anything under oracle/.  The product library
 * (qoaudio_amd/libqoaudio_amd.so) never links or calls it and has no CPU fallback.
 *
 * Each function cites the reference region (file:line into /root/reference) it
 * restates.  Pinning status: pinned -- see oracle/README.md (fixture fixed point KAT-1, frame-header chain over all 468
 * frames, teacher-forced slices over all 239,414 slices of the reference's fixture).
 */
#ifndef QOA_ORACLE_H
#define QOA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OQ_SLICE_LEN 20            /* src/lib.rs:12 */
#define OQ_LMS_LEN 4               /* src/lib.rs:13 */
#define OQ_HEADER_SIZE 8           /* src/lib.rs:14 */
#define OQ_MAGIC 0x716f6166u       /* "qoaf", src/lib.rs:15 */
#define OQ_MAX_SLICES_PER_CHANNEL_PER_FRAME 256 /* src/lib.rs:16 */
#define OQ_FRAME_LEN (256 * 20)    /* src/lib.rs:17-18 */
#define OQ_MAX_CHANNELS 8          /* src/lib.rs:19 */

/* src/lib.rs:70-74 */
typedef struct {
    int32_t history[OQ_LMS_LEN];
    int32_t weights[OQ_LMS_LEN];
} oq_lms;

/* src/lib.rs:95-102 */
typedef struct {
    uint8_t channels;
    uint32_t sample_rate;
    uint32_t samples;
} oq_desc;

/* EncodeError, src/lib.rs:106-115 */
enum {
    OQ_OK = 0,
    OQ_ENC_INVALID_CHANNELS = 1,
    OQ_ENC_INVALID_SAMPLE_RATE = 2,
    OQ_ENC_INVALID_SAMPLES = 3,
    OQ_ENC_IO_ERROR = 4
};

/* DecodeError, src/lib.rs:866-873 */
enum {
    OQ_DEC_NOT_QOA_FILE = 1,
    OQ_DEC_NO_SAMPLES = 2,
    OQ_DEC_INVALID_FRAME_HEADER = 3,
    OQ_DEC_INCOMPATIBLE_FRAME = 4,
    OQ_DEC_IO_ERROR = 5
};

/* Scalefactor search strategy.  Both must give byte-identical streams. */
enum {
    OQ_SEARCH_REF_STYLE = 0, /* rank-sorted order + both early exits, src/lib.rs:516-593 */
    OQ_SEARCH_EXHAUSTIVE = 1 /* all 16 candidates to slice_len; argmin (total, first, sf) */
};

/* src/lib.rs:85-91 (prev_scalefactor is written but never read there; omitted) */
typedef struct {
    oq_desc desc;
    oq_lms lms[OQ_MAX_CHANNELS];
    int search;
    uint64_t candidate_steps; /* instrumentation: candidate-sample steps executed */
    uint64_t slices;
} oq_encoder;

size_t oq_encoded_size(uint32_t channels, uint32_t samples);            /* src/lib.rs:374-378 */
uint16_t oq_frame_size(uint32_t channels, uint32_t slices);             /* src/lib.rs:602-604 */
int32_t oq_div(int32_t v, int sf);                                      /* src/lib.rs:613-617 */
int oq_encoder_init(oq_encoder *e, const oq_desc *d, int search);       /* src/lib.rs:335-361 */
int oq_encoder_encode(oq_encoder *e, const int16_t *pcm, size_t n_i16,  /* src/lib.rs:367-398 */
                      uint8_t *out, size_t cap, size_t *written);
void oq_encoder_write_header(const oq_encoder *e, uint8_t out[8]);      /* src/lib.rs:404-408 */
int oq_encoder_encode_frame(oq_encoder *e, const int16_t *pcm, size_t n_i16, /* src/lib.rs:418-439 */
                            uint8_t *out, size_t cap, size_t *written, size_t *frame_len);
int oq_encode_all(const int16_t *pcm, size_t n_i16, const oq_desc *d, int search, /* src/lib.rs:760-763 */
                  uint8_t *out, size_t cap, size_t *written);

/* One slice, exposed for known-answer tests: src/lib.rs:495-596.
 * total_rank_out (optional, 16 entries) receives every candidate's full-length rank
 * (exhaustive evaluation), independent of `search`. */
uint64_t oq_encode_slice(const oq_lms *entry, const int16_t *pcm, size_t start, size_t end,
                         size_t stride, int search, int *best_sf, oq_lms *best_lms,
                         uint64_t *total_rank_out, uint64_t *steps);

/* instrumentation (oracle/README.md "what the fixture pins"): per-candidate rank terms of one slice, all 16 candidates
 * run to the end; rank = err_sq + pen_sq (src/lib.rs:576) */
void oq_slice_rank_terms(const oq_lms *entry, const int16_t *pcm, size_t start, size_t end, size_t stride,
                         uint64_t err_sq[16], uint64_t pen_sq[16], uint64_t first_err_sq[16], uint64_t first_pen_sq[16]);

/* instrumentation: the slice word of each of the 16 candidates run to the end (before the tail's left shift) */
void oq_slice_candidates(const oq_lms *entry, const int16_t *pcm, size_t start, size_t end, size_t stride,
                         uint64_t words[16]);

/* ---- decoder: src/lib.rs:59-68, 127-330, 627-693 ---- */

/* FrameHeader, src/lib.rs:774-781 */
typedef struct {
    uint8_t num_channels;
    uint32_t sample_rate;
    uint16_t num_samples_per_channel;
} oq_frame_header;

enum { OQ_MODE_FIXED = 0, OQ_MODE_STREAMING = 1 }; /* ProcessingMode, src/lib.rs:31-45 */

enum { OQ_ITEM_SAMPLE = 0, OQ_ITEM_FRAME_HEADER = 1, OQ_ITEM_END = 2, OQ_ITEM_ERROR = 3 };

typedef struct {
    int kind;
    int16_t sample;
    oq_frame_header header;
    int error;
} oq_item;

typedef struct oq_decoder oq_decoder;

/* QoaDecoder::new over an in-memory reader (bytes are copied). Returns 0 or a OQ_DEC_* code. */
int oq_decoder_new(const uint8_t *bytes, size_t len, oq_decoder **out);        /* src/lib.rs:127-169 */
int oq_decoder_new_streaming(oq_decoder **out);                                /* src/lib.rs:632-638 */
void oq_decoder_free(oq_decoder *d);
int oq_decoder_mode(const oq_decoder *d, uint8_t *channels, uint32_t *rate, uint32_t *samples); /* :180-182 */
void oq_decoder_current_frame_header(const oq_decoder *d, oq_frame_header *h); /* src/lib.rs:186-188 */
void oq_decoder_next(oq_decoder *d, oq_item *item);                            /* src/lib.rs:668-692 */
/* test accessors for the whole-fixture pins (oracle/README.md): the decoder's per-channel LMS state (src/lib.rs:63) and one
 * channel's slice stepped from an explicit state (src/lib.rs:300-318); out[i * stride] = sample i. */
int oq_decoder_lms(const oq_decoder *d, size_t channel, oq_lms *out);
void oq_decode_slice(oq_lms *state, uint64_t slice, int16_t *out, size_t stride);
/* decode_frame: appends bytes to the reader and drains the iterator.  src/lib.rs:641-651.
 * out may be NULL to query the count; returns 0 or OQ_DEC_* (samples decoded before an error are lost,
 * as in the reference, but *n_out still reports how many were produced). */
int oq_decoder_decode_frame(oq_decoder *d, const uint8_t *frame, size_t len,
                            int16_t *out, size_t cap, size_t *n_out);
/* decode_all, src/lib.rs:713-747 */
int oq_decode_all(const uint8_t *bytes, size_t len, uint8_t *channels, uint32_t *rate,
                  int16_t *out, size_t cap, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
