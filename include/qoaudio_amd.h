/*
 * qoaudio_amd.h -- C ABI of the MI355X-native QOA encode/decode engine.
 *
 * This is the drop-in boundary for rafaelcaricio/qoaudio's public API (the crate has no FFI of
 * its own: its boundary is its Rust API, src/lib.rs).  Every entry point below names the
 * reference item it replaces (file:line into the reference tree).  Plain pointers and sizes
 * only -- no torch, no C++ types.  A Rust shim that re-exports the crate's original names over
 * these symbols is shown in INTEGRATION.md; include/qoaudio.hpp is the C++ mirror used by the
 * tests.
 *
 * All compute (slice search, LMS recurrence, slice decode) runs in hand-written HIP kernels for
 * gfx950.  There is NO CPU fallback: if no HIP device is usable every compute entry point
 * returns the IoError code and qoa_last_error_detail() says why.
 *
 * Threading: handles are independent and not internally synchronised (the reference's
 * `&mut self` rule); the library keeps no mutable global state except a thread-local error string.
 * Device: every call runs on the calling thread's current HIP device (hipSetDevice / torch.cuda.set_device).
 */
#ifndef QOAUDIO_AMD_H
#define QOAUDIO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- format constants: src/lib.rs:12-19 ---- */
#define QOA_SLICE_LEN 20
#define QOA_LMS_LEN 4
#define QOA_HEADER_SIZE 8
#define QOA_MAGIC 0x716f6166u
#define QOA_SLICES_PER_FRAME 256
#define QOA_FRAME_LEN (QOA_SLICES_PER_FRAME * QOA_SLICE_LEN)
#define QOA_MAX_CHANNELS 8

/* QoaDesc, src/lib.rs:95-102 */
typedef struct qoa_desc {
    uint8_t channels;
    uint32_t sample_rate;
    uint32_t samples; /* per channel */
} qoa_desc;

/* EncodeError, src/lib.rs:106-115.  0 = Ok. */
enum {
    QOA_OK = 0,
    QOA_ENCODE_INVALID_CHANNELS = 1,
    QOA_ENCODE_INVALID_SAMPLE_RATE = 2,
    QOA_ENCODE_INVALID_SAMPLES = 3,
    QOA_ENCODE_IO_ERROR = 4 /* also: output buffer too small, HIP failure (see qoa_last_error_detail) */
};

/* DecodeError, src/lib.rs:866-873.  0 = Ok. */
enum {
    QOA_DECODE_NOT_QOA_FILE = 1,
    QOA_DECODE_NO_SAMPLES = 2,
    QOA_DECODE_INVALID_FRAME_HEADER = 3,
    QOA_DECODE_INCOMPATIBLE_FRAME = 4,
    QOA_DECODE_IO_ERROR = 5 /* also: truncated stream (UnexpectedEof), HIP failure */
};

/* Display strings of the two error enums, src/lib.rs:877-906 ("IO error: " + detail for IoError). */
const char *qoa_encode_strerror(int code);
const char *qoa_decode_strerror(int code);
/* Detail of the last IoError raised on this thread by this library (HIP error text, "buffer too small"...).
 * The library never aborts and no C++ exception leaves it: host memory exhaustion inside an entry point comes back as the
 * family's IoError with the detail "out of host memory" (qoa_decoder_next: as a QOA_ITEM_ERROR item). */
const char *qoa_last_error_detail(void);
/* "qoaudio_amd <version> gfx950" */
const char *qoa_version(void);

/* Exact encoded size, src/lib.rs:374-378.  channels in 1..=8. */
size_t qoa_encoded_size(uint32_t channels, uint32_t samples);
/* qoa_frame_size, src/lib.rs:602-604 (wraps to u16 like the reference). */
uint16_t qoa_frame_size(uint32_t channels, uint32_t slices);

/* ================================================================== encoder */

typedef struct qoa_encoder qoa_encoder;

/* QoaEncoder::new, src/lib.rs:335-361.  Validation order channels -> sample_rate -> samples. */
int qoa_encoder_new(const qoa_desc *desc, qoa_encoder **out);
void qoa_encoder_free(qoa_encoder *enc);

/* QoaEncoder::encode, src/lib.rs:367-398.  n_i16 must equal samples*channels (else InvalidSamples).
 * pcm/out are HOST pointers; out must hold qoa_encoded_size() bytes. */
int qoa_encoder_encode(qoa_encoder *enc, const int16_t *pcm, size_t n_i16, uint8_t *out, size_t cap,
                       size_t *written);

/* QoaEncoder::write_header, src/lib.rs:404-408. */
int qoa_encoder_write_header(const qoa_encoder *enc, uint8_t out[8]);

/* QoaEncoder::encode_frame, src/lib.rs:418-439.  Empty / not a multiple of channels / more than
 * 5120 sample-times -> InvalidSamples.  LMS state (full i32) persists in device memory inside the
 * handle between calls.  *frame_len receives the sample-times encoded. */
int qoa_encoder_encode_frame(qoa_encoder *enc, const int16_t *pcm, size_t n_i16, uint8_t *out, size_t cap,
                             size_t *written, size_t *frame_len);

/* Test/checkpoint hook ("checkpoint/resume" analogue): read or overwrite the
 * encoder's carried LMS state, 8 i32 per channel = history[4] then weights[4] (src/lib.rs:70-74, :89). */
int qoa_encoder_get_lms(qoa_encoder *enc, int32_t *state /* channels*8 */);
int qoa_encoder_set_lms(qoa_encoder *enc, const int32_t *state /* channels*8 */);

/* Test hook: run ONE slice (src/lib.rs:495-596) from an explicit entry state through the device functions of
 * the encode kernel and return all 16 candidates: total rank, first-sample rank, packed slice (tail already
 * left-aligned) and end LMS state [16][8].  generic: 0 = one-asm slice with 32-bit weights, 1 = rolled tail
 * path, 2 = the same step as 20 separate statements, 3 = the i16-packed fast path with its fall-back, as K1 runs it.
 * first_rank[i] = (the row's winner << 32) | |err0| of candidate i (what orders the first-sample ranks); with
 * generic == 3, bit 63 is set when the packed path declined the slice (state or weight energy out of its range)
 * and the 32-bit path produced the results. */
int qoa_debug_encode_slice(const int32_t *state /* 8 */, const int16_t *samples, int len /* 1..20 */, int generic,
                           uint64_t *total_rank /* 16 */, uint64_t *first_rank /* 16 */, uint64_t *slices /* 16 */,
                           int32_t *end_states /* 16*8 */);

/* Test hook: the handle's later launches take the step-form switches of qoa_encode_plan_create's flags (QOA_PLAN_* below;
 * QOA_PLAN_COUNTERS is ignored).  Same bytes by construction -- the fuzz tests compare every form from hostile states. */
int qoa_debug_encoder_set_plan_flags(qoa_encoder *enc, uint32_t flags);

/* encode_all, src/lib.rs:760-763. */
int qoa_encode_all(const int16_t *pcm, size_t n_i16, const qoa_desc *desc, uint8_t *out, size_t cap,
                   size_t *written);

/* ---- batch encode on device-resident buffers (new: semantically encode_all per clip) ---- */

/* One clip of a batch.  pcm_offset counts int16 elements from the pcm base pointer, out_offset
 * counts bytes from the out base pointer; the clip's output occupies qoa_encoded_size() bytes. */
typedef struct qoa_clip {
    uint64_t pcm_offset;
    uint64_t out_offset;
    qoa_desc desc;
} qoa_clip;

typedef struct qoa_encode_plan qoa_encode_plan;

/* flags for qoa_encode_plan_create */
#define QOA_PLAN_DEFAULT 0u
#define QOA_PLAN_FORCE_GENERIC_PATH 1u /* debug: run the rolled tail-slice path for every slice */
#define QOA_PLAN_FORCE_32BIT_STEP 2u   /* debug: never take the i16-packed fast path (same bytes, tests compare) */
#define QOA_PLAN_NO_SPECULATION 4u     /* debug: packed path always with its penalty arithmetic (same bytes) */
#define QOA_PLAN_LDS_TABLES 8u         /* debug: quantiser / dequantiser looked up in an LDS table in every launch (same bytes; */
#define QOA_PLAN_NO_LDS_TABLES 128u    /*        default: only in launches with >= 8 wavefronts per SIMD) / in none */
#define QOA_PLAN_EXACT_QUOTIENT 256u   /* debug: speculative steps always with the exact qoa_div fix-up (same bytes) */
#define QOA_PLAN_NO_PENALTY_FORM 1024u /* debug: where the speculative steps miss, go straight to the full form (same bytes) */
#define QOA_PLAN_NO_BLOCK32 2048u       /* debug: slices the packed forms cannot take go one by one, not through the 32-bit block loop */
#define QOA_PLAN_NO_HELPER 4096u        /* debug: no helper wavefronts (K1's two-wavefront workgroups for launches that leave SIMDs idle) */
#define QOA_PLAN_FORCE_HELPER 8192u     /* debug: helper wavefronts whatever the launch size */
#define QOA_PLAN_NO_WIDE_FORM 16384u     /* debug: no wide form (exact quotient with the true 64-bit rank: full-scale noise); same bytes */
#define QOA_PLAN_COUNTERS 512u         /* count the fast paths' fall-backs (qoa_encode_plan_stats / _counters); without it the */
                                       /* kernel is handed no counter block and executes no atomic at all */
#define QOA_PLAN_WAVE_STATS 32768u     /* every compute wavefront records, once, at its exit, how many slices it ran on each step */
                                       /* form and how long it took (qoa_encode_plan_wave_stats): no atomics, same bytes */
#define QOA_PLAN_CPW_1 (1u << 4)       /* force chains-per-wavefront = 1 / 2 / 4 (default: 4) */
#define QOA_PLAN_CPW_2 (2u << 4)
#define QOA_PLAN_CPW_4 (4u << 4)

/* Validates every desc like QoaEncoder::new; uploads the chain table to the current device. */
int qoa_encode_plan_create(const qoa_clip *clips, size_t n_clips, uint32_t flags, qoa_encode_plan **out);
void qoa_encode_plan_free(qoa_encode_plan *plan);
/* Bytes the batch reads / writes (algorithmic bytes of one run). */
int qoa_encode_plan_bytes(const qoa_encode_plan *plan, uint64_t *pcm_bytes, uint64_t *out_bytes,
                          uint64_t *samples_i16);
/* How the plan's launches are shaped (diagnostics for bench.py): total compute wavefronts over its launches, chains per
 * wavefront of the largest launch, whether that launch takes the LDS-table step form (chosen from 8 wavefronts per SIMD up,
 * or forced by QOA_PLAN_LDS_TABLES).  Any pointer may be null. */
int qoa_encode_plan_launch_info(const qoa_encode_plan *plan, uint32_t *wavefronts, uint32_t *chains_per_wavefront,
                                uint32_t *lds_table_form);
/* The same and more as a struct that can grow: set struct_bytes = sizeof(qoa_launch_shape) before the call.
 * helper_wavefronts: the largest launch runs two-wavefront workgroups (a helper wavefront beside every compute wavefront:
 * launches of up to 512 compute wavefronts); launches: kernel launches per run (one per channel count in the batch). */
typedef struct qoa_launch_shape {
    uint32_t struct_bytes;
    uint32_t wavefronts;
    uint32_t chains_per_wavefront;
    uint32_t lds_table_form;
    uint32_t helper_wavefronts;
    uint32_t launches;
} qoa_launch_shape;
int qoa_encode_plan_launch_shape(const qoa_encode_plan *plan, qoa_launch_shape *shape);
/* Enqueue the encode of the whole batch on `stream` (a hipStream_t, may be NULL).  d_pcm and
 * d_out are DEVICE pointers.  Asynchronous: returns after the launch. */
int qoa_encode_plan_run(qoa_encode_plan *plan, const int16_t *d_pcm, uint8_t *d_out, void *stream);
/* Diagnostics of the plan's runs so far (synchronises the device); none of this affects the bytes produced.  Only plans
 * created with QOA_PLAN_COUNTERS count; for any other plan both calls return the IoError code (detail: "plan was created
 * without QOA_PLAN_COUNTERS").
 *   *declined_slices    (chain, slice) pairs for which the kernel's i16-packed path could not prove its result exact
 *                       and the 32-bit path evaluated the slice again;
 *   *unpacked_slices    full slices that went straight to the 32-bit path because a chain's LMS state was outside the
 *                       packed path's range (|weight| > 32767 or weight energy too high);
 *   *speculation_misses slices whose speculative evaluation (steps without penalty arithmetic) could not be proven and
 *                       was repeated with it -- each miss also parks its wavefront on the full form for 4 blocks
 *                       (64 slices).  Real music with strongly tonal passages shows a few per cent of these.
 * The first two are 0 on ordinary audio.  Any pointer may be null. */
int qoa_encode_plan_stats(qoa_encode_plan *plan, uint64_t *declined_slices, uint64_t *unpacked_slices,
                          uint64_t *speculation_misses);
/* All counters, out[0..n): the three above, then
 *   out[3] slices the cheap-quotient step form declined (level from |quotient|, sign from the residual: exact unless a
 *          scalefactor-0/1 candidate's quotient wraps to the opposite sign; such a candidate is marked and must be beaten by
 *          a bound derived from the slice's samples, else the slice is repeated with the exact qoa_div -- loud passages);
 *   out[4] (candidate, slice) pairs marked at all (most are settled in place: the marked candidate enters the election
 *          with the rank it provably costs at least, and loses);
 *   out[5] (wavefront, slice) pairs evaluated on the 32-bit step (block statement, one-slice statement or tail path), out[6] on the one-slice packed statement;
 *   out[7] slices the penalty form (cheap-quotient steps with the penalty arithmetic: the fall-back for speculation
 *          misses) declined and handed on;
 *   out[8] slices the wide form (exact quotient with the true 64-bit rank: full-scale noise) declined and handed to the
 *          full form, out[9] (wavefront, slice) pairs it settled.
 * Entries beyond the last counter read 0. */
int qoa_encode_plan_counters(qoa_encode_plan *plan, uint64_t *out, size_t n);

/* What a batch cost, wavefront by wavefront (plans created with QOA_PLAN_WAVE_STATS; the LAST run's records; synchronises
 * the device).  K1's speed depends on the material: a slice runs on the cheapest step form whose exactness the kernel can
 * prove for it (src/lib.rs:516-593 is the search being reproduced; :613-617 the quotient that wraps at scalefactors 0 / 1 on
 * loud material, :823-826 the weights that hard-clipped input drives out of the i16 range), a wavefront is as slow as the
 * forms its four chains need, and a batch takes as long as its slowest wavefront.  Forms, cheapest first:
 *   0 cheap-quotient speculative   1 exact-quotient speculative (or LDS-table)   2 penalty   3 wide (64-bit rank)
 *   4 full packed   5 32-bit block statement   6 one-slice statements (tail slices, ragged blocks)   7 blocks (not slices)
 * slices[f] counts wavefront-slices ATTEMPTED on form f (a slice a form declines is counted there and on the form that took
 * it over).  issue_slots[f] is the instruction count of one slice on that form, cycles_per_issue what a lone wavefront pays
 * per instruction (4.06) and loop_overhead what a block loop costs beyond its steps (barrier, dispatch, loop branch, the
 * winner's LDS round trip: 5-10 %, taken as 1.10): ordinary music costs about 519 x 4.06 x 1.10 = 2,320 cycles per slice, and
 * no input should cost more than bound_cycles_per_slice = 842 x 4.06 x 1.10 = 3,760 (every slice on the 32-bit statement):
 * clipped material runs at most ~1.6x slower than music.  cycles_per_slice_* are measured (s_memtime around the wavefront's
 * whole life / slices of its longest chain); slowest_* describe the wavefront with the highest measured figure. */
#define QOA_WAVE_FORMS 8
typedef struct qoa_wave_stats {
    uint32_t struct_bytes;                     /* in: sizeof(qoa_wave_stats) */
    uint32_t wavefronts;                       /* compute wavefronts that ran slices */
    uint64_t slices[QOA_WAVE_FORMS];
    uint32_t issue_slots[QOA_WAVE_FORMS];
    double cycles_per_issue;
    double loop_overhead;
    double bound_cycles_per_slice;
    double cycles_per_slice_min, cycles_per_slice_p50, cycles_per_slice_max;
    uint32_t slowest_wavefront;                /* index over the plan's launches in order */
    uint32_t slowest_channels;                 /* channel count of its launch */
    uint32_t slowest_slices_of_chain;          /* slices of its longest chain */
    uint32_t slowest_slices[QOA_WAVE_FORMS];   /* its form mix */
    double slowest_predicted_cycles_per_slice; /* sum(slowest_slices x issue_slots) x cycles_per_issue x loop_overhead / slices */
} qoa_wave_stats;
int qoa_encode_plan_wave_stats(qoa_encode_plan *plan, qoa_wave_stats *out);

/* ---- batch encode on HOST-resident buffers: time-sliced pipeline ----
 * Semantically encode_all per clip (src/lib.rs:760-763) with h_pcm / h_out in host memory (pinned memory makes
 * the copies asynchronous).  The batch is cut along TIME: frames [j*F, (j+1)*F) of all clips are one kernel
 * launch (LMS state carried in device memory, as encode_frame does, src/lib.rs:414), overlapped with the
 * host->device copy of the next time slice and the device->host copy of the previous one.  chunk_frames = F,
 * 0 picks the default (8).  Synchronous: returns when h_out is complete. */
int qoa_encode_batch_host(const qoa_clip *clips, size_t n_clips, const int16_t *h_pcm, uint8_t *h_out,
                          uint32_t chunk_frames);

/* The same pipeline as a persistent handle: its three streams, six events and device staging buffers are created once
 * and kept (buffers grow on demand), so repeated batches stop paying that set-up.  One caller at a time per handle;
 * every call is synchronous on return.  qoa_host_pipeline_encode == qoa_encode_batch_host; qoa_host_pipeline_decode is
 * the decode counterpart: clips[i].out_offset locates clip i's QOA stream (as encode wrote it: 8-byte file header +
 * frames) in h_enc, clips[i].pcm_offset its int16 PCM in h_pcm; frames [j*F, (j+1)*F) of all clips are one launch of the
 * decode kernels, overlapped with the copy-in of the next and the copy-out of the previous slice (2 bytes per sample
 * back over PCIe: that copy is the bound).  Every frame header is checked on the device against the clip's desc:
 * a mismatch anywhere returns QOA_DECODE_INVALID_FRAME_HEADER.  Decode entry points return QOA_DECODE_* codes. */
typedef struct qoa_host_pipeline qoa_host_pipeline;
int qoa_host_pipeline_create(qoa_host_pipeline **out);
void qoa_host_pipeline_free(qoa_host_pipeline *p);
int qoa_host_pipeline_encode(qoa_host_pipeline *p, const qoa_clip *clips, size_t n_clips, const int16_t *h_pcm,
                             uint8_t *h_out, uint32_t chunk_frames);
int qoa_host_pipeline_decode(qoa_host_pipeline *p, const qoa_clip *clips, size_t n_clips, const uint8_t *h_enc,
                             int16_t *h_pcm, uint32_t chunk_frames);
/* one-shot form of qoa_host_pipeline_decode */
int qoa_decode_batch_host(const qoa_clip *clips, size_t n_clips, const uint8_t *h_enc, int16_t *h_pcm,
                          uint32_t chunk_frames);

/* ---- multi-GPU batches in ONE process (BASELINE config 4; no counterpart in the reference, which is
 * single-threaded: what a Rust host would call per clip is encode_all, src/lib.rs:760-763) ----
 * devices[0 .. n_devices) are HIP device ordinals; devices[0] is the ROOT.  Clips are independent, so the batch is cut into
 * n_devices contiguous shards -- qoa_multi_shard gives shard i's [first, first + count): the first n_clips % n_devices
 * shards hold one clip more -- and device i encodes / decodes shard i with its own host thread, streams and time-sliced
 * pipeline (qoa_host_pipeline_*), with no collective on the data path.
 *   encode: pcm[i] is the base of shard i's PCM -- host memory, or memory of devices[i] -- and the shard's clips'
 *           pcm_offset count from it; `out` is ONE destination for the whole batch, host memory or memory of the root, and
 *           every out_offset counts from it.  A shard's encoded bytes travel to `out` slice by slice while the next slice
 *           is being encoded (peer copies over xGMI when `out` is on the root): that IS the gather of the encoded streams.
 *   decode: `enc` is ONE source (host or root memory, out_offset counts from it), pcm[i] the destination base of shard i.
 * A shard's PCM / the output that already sits in the memory of the device that works on it is used in place (no staging
 * copy) when the encoded side's base and offsets are 8-byte aligned -- otherwise, and for memory of any other device
 * (legal wherever the HIP runtime can copy from / to it), the shard goes through the pipeline's staging buffers.
 * Stream ordering: the pipelines run on their own streams.  Before anything starts, every call waits for ALL work
 * submitted so far to the devices it uses (hipDeviceSynchronize), so device buffers filled by earlier kernels or copies on
 * any stream are complete when they are read or overwritten; the call is synchronous: complete on return.  The persistent
 * handle keeps the chain / frame tables of the last batch (keyed on the clip list): repeating a batch rebuilds nothing.
 * Errors: a device ordinal that does not exist or is listed twice, and every HIP failure (peer access included), return
 * the IoError code with qoa_last_error_detail() naming the device; nothing aborts. */
typedef struct qoa_multi qoa_multi;
int qoa_multi_create(const int *devices, int n_devices, qoa_multi **out);
/* qoa_multi_create with flags.  QOA_MULTI_ALLOW_DUPLICATE_DEVICES is a TEST mode for boxes with fewer GPUs than the
 * configuration under test: a device may be listed more than once, every entry gets its own host thread and pipeline, and
 * every shard but the first treats the shared buffer (`out` of an encode, `enc` of a decode) as memory of a PEER device even
 * though it is its own -- staging buffers and copies, exactly the path a second GPU takes.  Same bytes, no speed-up. */
#define QOA_MULTI_ALLOW_DUPLICATE_DEVICES 1u
int qoa_multi_create_flags(const int *devices, int n_devices, uint32_t flags, qoa_multi **out);
void qoa_multi_free(qoa_multi *m);
int qoa_multi_shard(size_t n_clips, int n_devices, int index, size_t *first, size_t *count);
int qoa_multi_encode(qoa_multi *m, const qoa_clip *clips, size_t n_clips, const int16_t *const *pcm, uint8_t *out,
                     uint32_t chunk_frames);
int qoa_multi_decode(qoa_multi *m, const qoa_clip *clips, size_t n_clips, const uint8_t *enc, int16_t *const *pcm,
                     uint32_t chunk_frames);
/* one-shot forms (the handle lives for the call) */
int qoa_encode_batch_multi(const int *devices, int n_devices, const qoa_clip *clips, size_t n_clips,
                           const int16_t *const *pcm, uint8_t *out, uint32_t chunk_frames);
int qoa_decode_batch_multi(const int *devices, int n_devices, const qoa_clip *clips, size_t n_clips, const uint8_t *enc,
                           int16_t *const *pcm, uint32_t chunk_frames);

/* ================================================================== decoder */

/* FrameHeader, src/lib.rs:774-781 */
typedef struct qoa_frame_header {
    uint8_t num_channels;
    uint32_t sample_rate;
    uint16_t num_samples_per_channel;
} qoa_frame_header;

/* ProcessingMode, src/lib.rs:31-45 */
enum { QOA_MODE_FIXED_SAMPLES = 0, QOA_MODE_STREAMING = 1 };
typedef struct qoa_mode {
    int kind;
    uint8_t channels;
    uint32_t sample_rate;
    uint32_t samples;
} qoa_mode;

/* QoaItem (src/lib.rs:656-659) plus the iterator's None / Some(Err) outcomes */
enum { QOA_ITEM_SAMPLE = 0, QOA_ITEM_FRAME_HEADER = 1, QOA_ITEM_END = 2, QOA_ITEM_ERROR = 3 };
typedef struct qoa_item {
    int kind;
    int16_t sample;
    qoa_frame_header header;
    int error; /* QOA_DECODE_* when kind == QOA_ITEM_ERROR */
} qoa_item;

typedef struct qoa_decoder qoa_decoder;

/* QoaDecoder::new over an in-memory reader, src/lib.rs:127-169 (bytes are copied).  A generic
 * io::Read cannot cross a C ABI: the host shim reads the bytes, the GPU decodes whole frames and
 * the iterator drains the decoded PCM in the reference's exact item order. */
int qoa_decoder_new(const uint8_t *bytes, size_t len, qoa_decoder **out);
/* QoaDecoder::new over a LAZY reader, src/lib.rs:117-169 with `R: io::Read`: `read` fills buf with up to len bytes and
 * returns how many it stored -- 0 at the end of the stream, a negative value for an I/O error of the caller's reader
 * (reported as IoError; short reads are retried as read_exact does, :124-126).  new() pulls the 8 header bytes and, in
 * fixed mode, the first frame (the reference parses its header + LMS block inside new(), :161-166); every later frame is
 * pulled -- its 8-byte header, then the LMS block and slice rows that header announces -- only when qoa_decoder_next has
 * handed out the last sample of the frame before (:679-685).  The reference issues the same reads slice row by slice row;
 * here the reader is at most one frame ahead of the items.  `ctx` must stay valid until qoa_decoder_free. */
typedef int64_t (*qoa_read_fn)(void *ctx, uint8_t *buf, size_t len);
int qoa_decoder_new_reader(qoa_read_fn read, void *ctx, qoa_decoder **out);
/* QoaDecoder::new_streaming, src/lib.rs:632-638 */
int qoa_decoder_new_streaming(qoa_decoder **out);
void qoa_decoder_free(qoa_decoder *dec);
/* QoaDecoder::mode, src/lib.rs:180-182 */
int qoa_decoder_mode(const qoa_decoder *dec, qoa_mode *mode);
/* QoaDecoder::current_frame_header, src/lib.rs:186-188 */
int qoa_decoder_current_frame_header(const qoa_decoder *dec, qoa_frame_header *h);
/* QoaDecoder::total_duration, src/lib.rs:190-201: seconds, or a negative value for None (streaming). */
double qoa_decoder_total_duration(const qoa_decoder *dec);
/* Iterator::next, src/lib.rs:668-692.  After QOA_ITEM_ERROR the iterator is finished. */
int qoa_decoder_next(qoa_decoder *dec, qoa_item *item);
/* Bulk form of next(): copies up to cap consecutive Sample items, stops early at the next
 * FrameHeader / END / ERROR (which the following qoa_decoder_next call returns).  Returns the count. */
size_t qoa_decoder_read_samples(qoa_decoder *dec, int16_t *out, size_t cap);
/* QoaDecoder::decode_frame, src/lib.rs:641-651: append bytes, drain.  *n_out = samples produced;
 * if out is NULL or cap too small the samples stay buffered for a second call with len == 0. */
int qoa_decoder_decode_frame(qoa_decoder *dec, const uint8_t *frame, size_t len, int16_t *out, size_t cap,
                             size_t *n_out);

/* decode_all, src/lib.rs:713-747.  Call with out == NULL to obtain *n_out / channels / rate first. */
int qoa_decode_all(const uint8_t *bytes, size_t len, uint8_t *channels, uint32_t *sample_rate, int16_t *out,
                   size_t cap, size_t *n_out);

/* Host-only scan of a stream (no GPU work): walks it exactly as the reference decoder's reader would
 * (src/lib.rs:127-169, :205-289, :320-328) and reports how the iteration over it ends.
 *   *terminal  0 = clean end, else the QOA_DECODE_* error that QoaDecoder::new or the iterator raises;
 *   *in_new    1 if that error is raised by QoaDecoder::new itself (before any item);
 *   *n_frames  FrameHeader items the iterator yields (the streaming mode's leading default header excluded);
 *   *n_samples Sample items it yields. */
typedef struct qoa_stream_info {
    qoa_mode mode;
    int terminal;
    int in_new;
    uint64_t n_frames;
    uint64_t n_samples;
} qoa_stream_info;
int qoa_stream_scan(const uint8_t *bytes, size_t len, qoa_stream_info *info);

/* ---- batch decode on device-resident buffers ---- */

typedef struct qoa_decode_plan qoa_decode_plan;

/* Build the frame table by parsing host copies of the streams exactly like the reference decoder
 * walks them (src/lib.rs:205-289, :320-328).  streams[i] points at len[i] bytes; the device copy of
 * stream i must start at byte in_offset[i] of d_in, and its PCM lands at int16 index out_offset[i]
 * of d_out.  status[i] (optional) receives 0 or the QOA_DECODE_* error that ends stream i;
 * samples_out[i] (optional) the int16 count it yields before that. */
int qoa_decode_plan_from_streams(const uint8_t *const *streams, const size_t *len, const uint64_t *in_offset,
                                 const uint64_t *out_offset, size_t n_streams, int *status, uint64_t *samples_out,
                                 qoa_decode_plan **out);
/* Frame table for streams produced by qoa_encode_plan_run over the same clips (layout is a pure
 * function of the desc, src/lib.rs:374-378); here clip.pcm_offset is the PCM destination and
 * clip.out_offset the encoded source.  The kernel verifies every frame header against the plan. */
int qoa_decode_plan_from_clips(const qoa_clip *clips, size_t n_clips, qoa_decode_plan **out);
void qoa_decode_plan_free(qoa_decode_plan *plan);
int qoa_decode_plan_bytes(const qoa_decode_plan *plan, uint64_t *in_bytes, uint64_t *pcm_bytes,
                          uint64_t *samples_i16);
/* Enqueue the decode on `stream`.  d_in / d_out are DEVICE pointers.  Asynchronous.
 * Layout and speed: full frames (5120 sample-times) whose PCM address is a multiple of 128 bytes decode at the full rate; for 2, 4
 * and 8 channels so do frames off that grid on an 8-byte grid (the kernel's absolute-line variant, ~0.9-1.0x) PROVIDED d_out
 * itself is 128-byte aligned (any hipMalloc is): offsets were classified against it when the plan was built; other dword-aligned
 * frames take write-back stores (about half the rate), short last frames and PCM off the dword grid the general kernel. */
int qoa_decode_plan_run(qoa_decode_plan *plan, const uint8_t *d_in, int16_t *d_out, void *stream);
/* After synchronising: number of frames whose header on the device disagreed with the plan in the LAST run
 * (from_clips plans only; 0 for a well-formed batch). */
int qoa_decode_plan_mismatches(qoa_decode_plan *plan, uint32_t *count);

#ifdef __cplusplus
}
#endif
#endif /* QOAUDIO_AMD_H */
