// qoa_format.h -- QOA format constants and tables shared by host and device code.
// Values restate the reference's tables: src/lib.rs:12-27 (consts, reciprocals), :847-864 (dequant).
#pragma once
#include <stdint.h>

namespace qoa {

constexpr int kSliceLen = 20;          // src/lib.rs:12
constexpr int kLmsLen = 4;             // src/lib.rs:13
constexpr int kHeaderSize = 8;         // src/lib.rs:14
constexpr uint32_t kMagic = 0x716f6166u; // "qoaf", src/lib.rs:15
constexpr int kSlicesPerFrame = 256;   // src/lib.rs:16-17
constexpr int kFrameLen = kSlicesPerFrame * kSliceLen; // 5120, src/lib.rs:18
constexpr int kMaxChannels = 8;        // src/lib.rs:19

// src/lib.rs:25-27
#define QOA_RECIPROCALS {65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32}

// Magnitudes of QOA_DEQUANT_TAB[sf][0,2,4,6] (src/lib.rs:847-864); the odd entries are their negations.
#define QOA_DEQUANT_MAGS                                                                              \
    {{1, 3, 5, 7},           {5, 18, 32, 49},        {16, 53, 95, 147},       {34, 113, 203, 315},     \
     {63, 210, 378, 588},    {104, 345, 621, 966},   {158, 528, 950, 1477},   {228, 760, 1368, 2128},  \
     {316, 1053, 1895, 2947}, {422, 1405, 2529, 3934}, {548, 1828, 3290, 5117}, {696, 2320, 4176, 6496}, \
     {868, 2893, 5207, 8099}, {1064, 3548, 6386, 9933}, {1286, 4288, 7718, 12005}, {1536, 5120, 9216, 14336}}

// One encode chain = one (clip, channel): the unit of strictly serial work (src/lib.rs:482).
struct EncChain {
    uint64_t pcm_off;       // int16 index (from the pcm base) of this channel's first sample
    uint64_t out_off;       // byte offset (from the out base) of this call's output for the clip
    uint32_t samples;       // sample-times to encode in this call
    uint32_t hdr_hi;        // (channels << 24) | sample_rate : frame-header bits 63..32 (src/lib.rs:448-449)
    uint32_t total_samples; // QoaDesc.samples for the 8-byte file header (src/lib.rs:382-385)
    uint16_t channel;
    uint8_t flags;          // bit0: write the file header at out_off
    uint8_t pad;
};
static_assert(sizeof(EncChain) == 32, "EncChain layout");

// One decode chain = one (stream, frame, channel): frames re-seed the LMS state (src/lib.rs:270-281).
struct DecChain {
    uint64_t frame_off;  // byte offset of the frame header in the encoded buffer
    uint64_t out_off;    // int16 index of this channel's first output sample
    uint32_t n_out;      // sample-times this frame yields (may be < rows*20: tail, or truncated stream)
    uint16_t rows;       // slice rows to decode (<= 3277: a u16 sample count / 20)
    uint8_t channels;
    uint8_t channel;
    uint32_t expect_hi;  // expected frame-header words (from_clips plans); verify == 0 skips the check
    uint32_t expect_lo;
};
static_assert(sizeof(DecChain) == 32, "DecChain layout");

} // namespace qoa
