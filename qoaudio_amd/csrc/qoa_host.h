// qoa_host.h -- host-side stream walking shared by the decoder handle and the batch decode plan.
//
// This is byte parsing only (no sample arithmetic): it reproduces how the reference decoder moves
// its reader through a stream -- decode_frame_header_and_lms (src/lib.rs:205-289) and the
// slice-row consumption rule of decode_one_slice_per_channel (src/lib.rs:291-330) -- so that the
// GPU can decode every frame of the stream in parallel and the iterator can replay the
// reference's exact item order, including where an error or a clean EOF ends it.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace qoa {

struct FrameRec {
    uint64_t pos;      // byte offset of the 8-byte frame header in the stream
    uint8_t channels;
    uint32_t rate;
    uint16_t nspc;     // num_samples_per_channel from the header
    uint32_t rows;     // slice rows that are fully present (== ceil(nspc/20) unless the stream is cut)
    uint32_t n_out;    // sample-times the iterator yields for this frame before the next event
};

enum Terminal : int {
    kTermEnd = 0,          // clean EOF: Iterator::next returns None
    // otherwise a QOA_DECODE_* code: the Some(Err(..)) that ends the iteration
};

enum WalkMode : int { kFixed = 0, kStreaming = 1 };

struct Walker {
    WalkMode mode = kFixed;
    uint8_t fixed_channels = 0;
    uint32_t fixed_rate = 0;
    uint32_t fixed_samples = 0;
    bool first = true; // no frame header parsed yet (the `first` argument of :205)

    // Walk frames starting at *pos until clean EOF or an error.  Appends to `frames`; returns the
    // terminal (0 = END, >0 = QOA_DECODE_* error).  *pos is left where the reference's reader would be.
    // `err_before_header` semantics: an error raised while parsing a header/LMS block emits no
    // FrameHeader item for that frame (it is simply not appended).
    int walk(const uint8_t *data, size_t len, size_t *pos, std::vector<FrameRec> &frames);
};

} // namespace qoa
