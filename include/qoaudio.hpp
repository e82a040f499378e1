// qoaudio.hpp -- C++17 host-side mirror of rafaelcaricio/qoaudio's public Rust API, header-only over
// the C ABI in qoaudio_amd.h.  Same names, argument meaning and error behaviour as the crate
// (file:line citations are into the reference's src/lib.rs) so that code and tests written against
// the crate port 1:1:
//
//   Rust                                            C++ (namespace qoaudio)
//   QoaDesc{channels,sample_rate,samples}           QoaDesc{...}
//   QoaEncoder::new(&desc)? / .encode(&pcm)?        QoaEncoder enc(desc); enc.encode(pcm)
//   enc.write_header(&mut w)? / encode_frame(..)?   enc.write_header(out) / enc.encode_frame(pcm, out)
//   encode_all(&pcm, &desc)?                        encode_all(pcm, desc)
//   QoaDecoder::new(reader)? + Iterator             QoaDecoder dec(bytes); while (auto it = dec.next()) ...
//   QoaItem::{Sample, FrameHeader}                  std::variant<Sample, FrameHeader>
//   decode_all(reader)? -> DecodedQoa               decode_all(bytes)
//   Result<_, EncodeError/DecodeError>              throws EncodeError / DecodeError (kind() = the variant)
//
//   (none: one clip at a time)                      HostPipeline p; p.encode(clips, pcm) / p.decode(clips, bytes): batches
//
// Rust's generic io::Read / io::Write cannot cross a C ABI: readers are byte buffers (or std::istream,
// read eagerly), writers are std::vector<uint8_t> / std::ostream.
#pragma once
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <istream>
#include <iterator>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "qoaudio_amd.h"

namespace qoaudio {

// src/lib.rs:12-19
constexpr size_t QOA_SLICE_LEN_ = QOA_SLICE_LEN;
constexpr size_t QOA_FRAME_LEN_ = QOA_FRAME_LEN;
constexpr size_t QOA_MAX_CHANNELS_ = QOA_MAX_CHANNELS;

// src/lib.rs:95-102
struct QoaDesc {
    uint8_t channels;
    uint32_t sample_rate;
    uint32_t samples;
};

// src/lib.rs:106-115
class EncodeError : public std::runtime_error {
public:
    enum Kind { InvalidChannels = 1, InvalidSampleRate = 2, InvalidSamples = 3, IoError = 4 };
    explicit EncodeError(int code)
        : std::runtime_error(std::string(qoa_encode_strerror(code)) + (code == IoError ? qoa_last_error_detail() : "")),
          kind_(static_cast<Kind>(code)) {}
    Kind kind() const { return kind_; }
private:
    Kind kind_;
};

// src/lib.rs:866-873
class DecodeError : public std::runtime_error {
public:
    enum Kind { NotQoaFile = 1, NoSamples = 2, InvalidFrameHeader = 3, IncompatibleFrame = 4, IoError = 5 };
    explicit DecodeError(int code)
        : std::runtime_error(std::string(qoa_decode_strerror(code)) + (code == IoError ? qoa_last_error_detail() : "")),
          kind_(static_cast<Kind>(code)) {}
    Kind kind() const { return kind_; }
private:
    Kind kind_;
};

// src/lib.rs:774-781
struct FrameHeader {
    uint8_t num_channels = 0;
    uint32_t sample_rate = 0;
    uint16_t num_samples_per_channel = 0;
};

// src/lib.rs:31-45
struct ProcessingMode {
    enum Kind { FixedSamples, Streaming } kind;
    uint8_t channels = 0;
    uint32_t sample_rate = 0;
    uint32_t samples = 0;
};

// src/lib.rs:656-659
struct Sample { int16_t value; };
using QoaItem = std::variant<Sample, FrameHeader>;

// src/lib.rs:696-704
struct DecodedQoa {
    uint8_t num_channels;
    uint32_t sample_rate;
    std::vector<int16_t> samples;
};

// src/lib.rs:85-91, :333-439
class QoaEncoder {
public:
    explicit QoaEncoder(const QoaDesc &desc) : desc_(desc) { // QoaEncoder::new, :335-361
        qoa_desc d{desc.channels, desc.sample_rate, desc.samples};
        int rc = qoa_encoder_new(&d, &h_);
        if (rc) throw EncodeError(rc);
    }
    ~QoaEncoder() { qoa_encoder_free(h_); }
    QoaEncoder(const QoaEncoder &) = delete;
    QoaEncoder &operator=(const QoaEncoder &) = delete;

    std::vector<uint8_t> encode(const std::vector<int16_t> &sample_data) { // :367-398
        std::vector<uint8_t> out(qoa_encoded_size(desc_.channels, desc_.samples));
        size_t written = 0;
        int rc = qoa_encoder_encode(h_, sample_data.data(), sample_data.size(), out.data(), out.size(), &written);
        if (rc) throw EncodeError(rc);
        out.resize(written);
        return out;
    }
    void write_header(std::vector<uint8_t> &writer) const { // :404-408
        uint8_t b[8];
        int rc = qoa_encoder_write_header(h_, b);
        if (rc) throw EncodeError(rc);
        writer.insert(writer.end(), b, b + 8);
    }
    // :418-439 -- returns the number of samples per channel encoded in this frame
    size_t encode_frame(const int16_t *sample_data, size_t n_i16, std::vector<uint8_t> &writer) {
        uint8_t buf[8 + 16 * QOA_MAX_CHANNELS + 8 * QOA_SLICES_PER_FRAME * QOA_MAX_CHANNELS];
        size_t written = 0, frame_len = 0;
        int rc = qoa_encoder_encode_frame(h_, sample_data, n_i16, buf, sizeof buf, &written, &frame_len);
        if (rc) throw EncodeError(rc);
        writer.insert(writer.end(), buf, buf + written);
        return frame_len;
    }
    size_t encode_frame(const std::vector<int16_t> &sample_data, std::vector<uint8_t> &writer) {
        return encode_frame(sample_data.data(), sample_data.size(), writer);
    }
private:
    QoaDesc desc_;
    qoa_encoder *h_ = nullptr;
};

// src/lib.rs:760-763
inline std::vector<uint8_t> encode_all(const std::vector<int16_t> &sample_data, const QoaDesc &desc) {
    QoaEncoder enc(desc);
    return enc.encode(sample_data);
}

// src/lib.rs:59-68, :117-201, :619-693
class QoaDecoder {
public:
    explicit QoaDecoder(const std::vector<uint8_t> &bytes) { init(bytes.data(), bytes.size()); } // QoaDecoder::new
    QoaDecoder(const uint8_t *bytes, size_t len) { init(bytes, len); }
    explicit QoaDecoder(std::istream &reader) {
        std::vector<uint8_t> b((std::istreambuf_iterator<char>(reader)), std::istreambuf_iterator<char>());
        init(b.data(), b.size());
    }
    static QoaDecoder open(const std::string &path) { // :619-625
        std::ifstream f(path, std::ios::binary);
        if (!f) throw DecodeError(DecodeError::IoError);
        return QoaDecoder(f);
    }
    static QoaDecoder new_streaming() { // :632-638
        QoaDecoder d;
        int rc = qoa_decoder_new_streaming(&d.h_);
        if (rc) throw DecodeError(rc);
        return d;
    }
    ~QoaDecoder() { qoa_decoder_free(h_); }
    QoaDecoder(QoaDecoder &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    QoaDecoder(const QoaDecoder &) = delete;
    QoaDecoder &operator=(const QoaDecoder &) = delete;

    ProcessingMode mode() const { // :180-182
        qoa_mode m;
        qoa_decoder_mode(h_, &m);
        ProcessingMode r;
        r.kind = m.kind == QOA_MODE_STREAMING ? ProcessingMode::Streaming : ProcessingMode::FixedSamples;
        r.channels = m.channels;
        r.sample_rate = m.sample_rate;
        r.samples = m.samples;
        return r;
    }
    FrameHeader current_frame_header() const { // :186-188
        qoa_frame_header h;
        qoa_decoder_current_frame_header(h_, &h);
        return FrameHeader{h.num_channels, h.sample_rate, h.num_samples_per_channel};
    }
    std::optional<double> total_duration() const { // :190-201 (seconds)
        double d = qoa_decoder_total_duration(h_);
        if (d < 0) return std::nullopt;
        return d;
    }
    // Iterator::next, :668-692: nullopt = None, throws DecodeError for Some(Err(..))
    std::optional<QoaItem> next() {
        qoa_item it;
        qoa_decoder_next(h_, &it);
        switch (it.kind) {
        case QOA_ITEM_SAMPLE: return QoaItem{Sample{it.sample}};
        case QOA_ITEM_FRAME_HEADER:
            return QoaItem{FrameHeader{it.header.num_channels, it.header.sample_rate, it.header.num_samples_per_channel}};
        case QOA_ITEM_END: return std::nullopt;
        default: throw DecodeError(it.error);
        }
    }
    // decode_frame, :641-651
    std::vector<int16_t> decode_frame(const uint8_t *frame_data, size_t len) {
        size_t n = 0;
        int rc = qoa_decoder_decode_frame(h_, frame_data, len, nullptr, 0, &n);
        if (n == 0) {
            if (rc) throw DecodeError(rc);
            return {};
        }
        std::vector<int16_t> out(n);
        rc = qoa_decoder_decode_frame(h_, nullptr, 0, out.data(), out.size(), &n);
        if (rc) throw DecodeError(rc);
        return out;
    }
private:
    QoaDecoder() = default;
    void init(const uint8_t *bytes, size_t len) {
        int rc = qoa_decoder_new(bytes, len, &h_);
        if (rc) throw DecodeError(rc);
    }
    qoa_decoder *h_ = nullptr;
};

// src/lib.rs:713-747
inline DecodedQoa decode_all(const std::vector<uint8_t> &bytes) {
    DecodedQoa d{};
    size_t n = 0;
    int rc = qoa_decode_all(bytes.data(), bytes.size(), &d.num_channels, &d.sample_rate, nullptr, 0, &n);
    if (rc) throw DecodeError(rc);
    d.samples.resize(n);
    if (n) {
        rc = qoa_decode_all(bytes.data(), bytes.size(), &d.num_channels, &d.sample_rate, d.samples.data(), n, &n);
        if (rc) throw DecodeError(rc);
    }
    return d;
}

// src/lib.rs:750-754
inline DecodedQoa open_and_decode_all(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw DecodeError(DecodeError::IoError);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return decode_all(b);
}

// ---- batches (no counterpart in the reference, which is one clip at a time: per clip this IS encode_all / decode_all,
// src/lib.rs:760-763, :713-747) -- the form in which a host with many clips gets the GPU's throughput ----

// One clip of a batch: pcm_offset counts int16 elements from the batch's PCM base, out_offset bytes from its encoded base.
struct BatchClip {
    uint64_t pcm_offset;
    uint64_t out_offset;
    QoaDesc desc;
};

// Contiguous layout of n identical-shape clips: PCM [n][samples][channels], encoded [n][qoa_encoded_size].
inline std::vector<BatchClip> uniform_clips(size_t n, const QoaDesc &desc) {
    std::vector<BatchClip> v(n);
    const uint64_t per_pcm = (uint64_t)desc.samples * desc.channels, per_enc = qoa_encoded_size(desc.channels, desc.samples);
    for (size_t i = 0; i < n; i++) v[i] = BatchClip{i * per_pcm, i * per_enc, desc};
    return v;
}

// qoa_host_pipeline_*: host PCM -> host bytes and back through the time-sliced pipeline (copies overlapped with the kernels);
// a persistent handle that keeps its streams, staging buffers and the tables of its last batch.
class HostPipeline {
public:
    HostPipeline() {
        int rc = qoa_host_pipeline_create(&h_);
        if (rc) throw EncodeError(rc);
    }
    ~HostPipeline() { qoa_host_pipeline_free(h_); }
    HostPipeline(const HostPipeline &) = delete;
    HostPipeline &operator=(const HostPipeline &) = delete;

    // encode_all per clip; `out` is sized to hold every clip's bytes at its out_offset
    std::vector<uint8_t> encode(const std::vector<BatchClip> &clips, const std::vector<int16_t> &pcm, uint32_t chunk_frames = 0) {
        size_t need = 0;
        for (const BatchClip &c : clips) {
            if ((uint64_t)c.desc.samples * c.desc.channels + c.pcm_offset > pcm.size()) throw EncodeError(EncodeError::InvalidSamples);
            need = std::max<size_t>(need, c.out_offset + qoa_encoded_size(c.desc.channels, c.desc.samples));
        }
        std::vector<uint8_t> out(need);
        std::vector<qoa_clip> cc = raw(clips);
        int rc = qoa_host_pipeline_encode(h_, cc.data(), cc.size(), pcm.data(), out.data(), chunk_frames);
        if (rc) throw EncodeError(rc);
        return out;
    }
    // decode_all per clip (every frame header is verified on the device against the clip's desc)
    std::vector<int16_t> decode(const std::vector<BatchClip> &clips, const std::vector<uint8_t> &enc, uint32_t chunk_frames = 0) {
        size_t need = 0;
        for (const BatchClip &c : clips) {
            if (c.out_offset + qoa_encoded_size(c.desc.channels, c.desc.samples) > enc.size()) throw DecodeError(DecodeError::IoError);
            need = std::max<size_t>(need, c.pcm_offset + (uint64_t)c.desc.samples * c.desc.channels);
        }
        std::vector<int16_t> pcm(need);
        std::vector<qoa_clip> cc = raw(clips);
        int rc = qoa_host_pipeline_decode(h_, cc.data(), cc.size(), enc.data(), pcm.data(), chunk_frames);
        if (rc) throw DecodeError(rc);
        return pcm;
    }
private:
    static std::vector<qoa_clip> raw(const std::vector<BatchClip> &clips) {
        std::vector<qoa_clip> cc(clips.size());
        for (size_t i = 0; i < clips.size(); i++)
            cc[i] = qoa_clip{clips[i].pcm_offset, clips[i].out_offset, qoa_desc{clips[i].desc.channels, clips[i].desc.sample_rate, clips[i].desc.samples}};
        return cc;
    }
    qoa_host_pipeline *h_ = nullptr;
};

} // namespace qoaudio
