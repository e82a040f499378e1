// The reference crate's own test-suite (src/lib.rs:991-1298), restated against the C++ mirror in
// include/qoaudio.hpp.  Each test cites the reference test it follows.  Built and run by
// tests/test_cpp_mirror.py; argv[1] = path of the fixture .qoa.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iterator>

#include "qoaudio.hpp"

using namespace qoaudio;

static std::vector<uint8_t> QOA_BYTES;
static int failures = 0;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("  CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            throw std::runtime_error("check failed");                      \
        }                                                                  \
    } while (0)

static void run(const char *name, const std::function<void()> &f) {
    try {
        f();
        std::printf("ok   %s\n", name);
    } catch (const std::exception &e) {
        failures++;
        std::printf("FAIL %s: %s\n", name, e.what());
    }
}

// src/lib.rs:998-1032
static void test_iterating_through_whole_file() {
    QoaDecoder qoa(QOA_BYTES);
    auto m = qoa.mode();
    CHECK(m.kind == ProcessingMode::FixedSamples && m.channels == 2 && m.sample_rate == 44100 && m.samples == 2394122);
    size_t frame_headers_seen = 0, samples_seen = 0;
    while (auto item = qoa.next()) {
        if (std::holds_alternative<Sample>(*item)) {
            samples_seen++;
        } else {
            auto h = std::get<FrameHeader>(*item);
            CHECK(h.num_channels == 2);
            CHECK(h.sample_rate == 44100);
            frame_headers_seen++;
            if (frame_headers_seen < 468) CHECK(h.num_samples_per_channel == 5120);
            else CHECK(h.num_samples_per_channel == 3082);
        }
    }
    CHECK(frame_headers_seen == 468);
    CHECK(samples_seen == 2394122u * 2);
}

// src/lib.rs:1034-1066
static void test_decode_streaming_frames() {
    auto qoa = QoaDecoder::new_streaming();
    CHECK(qoa.mode().kind == ProcessingMode::Streaming);
    auto frame_size_at = [](size_t off) {
        return (size_t)((QOA_BYTES[off + 6] << 8) | QOA_BYTES[off + 7]);
    };
    size_t first_frame_end = 8 + frame_size_at(8);
    auto samples = qoa.decode_frame(QOA_BYTES.data() + 8, first_frame_end - 8);
    CHECK(samples.size() == 5120 * 2);
    size_t second_frame_end = first_frame_end + frame_size_at(first_frame_end);
    samples = qoa.decode_frame(QOA_BYTES.data() + first_frame_end, second_frame_end - first_frame_end);
    CHECK(samples.size() == 5120 * 2);
}

// src/lib.rs:1068-1074
static void test_decode_all() {
    auto decoded = decode_all(QOA_BYTES);
    CHECK(decoded.sample_rate == 44100);
    CHECK(decoded.num_channels == 2);
    CHECK(decoded.samples.size() == 2394122u * 2);
}

// src/lib.rs:1076-1108
static void test_encode_decode_simple() {
    uint32_t sample_rate = 44100;
    uint32_t samples_per_channel = (uint32_t)(sample_rate * 0.1);
    std::vector<int16_t> samples;
    for (uint32_t i = 0; i < samples_per_channel; i++) {
        double t = (double)i / sample_rate;
        samples.push_back((int16_t)(std::sin(t * 440.0 * 2.0 * M_PI) * 16384.0));
    }
    QoaDesc desc{1, sample_rate, samples_per_channel};
    auto encoded = encode_all(samples, desc);
    CHECK(encoded.size() > QOA_HEADER_SIZE);
    auto decoded = decode_all(encoded);
    CHECK(decoded.sample_rate == sample_rate);
    CHECK(decoded.num_channels == 1);
    CHECK(decoded.samples.size() == samples.size());
}

// src/lib.rs:1110-1146
static void test_encode_decode_stereo() {
    uint32_t sample_rate = 44100;
    uint32_t samples_per_channel = (uint32_t)(sample_rate * 0.1);
    std::vector<int16_t> samples;
    for (uint32_t i = 0; i < samples_per_channel; i++) {
        double t = (double)i / sample_rate;
        samples.push_back((int16_t)(std::sin(t * 440.0 * 2.0 * M_PI) * 16384.0));
        samples.push_back((int16_t)(std::cos(t * 440.0 * 2.0 * M_PI) * 16384.0));
    }
    QoaDesc desc{2, sample_rate, samples_per_channel};
    auto encoded = encode_all(samples, desc);
    CHECK(encoded.size() > QOA_HEADER_SIZE);
    auto decoded = decode_all(encoded);
    CHECK(decoded.sample_rate == sample_rate);
    CHECK(decoded.num_channels == 2);
    CHECK(decoded.samples.size() == samples.size());
}

// src/lib.rs:1148-1183
static void test_encoder_errors() {
    std::vector<int16_t> samples(1000, 0);
    auto kind_of = [&](QoaDesc d) {
        try {
            encode_all(samples, d);
        } catch (const EncodeError &e) {
            return (int)e.kind();
        }
        return 0;
    };
    CHECK(kind_of(QoaDesc{0, 44100, 1000}) == EncodeError::InvalidChannels);
    CHECK(kind_of(QoaDesc{1, 0, 1000}) == EncodeError::InvalidSampleRate);
    CHECK(kind_of(QoaDesc{1, 44100, 0}) == EncodeError::InvalidSamples);
}

// src/lib.rs:1185-1230
static void test_round_trip_audio() {
    std::vector<int16_t> samples;
    for (int i = 0; i < 1000; i++) samples.push_back((int16_t)(((i % 200) - 100) * 100));
    QoaDesc desc{1, 44100, 1000};
    auto encoded = encode_all(samples, desc);
    auto decoded = decode_all(encoded);
    CHECK(decoded.sample_rate == 44100 && decoded.num_channels == 1 && decoded.samples.size() == samples.size());
    int max_diff = 0;
    for (size_t i = 0; i < samples.size(); i++) max_diff = std::max(max_diff, std::abs(samples[i] - decoded.samples[i]));
    CHECK(max_diff < 8000);
}

// src/lib.rs:1232-1259
static void test_full_file_round_trip() {
    auto original = decode_all(QOA_BYTES);
    QoaDesc desc{original.num_channels, original.sample_rate, (uint32_t)(original.samples.size() / original.num_channels)};
    auto encoded = encode_all(original.samples, desc);
    auto decoded = decode_all(encoded);
    CHECK(decoded.num_channels == original.num_channels);
    CHECK(decoded.sample_rate == original.sample_rate);
    CHECK(decoded.samples.size() == original.samples.size());
    double mse = 0;
    for (size_t i = 0; i < original.samples.size(); i++) {
        double d = (double)original.samples[i] - (double)decoded.samples[i];
        mse += d * d;
    }
    double rms = std::sqrt(mse / original.samples.size());
    CHECK(rms < 500.0);
}

// src/lib.rs:1261-1297
static void test_streaming_encode() {
    auto original = decode_all(QOA_BYTES);
    size_t channels = original.num_channels;
    uint32_t samples_per_channel = (uint32_t)(original.samples.size() / channels);
    QoaDesc desc{original.num_channels, original.sample_rate, samples_per_channel};
    QoaEncoder oneshot_enc(desc);
    auto oneshot = oneshot_enc.encode(original.samples);
    QoaEncoder streaming_enc(desc);
    std::vector<uint8_t> streamed;
    streaming_enc.write_header(streamed);
    size_t offset = 0, total = samples_per_channel;
    while (offset < total) {
        size_t frame_len = std::min<size_t>(total - offset, QOA_FRAME_LEN);
        size_t n = streaming_enc.encode_frame(original.samples.data() + offset * channels, frame_len * channels, streamed);
        CHECK(n == frame_len);
        offset += frame_len;
    }
    CHECK(streamed == oneshot);
}

// benches/qoa_benchmarks.rs:6-14: item count of the decode bench
static void test_bench_item_count() {
    QoaDecoder decoder(QOA_BYTES);
    size_t count = 0;
    while (decoder.next()) count++;
    CHECK(count == 2394122u * 2 + 468);
}

// (not a reference test: the reference encodes one clip at a time.)  A batch of clips cut from the fixture through the host
// pipeline must equal encode_all per clip (src/lib.rs:760-763) byte for byte, and its decode must equal decode_all of those bytes.
static void test_host_pipeline_batch_equals_per_clip_calls() {
    auto original = decode_all(QOA_BYTES);
    const size_t channels = original.num_channels, n_clips = 12;
    const uint32_t samples = 3 * QOA_FRAME_LEN + 1234;
    QoaDesc desc{original.num_channels, original.sample_rate, samples};
    auto clips = uniform_clips(n_clips, desc);
    std::vector<int16_t> pcm(n_clips * samples * channels);
    for (size_t i = 0; i < n_clips; i++)
        std::memcpy(&pcm[clips[i].pcm_offset], &original.samples[i * 150000 * channels], samples * channels * sizeof(int16_t));
    HostPipeline pipe;
    auto enc = pipe.encode(clips, pcm);
    auto back = pipe.decode(clips, enc);
    const size_t per_enc = qoa_encoded_size(desc.channels, desc.samples);
    CHECK(enc.size() == n_clips * per_enc && back.size() == pcm.size());
    for (size_t i = 0; i < n_clips; i++) {
        std::vector<int16_t> one(pcm.begin() + clips[i].pcm_offset, pcm.begin() + clips[i].pcm_offset + samples * channels);
        auto want = encode_all(one, desc);
        CHECK(want.size() == per_enc && std::memcmp(want.data(), &enc[clips[i].out_offset], per_enc) == 0);
        auto dec = decode_all(want);
        CHECK(dec.samples.size() == one.size() && std::memcmp(dec.samples.data(), &back[clips[i].pcm_offset], one.size() * 2) == 0);
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { std::printf("usage: %s fixture.qoa\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    QOA_BYTES.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    if (QOA_BYTES.size() != 1934040) { std::printf("fixture not found / wrong size\n"); return 2; }
    run("test_iterating_through_whole_file", test_iterating_through_whole_file);
    run("test_decode_streaming_frames", test_decode_streaming_frames);
    run("test_decode_all", test_decode_all);
    run("test_encode_decode_simple", test_encode_decode_simple);
    run("test_encode_decode_stereo", test_encode_decode_stereo);
    run("test_encoder_errors", test_encoder_errors);
    run("test_round_trip_audio", test_round_trip_audio);
    run("test_full_file_round_trip", test_full_file_round_trip);
    run("test_streaming_encode", test_streaming_encode);
    run("test_bench_item_count", test_bench_item_count);
    run("test_host_pipeline_batch_equals_per_clip_calls", test_host_pipeline_batch_equals_per_clip_calls);
    std::printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
