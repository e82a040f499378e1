// qoa_capi.cpp -- the C ABI of include/qoaudio_amd.h: host logic around the gfx950 kernels.
//
// Host responsibilities: validation in the reference's order, the exact output-size formula,
// chain/frame tables, staging between host pointers and HBM, walking decoder streams, and the
// pull-iterator's item order.  All sample arithmetic is in qoa_kernels.hip.  There is no CPU
// compute path: when HIP is unusable every compute entry point returns the IoError code.
#include "../../include/qoaudio_amd.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "qoa_format.h"
#include "qoa_host.h"
#include "qoa_kernels.h"

using namespace qoa;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_detail;

static int fail_hip(hipError_t e, const char *what, int code) {
    g_detail = std::string(what) + ": " + hipGetErrorString(e);
    return code;
}
static int fail_msg(const char *msg, int code) {
    g_detail = msg;
    return code;
}

// No C++ exception may cross the C ABI ("never abort"): the host side keeps its tables in std::vector / std::string,
// whose allocations throw.  Every `int` entry point below is a function-try-block that ends here: the caller gets the family's
// IoError code and the reason in qoa_last_error_detail().  (What a half-built object held is released by its owner's
// destructor where one exists; under memory exhaustion a leaked host table is the lesser evil.)
static int on_exception(int io_code) noexcept {
    try {
        try { throw; }
        catch (const std::bad_alloc &) { g_detail = "out of host memory"; }
        catch (const std::exception &e) { g_detail = std::string("host exception: ") + e.what(); }
        catch (...) { g_detail = "host exception"; }
    } catch (...) { // (even the message could not be stored)
    }
    return io_code;
}

#define HIP_TRY(expr, code)                                          \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return fail_hip(_e, #expr, (code));    \
    } while (0)

extern "C" const char *qoa_last_error_detail(void) { return g_detail.c_str(); }
extern "C" const char *qoa_version(void) { return "qoaudio_amd 0.1.0 gfx950"; }

// Display strings, src/lib.rs:877-906
extern "C" const char *qoa_encode_strerror(int code) {
    switch (code) {
    case QOA_OK: return "Ok";
    case QOA_ENCODE_INVALID_CHANNELS: return "Invalid number of channels (must be 1-8)";
    case QOA_ENCODE_INVALID_SAMPLE_RATE: return "Invalid sample rate (must be > 0)";
    case QOA_ENCODE_INVALID_SAMPLES: return "Invalid number of samples (must be > 0)";
    case QOA_ENCODE_IO_ERROR: return "IO error: ";
    default: return "unknown encode error";
    }
}
extern "C" const char *qoa_decode_strerror(int code) {
    switch (code) {
    case QOA_OK: return "Ok";
    case QOA_DECODE_NOT_QOA_FILE: return "File is not a qoa file";
    case QOA_DECODE_NO_SAMPLES: return "File has no samples";
    case QOA_DECODE_INVALID_FRAME_HEADER: return "File has invalid frame header";
    case QOA_DECODE_INCOMPATIBLE_FRAME: return "Incompatible frame header";
    case QOA_DECODE_IO_ERROR: return "IO error: ";
    default: return "unknown decode error";
    }
}

// ------------------------------------------------------------------------------------------------
// pure format helpers
// ------------------------------------------------------------------------------------------------
extern "C" size_t qoa_encoded_size(uint32_t channels, uint32_t samples) { // src/lib.rs:374-378
    size_t total = samples;
    size_t num_slices = (total + kSliceLen - 1) / kSliceLen;
    size_t num_frames = (total + kFrameLen - 1) / kFrameLen;
    return kHeaderSize + num_frames * (8 + kLmsLen * 4 * (size_t)channels) + num_slices * 8 * (size_t)channels;
}
extern "C" uint16_t qoa_frame_size(uint32_t channels, uint32_t slices) { // src/lib.rs:602-604
    return (uint16_t)(8 + kLmsLen * 4 * channels + 8 * slices * channels);
}

static int validate_desc(const qoa_desc *d) { // src/lib.rs:336-344, in this order
    if (d->channels == 0 || d->channels > kMaxChannels) return QOA_ENCODE_INVALID_CHANNELS;
    if (d->sample_rate == 0) return QOA_ENCODE_INVALID_SAMPLE_RATE;
    if (d->samples == 0) return QOA_ENCODE_INVALID_SAMPLES;
    return QOA_OK;
}

static inline uint32_t be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static inline uint64_t be64(const uint8_t *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }

// Chains per wavefront: 4 (all 64 lanes busy), whatever the launch size.  The kernel is integer-VALU-issue bound and a
// gfx950 SIMD issues these integer ops at one wave-instruction per ~4 cycles whether one or two wavefronts are resident, so
// throughput = 1 / (wave-instructions per SIMD): 4 chains per wavefront is 1.8x faster than 1 once there are more wavefronts
// than SIMDs.  Below that, round 3 spread the chains out (1 per wavefront up to 512 chains, 2 up to 1024) so that chains would
// not wait for their wave-mates' slow stretches -- and shipped a 2x regression: a HELPER workgroup (qoa_kernels.hip) whose two
// wavefronts have only 16 or 32 live lanes issues 15-38 % slower than the same code with 64, varying from compute unit to
// compute unit and launch to launch (not with the data: eight identical chains spread 2333-3231 cycles per slice), and from 512
// such workgroups up half of the compute wavefronts also share a SIMD with another workgroup's helper (2,050 ns per slice
// against 1,000).  Measured on MI355X over 1 .. 1024 chains x 1 / 2 / 3 / 5 / 8 channels (tools/k1lab/small_launch.py,
// profiles/r04_k1lab_log.md): 4 chains per wavefront with helper wavefronts is the fastest packing at EVERY size
// (976-1005 ns per slice; 1 or 2 per wavefront with helpers 1090-2070, without 990-1065).  Clips are handed out longest
// first, so the chains that share a wavefront are of similar length.  QOA_PLAN_CPW_1/2/4 force a packing (tests, lab).
static int pick_cpw(size_t n_chains, uint32_t flags) {
    (void)n_chains;
    int forced = (int)((flags >> 4) & 7u);
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return 4;
}

// ------------------------------------------------------------------------------------------------
// device buffer helper
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(n, 256);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
};

// ------------------------------------------------------------------------------------------------
// batch encode plan
// ------------------------------------------------------------------------------------------------
struct EncGroup {
    int nch = 0;
    int cpw = 1;
    unsigned n = 0;
    DevBuf chains;
    DevBuf lms;
    DevBuf wave_rec; // QOA_PLAN_WAVE_STATS: kWaveRecWords x u32 per compute wavefront, written by the last run
};

#if defined(QOA_LAB_HWID)
constexpr size_t kCounters = 10 + 4096 + 2048 + 2048 * 8; // (+ 8 per-form counts per compute wavefront: tools/k1lab/forms.py) + one placement record per wavefront, cycles per compute wavefront (tools/k1lab/small_launch.py)
#elif defined(QOA_LAB_STAMPS)
constexpr size_t kCounters = 19; // + the timing experiment's (tools/k1lab/stamps.sh)
#else
constexpr size_t kCounters = 10; // kernel diagnostics per plan (qoa_kernels.h: d_stats)
#endif
struct qoa_encode_plan {
    std::vector<EncGroup *> groups;
    uint64_t pcm_bytes = 0, out_bytes = 0, samples = 0;
    uint32_t flags = 0;
    DevBuf stats; // kCounters x u64, accumulated over every run of the plan
    ~qoa_encode_plan() { for (auto *g : groups) delete g; }
};

static void make_chains(const qoa_clip &c, bool file_header, uint32_t samples_now, std::vector<EncChain> &v) {
    for (unsigned ch = 0; ch < c.desc.channels; ch++) {
        EncChain e;
        e.pcm_off = c.pcm_offset + ch;
        e.out_off = c.out_offset;
        e.samples = samples_now;
        e.hdr_hi = ((uint32_t)c.desc.channels << 24) | c.desc.sample_rate; // (ch<<56)|(rate<<32), src/lib.rs:448-449
        e.total_samples = c.desc.samples;
        e.channel = (uint16_t)ch;
        e.flags = file_header ? 1 : 0;
        e.pad = 0;
        v.push_back(e);
    }
}

extern "C" int qoa_encode_plan_create(const qoa_clip *clips, size_t n_clips, uint32_t flags, qoa_encode_plan **out) try {
    if (!out) return fail_msg("null out pointer", QOA_ENCODE_IO_ERROR);
    *out = nullptr;
    std::vector<EncChain> per_nch[kMaxChannels + 1];
    uint64_t pcm_bytes = 0, out_bytes = 0, samples = 0;
    for (size_t i = 0; i < n_clips; i++) {
        int rc = validate_desc(&clips[i].desc);
        if (rc) return rc;
        if (clips[i].out_offset % 8 != 0) return fail_msg("qoa_clip.out_offset must be 8-byte aligned", QOA_ENCODE_IO_ERROR);
        uint64_t n = (uint64_t)clips[i].desc.samples * clips[i].desc.channels;
        samples += n;
        pcm_bytes += n * 2;
        out_bytes += qoa_encoded_size(clips[i].desc.channels, clips[i].desc.samples);
    }
    // Chains are serial and a wavefront runs as long as its longest chain, so clips are handed out longest first:
    // wavefronts hold chains of similar length and the long ones start early.  (Which chains share a wavefront has no
    // effect on the bytes.)  Channels of a clip stay adjacent: they write the same interleaved output lines.
    std::vector<size_t> order(n_clips);
    for (size_t i = 0; i < n_clips; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return clips[a].desc.samples > clips[b].desc.samples; });
    for (size_t i : order) make_chains(clips[i], true, clips[i].desc.samples, per_nch[clips[i].desc.channels]);
    std::unique_ptr<qoa_encode_plan> plan(new (std::nothrow) qoa_encode_plan()); // (released on every way out, a throw included)
    if (!plan) return fail_msg("out of memory", QOA_ENCODE_IO_ERROR);
    plan->groups.reserve(kMaxChannels);
    plan->pcm_bytes = pcm_bytes;
    plan->out_bytes = out_bytes;
    plan->samples = samples;
    plan->flags = flags;
    {
        hipError_t e = plan->stats.reserve(kCounters * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemset(plan->stats.p, 0, kCounters * sizeof(uint64_t));
        if (e != hipSuccess) return fail_hip(e, "encode plan counters", QOA_ENCODE_IO_ERROR);
    }
    for (int nch = 1; nch <= kMaxChannels; nch++) {
        auto &v = per_nch[nch];
        if (v.empty()) continue;
        auto *g = new EncGroup();
        plan->groups.push_back(g); // (capacity reserved above: cannot throw between the two lines)
        g->nch = nch;
        g->n = (unsigned)v.size();
        g->cpw = pick_cpw(v.size(), flags);
        if ((flags & QOA_PLAN_WAVE_STATS) && g->cpw != 4)
            return fail_msg("QOA_PLAN_WAVE_STATS needs the default packing (no QOA_PLAN_CPW_1 / _2)", QOA_ENCODE_IO_ERROR);
        hipError_t e = g->chains.reserve(v.size() * sizeof(EncChain));
        if (e == hipSuccess) e = g->lms.reserve(v.size() * 8 * sizeof(int32_t));
        if (e == hipSuccess && (flags & QOA_PLAN_WAVE_STATS)) {
            const size_t rec_bytes = (size_t)((v.size() + 3) / 4) * kWaveRecWords * sizeof(uint32_t);
            e = g->wave_rec.reserve(rec_bytes);
            if (e == hipSuccess) e = hipMemset(g->wave_rec.p, 0, rec_bytes); // (a plan that never ran reports no wavefronts)
        }
        if (e == hipSuccess) e = hipMemcpy(g->chains.p, v.data(), v.size() * sizeof(EncChain), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail_hip(e, "encode plan upload", QOA_ENCODE_IO_ERROR);
    }
    *out = plan.release();
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" void qoa_encode_plan_free(qoa_encode_plan *plan) { delete plan; }

extern "C" int qoa_encode_plan_bytes(const qoa_encode_plan *plan, uint64_t *pcm_bytes, uint64_t *out_bytes,
                                     uint64_t *samples_i16) try {
    if (!plan) return fail_msg("null plan", QOA_ENCODE_IO_ERROR);
    if (pcm_bytes) *pcm_bytes = plan->pcm_bytes;
    if (out_bytes) *out_bytes = plan->out_bytes;
    if (samples_i16) *samples_i16 = plan->samples;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// The LDS-table step form spends fewer issue slots per slice than the cheap-quotient register form but has one LDS round trip
// per step on its serial chain: it only pays once a launch gives every SIMD about 8 wavefronts to cover that latency
// (measured, stereo clips x 32 frames: 4096 clips 72 vs 78 G samples/s, 8192: 81.5 vs 85.1, 16384: 91.9 vs 89.5 --
// tools/k1lab/forms.py).
static bool plan_lds_form(uint32_t flags, const EncGroup *g) {
    if (flags & QOA_PLAN_LDS_TABLES) return true;
    return !(flags & QOA_PLAN_NO_LDS_TABLES) && (g->n + g->cpw - 1) / g->cpw >= 8192u;
}

static unsigned plan_kernel_flags(uint32_t pf) {
    return ((pf & QOA_PLAN_FORCE_GENERIC_PATH) ? 1u : 0u) | ((pf & QOA_PLAN_FORCE_32BIT_STEP) ? 4u : 0u) |
           ((pf & QOA_PLAN_NO_SPECULATION) ? 8u : 0u) | ((pf & QOA_PLAN_EXACT_QUOTIENT) ? 32u : 0u) |
           ((pf & QOA_PLAN_NO_PENALTY_FORM) ? 64u : 0u) | ((pf & QOA_PLAN_NO_BLOCK32) ? 128u : 0u) |
           ((pf & QOA_PLAN_NO_HELPER) ? 256u : 0u) | ((pf & QOA_PLAN_FORCE_HELPER) ? 512u : 0u) |
           ((pf & QOA_PLAN_NO_WIDE_FORM) ? 1024u : 0u);
}

extern "C" int qoa_encode_plan_launch_shape(const qoa_encode_plan *plan, qoa_launch_shape *shape) try {
    if (!plan || !shape) return fail_msg("null plan / shape", QOA_ENCODE_IO_ERROR);
    if (shape->struct_bytes < sizeof(qoa_launch_shape)) return fail_msg("qoa_launch_shape.struct_bytes too small", QOA_ENCODE_IO_ERROR);
    uint32_t waves = 0, cpw = 0, lds = 0, helper = 0, biggest = 0, launches = 0;
    for (auto *g : plan->groups) {
        const uint32_t w = (g->n + g->cpw - 1) / g->cpw;
        waves += w;
        launches++;
        if (w > biggest) {
            biggest = w; cpw = (uint32_t)g->cpw; lds = plan_lds_form(plan->flags, g) ? 1u : 0u;
            helper = encode_launch_has_helper(g->cpw, g->n, plan_kernel_flags(plan->flags)) ? 1u : 0u;
        }
    }
    shape->wavefronts = waves;
    shape->chains_per_wavefront = cpw;
    shape->lds_table_form = lds;
    shape->helper_wavefronts = helper;
    shape->launches = launches;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// (the four-pointer form of rounds 1-3; round 4 had added a fifth out-parameter in place -- an ABI break for C callers,
// advisor round 4 -- which now lives in qoa_launch_shape)
extern "C" int qoa_encode_plan_launch_info(const qoa_encode_plan *plan, uint32_t *wavefronts, uint32_t *chains_per_wavefront,
                                           uint32_t *lds_table_form) try {
    qoa_launch_shape sh;
    sh.struct_bytes = sizeof sh;
    int rc = qoa_encode_plan_launch_shape(plan, &sh);
    if (rc) return rc;
    if (wavefronts) *wavefronts = sh.wavefronts;
    if (chains_per_wavefront) *chains_per_wavefront = sh.chains_per_wavefront;
    if (lds_table_form) *lds_table_form = sh.lds_table_form;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// kernel flags: bit0 rolled tail path for every slice, bit1 fresh LMS state (QoaEncoder::new's, src/lib.rs:346-352),
// bit2 no i16-packed fast path, bit3 no speculative (penalty-free) steps, bit4 speculative steps on the LDS quantiser table,
// bit5 no cheap-quotient speculative steps, bit6 no penalty form, bit7 no 32-bit block statement,
// bit8 no helper wavefronts, bit9 helper wavefronts whatever the launch size, bit10 no wide form
extern "C" int qoa_encode_plan_run(qoa_encode_plan *plan, const int16_t *d_pcm, uint8_t *d_out, void *stream) try {
    if (!plan) return fail_msg("null plan", QOA_ENCODE_IO_ERROR);
    const unsigned kflags = 2u | plan_kernel_flags(plan->flags);
    for (auto *g : plan->groups) {
        const unsigned lds = plan_lds_form(plan->flags, g) ? 16u : 0u;
        if (g->wave_rec.p) // (records of the LAST run: zeroed on the same stream, ahead of the launch that writes them)
            HIP_TRY(hipMemsetAsync(g->wave_rec.p, 0, (size_t)((g->n + 3) / 4) * kWaveRecWords * sizeof(uint32_t), (hipStream_t)stream),
                    QOA_ENCODE_IO_ERROR);
        HIP_TRY(launch_encode(g->cpw, g->nch, d_pcm, d_out, (const EncChain *)g->chains.p, g->n, (int32_t *)g->lms.p,
                              kflags | lds, (plan->flags & QOA_PLAN_COUNTERS) ? (unsigned long long *)plan->stats.p : nullptr,
                              (hipStream_t)stream, (unsigned *)g->wave_rec.p),
                QOA_ENCODE_IO_ERROR);
    }
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_encode_plan_stats(qoa_encode_plan *plan, uint64_t *declined_slices, uint64_t *unpacked_slices,
                                     uint64_t *speculation_misses) try {
    if (!plan) return fail_msg("null plan", QOA_ENCODE_IO_ERROR);
    if (!(plan->flags & QOA_PLAN_COUNTERS)) return fail_msg("plan was created without QOA_PLAN_COUNTERS", QOA_ENCODE_IO_ERROR);
    uint64_t h[3] = {0, 0, 0};
    HIP_TRY(hipDeviceSynchronize(), QOA_ENCODE_IO_ERROR); // runs may be in flight on any stream; this is a diagnostic
    HIP_TRY(hipMemcpy(h, plan->stats.p, sizeof h, hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
    if (declined_slices) *declined_slices = h[0];
    if (unpacked_slices) *unpacked_slices = h[1];
    if (speculation_misses) *speculation_misses = h[2];
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_encode_plan_counters(qoa_encode_plan *plan, uint64_t *out, size_t n) try {
    if (!plan || !out) return fail_msg("null plan / out", QOA_ENCODE_IO_ERROR);
    if (!(plan->flags & QOA_PLAN_COUNTERS)) return fail_msg("plan was created without QOA_PLAN_COUNTERS", QOA_ENCODE_IO_ERROR);
    std::vector<uint64_t> h(kCounters);
    HIP_TRY(hipDeviceSynchronize(), QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpy(h.data(), plan->stats.p, kCounters * sizeof(uint64_t), hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
    for (size_t i = 0; i < n; i++) out[i] = i < kCounters ? h[i] : 0;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// Issue slots per slice of each step form's block loop as tools/gen_slice_asm.py generates it (the cheap-quotient
// loop 519, exact-quotient speculative 598 (the LDS-table form 477 + 26 LDS reads is reported in the same column), penalty 586,
// wide 664, full packed 784, 32-bit block statement 842, the one-slice statements ~1,100 with their ds_bpermute exchange), and
// what a lone wavefront pays per issued instruction (tools/ubench/execmask.hip: 4.06 cycles).  A slice therefore costs between
// 519 x 4.06 and 842 x 4.06 cycles in the block loops: clipped material, whose chains keep their weights beyond the i16
// window, is bounded by the 32-bit statement's rate, 1.62x the cheap form's.
static const uint32_t kFormIssueSlots[QOA_WAVE_FORMS] = {519, 598, 586, 664, 784, 842, 1100, 0};
static const double kCyclesPerIssue = 4.06;
// what a block loop costs beyond its steps' issue slots -- the per-block barrier and dispatch, the taken loop branch, the
// winner's LDS round trip: measured 5 % (32-bit statement) to 10 % (cheap-quotient form) of a slice (profiles/r05_k1_slack.md)
static const double kLoopOverhead = 1.10;

extern "C" int qoa_encode_plan_wave_stats(qoa_encode_plan *plan, qoa_wave_stats *out) try {
    if (!plan || !out) return fail_msg("null plan / out", QOA_ENCODE_IO_ERROR);
    if (!(plan->flags & QOA_PLAN_WAVE_STATS)) return fail_msg("plan was created without QOA_PLAN_WAVE_STATS", QOA_ENCODE_IO_ERROR);
    if (out->struct_bytes < sizeof(qoa_wave_stats)) return fail_msg("qoa_wave_stats.struct_bytes too small", QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipDeviceSynchronize(), QOA_ENCODE_IO_ERROR); // the run may be in flight on any stream; this is a diagnostic
    const uint32_t keep = out->struct_bytes;
    memset(out, 0, sizeof *out);
    out->struct_bytes = keep;
    for (int k = 0; k < QOA_WAVE_FORMS; k++) out->issue_slots[k] = kFormIssueSlots[k];
    out->cycles_per_issue = kCyclesPerIssue;
    out->loop_overhead = kLoopOverhead;
    out->bound_cycles_per_slice = kFormIssueSlots[5] * kCyclesPerIssue * kLoopOverhead;
    std::vector<double> cps;
    double worst = -1.0;
    uint32_t index = 0;
    for (auto *g : plan->groups) {
        const size_t waves = (g->n + 3) / 4;
        std::vector<uint32_t> h(waves * kWaveRecWords);
        HIP_TRY(hipMemcpy(h.data(), g->wave_rec.p, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
        for (size_t w = 0; w < waves; w++, index++) {
            const uint32_t *r = &h[w * kWaveRecWords];
            if (!r[kWaveRecForms + 3] || !r[kWaveRecForms + 2]) continue; // no record (an empty time slice), or no slices
            out->wavefronts++;
            for (int k = 0; k < QOA_WAVE_FORMS; k++) out->slices[k] += r[k];
            const double cycles = (double)(((uint64_t)r[kWaveRecForms + 1] << 32) | r[kWaveRecForms]);
            const double per = cycles / (double)r[kWaveRecForms + 2];
            cps.push_back(per);
            if (per > worst) {
                worst = per;
                out->slowest_wavefront = index;
                out->slowest_channels = (uint32_t)g->nch;
                out->slowest_slices_of_chain = r[kWaveRecForms + 2];
                double pred = 0;
                for (int k = 0; k < QOA_WAVE_FORMS; k++) {
                    out->slowest_slices[k] = r[k];
                    pred += (double)r[k] * kFormIssueSlots[k] * kCyclesPerIssue * kLoopOverhead;
                }
                out->slowest_predicted_cycles_per_slice = pred / (double)r[kWaveRecForms + 2];
            }
        }
    }
    if (!cps.empty()) {
        std::sort(cps.begin(), cps.end());
        out->cycles_per_slice_p50 = cps[cps.size() / 2];
        out->cycles_per_slice_max = cps.back();
        out->cycles_per_slice_min = cps.front();
    }
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// Host-memory batch encode: time-sliced pipeline
//
// (The copies in and out use hipMemcpyDefault: `h_pcm` / `h_out` may be any pointer of the unified address space -- host
// memory, pinned or not, memory of this device, or memory of a PEER device.  qoa_multi_* below relies on that: every
// device of a multi-GPU batch runs this pipeline on its shard and its stage-out IS the delivery to the root device.)
//
// A batch whose PCM lives in host memory cannot hide PCIe behind the kernel by splitting it into groups of
// clips: the kernel's duration is set by the serial chain, not by the number of clips.  It CAN be split along
// time: the kernel carries every chain's LMS state in device memory (the streaming-encoder mechanism), so
// frames [j*F, (j+1)*F) of ALL clips are one launch that still fills the chip, while the copy engines move
// slice j+1 in and slice j-1 out.  Output bytes are identical to encode_all per clip.
// ------------------------------------------------------------------------------------------------
// Chain tables of a pipelined batch, kept between calls of a persistent pipeline: all time slices' tables in ONE device
// buffer, keyed on what they were derived from (the clip list, the slice length, which sides are used in place).  A
// repeated batch -- qoa_multi_encode on the same clip list, a bench loop -- uploads nothing and builds nothing.
struct PipeKey {
    std::vector<uint8_t> clips; // raw bytes of the qoa_clip array
    uint32_t F = 0;
    bool src_direct = false, dst_direct = false, valid = false;
    bool matches(const qoa_clip *c, size_t n, uint32_t f, bool s, bool d) const {
        return valid && F == f && src_direct == s && dst_direct == d && clips.size() == n * sizeof(qoa_clip) &&
               memcmp(clips.data(), c, clips.size()) == 0;
    }
    void set(const qoa_clip *c, size_t n, uint32_t f, bool s, bool d) {
        clips.assign((const uint8_t *)c, (const uint8_t *)c + n * sizeof(qoa_clip));
        F = f; src_direct = s; dst_direct = d; valid = true;
    }
};

struct qoa_host_pipeline {
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    hipEvent_t in_done[2] = {nullptr, nullptr}, k_done[2] = {nullptr, nullptr}, out_done[2] = {nullptr, nullptr};
    DevBuf pcm[2], enc[2], lms; // device staging, grown on demand and kept between calls
    DevBuf enc_tables;          // [time slice][chain]: EncChain tables of the batch enc_key describes
    PipeKey enc_key, dec_key;
    unsigned enc_group_begin[kMaxChannels + 2] = {0};
    struct DecTablesHolder *dec = nullptr;  // decode tables (one set per time slice), defined further down
    // qoa_multi debug mode (QOA_MULTI_ALLOW_DUPLICATE_DEVICES): treat the batch's shared buffer (`out` of an encode, `enc` of
    // a decode) as memory of ANOTHER device even though it is this one's -- the shard then takes the staging + copy path a
    // second GPU takes, on a box that has one
    bool shared_side_is_remote = false;
    bool ready = false;
    hipError_t init() {
        if (ready) return hipSuccess;
        hipError_t e = hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking);
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            e = hipEventCreateWithFlags(&in_done[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&k_done[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&out_done[b], hipEventDisableTiming);
        }
        ready = (e == hipSuccess);
        return e;
    }
    ~qoa_host_pipeline();
};
typedef qoa_host_pipeline HostPipe;

// Whatever path a pipeline call leaves by (an error in the middle included), no copy into or out of the caller's host
// buffers may still be in flight when it returns: the caller is free to release them.
struct PipeDrain {
    HostPipe &p;
    ~PipeDrain() {
        if (p.s_in) (void)hipStreamSynchronize(p.s_in);
        if (p.s_k) (void)hipStreamSynchronize(p.s_k);
        if (p.s_out) (void)hipStreamSynchronize(p.s_out);
    }
};

// Is `p` memory of the calling thread's current device?  (Then a pipeline stage that would only copy it to a staging
// buffer on the same device is skipped: the kernels read / write it in place.)  Plain host memory is unknown to the runtime.
static bool on_current_device(const void *p) {
    hipPointerAttribute_t a;
    int cur = -1;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipGetDevice(&cur) != hipSuccess) return false;
    return a.type == hipMemoryTypeDevice && a.device == cur;
}

// Device memory handed to a pipeline is read / written by kernels on the pipeline's own (non-blocking) streams, which are
// not ordered against whatever stream produced it (a memset, a generator kernel, a copy): the call first waits for
// everything submitted to the device so far.  (Host memory is the caller's to have finished with, as for any memcpy.)
static hipError_t order_after_device_work() { return hipDeviceSynchronize(); }

static int host_pipe_encode(HostPipe &P, const qoa_clip *clips, size_t n_clips, const int16_t *h_pcm, uint8_t *h_out,
                            uint32_t chunk_frames) {
    if (!clips || !h_pcm || !h_out) return fail_msg("null pointer", QOA_ENCODE_IO_ERROR);
    if (n_clips == 0) return QOA_OK;
    uint32_t max_frames = 0;
    for (size_t i = 0; i < n_clips; i++) {
        int rc = validate_desc(&clips[i].desc);
        if (rc) return rc;
        max_frames = std::max<uint32_t>(max_frames, (clips[i].desc.samples + kFrameLen - 1) / kFrameLen);
    }
    // qoa_multi_*: a shard whose PCM / output already sits in this device's memory is used in place (no staging copy in /
    // out); with both in place there is nothing to overlap and the batch is one launch
    const bool src_direct = on_current_device(h_pcm);
    bool dst_direct = on_current_device(h_out) && !P.shared_side_is_remote;
    const bool device_memory = src_direct || on_current_device(h_out);
    if (dst_direct) { // the kernel would store 8-byte words straight into the caller's buffer: any odd offset sends the
                      // batch through the staging buffers instead (as it would go for host memory)
        if ((uintptr_t)h_out % 8 != 0) dst_direct = false;
        for (size_t i = 0; i < n_clips && dst_direct; i++)
            if (clips[i].out_offset % 8 != 0) dst_direct = false;
    }
    uint32_t F = chunk_frames ? chunk_frames : 8; // config 3: 102.1 ms at 8 or 4 frames, 104.9 at 16, 111.3 at 32
    if (src_direct && dst_direct) F = max_frames;
    const uint32_t n_chunks = (max_frames + F - 1) / F;

    // device staging layout: per clip a PCM region of F frames and an encoded region of 8 + F frames, per buffer
    std::vector<uint64_t> pcm_reg(n_clips + 1, 0), enc_reg(n_clips + 1, 0);
    bool uniform = true;
    for (size_t i = 0; i < n_clips; i++) {
        const qoa_desc &d = clips[i].desc;
        pcm_reg[i + 1] = pcm_reg[i] + (uint64_t)F * kFrameLen * d.channels;                 // int16 elements
        enc_reg[i + 1] = enc_reg[i] + 8 + (uint64_t)F * qoa_frame_size(d.channels, kSlicesPerFrame); // bytes, 8-aligned
        if (i > 0) {
            const qoa_desc &p = clips[i - 1].desc;
            uniform = uniform && d.channels == p.channels && d.samples == p.samples &&
                      clips[i].pcm_offset - clips[i - 1].pcm_offset == clips[1].pcm_offset - clips[0].pcm_offset &&
                      clips[i].out_offset - clips[i - 1].out_offset == clips[1].out_offset - clips[0].out_offset;
        }
    }
    // the 2-D copies below take the offset differences as pitches: they must be ascending and at least a row wide
    // (equally spaced DESCENDING offsets, or overlapping rows, pass the equal-difference test above as well)
    uniform = uniform && n_clips > 1 && clips[1].pcm_offset > clips[0].pcm_offset && clips[1].out_offset > clips[0].out_offset &&
              clips[1].pcm_offset - clips[0].pcm_offset >= (uint64_t)clips[0].desc.samples * clips[0].desc.channels &&
              clips[1].out_offset - clips[0].out_offset >= qoa_encoded_size(clips[0].desc.channels, clips[0].desc.samples);

    HIP_TRY(P.init(), QOA_ENCODE_IO_ERROR);
    if (device_memory) HIP_TRY(order_after_device_work(), QOA_ENCODE_IO_ERROR);
    PipeDrain drain{P};
    size_t n_chains_total = 0;
    for (size_t i = 0; i < n_clips; i++) n_chains_total += clips[i].desc.channels;
    for (int b = 0; b < 2; b++) {
        if (!src_direct) HIP_TRY(P.pcm[b].reserve(pcm_reg[n_clips] * 2), QOA_ENCODE_IO_ERROR);
        if (!dst_direct) HIP_TRY(P.enc[b].reserve(enc_reg[n_clips]), QOA_ENCODE_IO_ERROR);
    }
    HIP_TRY(P.lms.reserve(n_chains_total * 8 * sizeof(int32_t)), QOA_ENCODE_IO_ERROR);

    // chains are ordered by channel count (one launch per count), the order is the same in every time slice so
    // that chain ids -- and with them the carried LMS states -- are stable
    std::vector<size_t> by_nch[kMaxChannels + 1];
    for (size_t i = 0; i < n_clips; i++) by_nch[clips[i].desc.channels].push_back(i);
    auto samples_now = [&](const qoa_desc &d, uint32_t j) {
        const uint64_t first = (uint64_t)j * F * kFrameLen;
        return d.samples > first ? (uint32_t)std::min<uint64_t>(d.samples - first, (uint64_t)F * kFrameLen) : 0u;
    };

    // ---- chain tables of all time slices: built and uploaded once per (clip list, slice length, in-place sides) ----
    if (!P.enc_key.matches(clips, n_clips, F, src_direct, dst_direct)) {
        P.enc_key.valid = false;
        std::vector<EncChain> all;
        all.reserve((size_t)n_chunks * n_chains_total);
        for (uint32_t j = 0; j < n_chunks; j++) {
            unsigned gb = 0;
            for (int nch = 1; nch <= kMaxChannels; nch++) {
                P.enc_group_begin[nch] = gb;
                for (size_t i : by_nch[nch]) {
                    const qoa_desc &d = clips[i].desc;
                    const uint64_t first = (uint64_t)j * F * kFrameLen;
                    qoa_clip c;
                    c.pcm_offset = src_direct ? clips[i].pcm_offset + first * d.channels : pcm_reg[i];
                    // in place the slice continues where the one before ended: file header + the frames so far
                    c.out_offset = !dst_direct ? enc_reg[i]
                                   : clips[i].out_offset + (j == 0 ? 0 : 8 + (uint64_t)j * F * qoa_frame_size(d.channels, kSlicesPerFrame));
                    c.desc = d;
                    make_chains(c, j == 0, samples_now(d, j), all);
                    gb += d.channels;
                }
            }
            P.enc_group_begin[kMaxChannels + 1] = gb;
        }
        HIP_TRY(P.enc_tables.reserve(all.size() * sizeof(EncChain)), QOA_ENCODE_IO_ERROR);
        HIP_TRY(hipStreamSynchronize(P.s_k), QOA_ENCODE_IO_ERROR); // (no launch of an earlier call may still read the old tables)
        HIP_TRY(hipMemcpy(P.enc_tables.p, all.data(), all.size() * sizeof(EncChain), hipMemcpyHostToDevice), QOA_ENCODE_IO_ERROR);
        P.enc_key.set(clips, n_clips, F, src_direct, dst_direct);
    }
    const unsigned *group_begin = P.enc_group_begin;

    for (uint32_t j = 0; j < n_chunks + 1; j++) {
        const int b = (int)(j & 1u);
        if (j < n_chunks) {
            // ---- stage in: PCM of time slice j (buffer b is free once the kernel of slice j-2 has run) ----
            if (!src_direct) {
                if (j >= 2) HIP_TRY(hipStreamWaitEvent(P.s_in, P.k_done[b], 0), QOA_ENCODE_IO_ERROR);
                if (uniform) {
                    const qoa_desc &d = clips[0].desc;
                    const uint64_t first = (uint64_t)j * F * kFrameLen;
                    HIP_TRY(hipMemcpy2DAsync(P.pcm[b].p, (pcm_reg[1] - pcm_reg[0]) * 2, h_pcm + clips[0].pcm_offset + first * d.channels,
                                             (clips[1].pcm_offset - clips[0].pcm_offset) * 2, (size_t)samples_now(d, j) * d.channels * 2,
                                             n_clips, hipMemcpyDefault, P.s_in),
                            QOA_ENCODE_IO_ERROR);
                } else {
                    for (size_t i = 0; i < n_clips; i++) {
                        const qoa_desc &d = clips[i].desc;
                        const uint64_t first = (uint64_t)j * F * kFrameLen;
                        const uint32_t now = samples_now(d, j);
                        if (now)
                            HIP_TRY(hipMemcpyAsync((int16_t *)P.pcm[b].p + pcm_reg[i], h_pcm + clips[i].pcm_offset + first * d.channels,
                                                   (size_t)now * d.channels * 2, hipMemcpyDefault, P.s_in),
                                    QOA_ENCODE_IO_ERROR);
                    }
                }
                HIP_TRY(hipEventRecord(P.in_done[b], P.s_in), QOA_ENCODE_IO_ERROR);
                HIP_TRY(hipStreamWaitEvent(P.s_k, P.in_done[b], 0), QOA_ENCODE_IO_ERROR);
            }

            // ---- encode time slice j (its output buffer is free once slice j-2 has been copied out) ----
            if (j >= 2 && !dst_direct) HIP_TRY(hipStreamWaitEvent(P.s_k, P.out_done[b], 0), QOA_ENCODE_IO_ERROR);
            const EncChain *table = (const EncChain *)P.enc_tables.p + (size_t)j * n_chains_total;
            for (int nch = 1; nch <= kMaxChannels; nch++) {
                const unsigned n = group_begin[nch + 1 <= kMaxChannels ? nch + 1 : kMaxChannels + 1] - group_begin[nch];
                if (!n) continue;
                HIP_TRY(launch_encode(pick_cpw(n, 0), nch, src_direct ? h_pcm : (const int16_t *)P.pcm[b].p,
                                      dst_direct ? h_out : (uint8_t *)P.enc[b].p, table + group_begin[nch], n,
                                      (int32_t *)P.lms.p + (size_t)group_begin[nch] * 8, j == 0 ? 2u : 0u, nullptr, P.s_k),
                        QOA_ENCODE_IO_ERROR);
            }
            HIP_TRY(hipEventRecord(P.k_done[b], P.s_k), QOA_ENCODE_IO_ERROR);
        }
        if (j >= 1 && !dst_direct) {
            // ---- stage out: encoded bytes of time slice j-1 ----
            const uint32_t jj = j - 1;
            const int bo = (int)(jj & 1u);
            HIP_TRY(hipStreamWaitEvent(P.s_out, P.k_done[bo], 0), QOA_ENCODE_IO_ERROR);
            auto piece = [&](const qoa_desc &d, uint64_t *dst_off, uint64_t *src_off, uint64_t *bytes) {
                const uint64_t first = (uint64_t)jj * F * kFrameLen;
                if (d.samples <= first) { *bytes = 0; return; }
                const uint64_t now = std::min<uint64_t>(d.samples - first, (uint64_t)F * kFrameLen);
                const uint64_t frames_before = (uint64_t)jj * F;
                const uint64_t fb = 8 + 16 * (uint64_t)d.channels + 8 * (uint64_t)kSlicesPerFrame * d.channels;
                const uint64_t full = now / kFrameLen, rest = now % kFrameLen;
                *bytes = full * fb + (rest ? 8 + 16 * (uint64_t)d.channels + 8 * ((rest + kSliceLen - 1) / kSliceLen) * d.channels : 0);
                *dst_off = 8 + frames_before * fb;
                *src_off = 0;
                if (jj == 0) { *bytes += 8; *dst_off = 0; }
            };
            if (uniform) {
                uint64_t dst, src, bytes;
                piece(clips[0].desc, &dst, &src, &bytes);
                if (bytes)
                    HIP_TRY(hipMemcpy2DAsync(h_out + clips[0].out_offset + dst, clips[1].out_offset - clips[0].out_offset,
                                             (uint8_t *)P.enc[bo].p + src, enc_reg[1] - enc_reg[0], bytes, n_clips,
                                             hipMemcpyDefault, P.s_out),
                            QOA_ENCODE_IO_ERROR);
            } else {
                for (size_t i = 0; i < n_clips; i++) {
                    uint64_t dst, src, bytes;
                    piece(clips[i].desc, &dst, &src, &bytes);
                    if (bytes)
                        HIP_TRY(hipMemcpyAsync(h_out + clips[i].out_offset + dst, (uint8_t *)P.enc[bo].p + enc_reg[i] + src, bytes,
                                               hipMemcpyDefault, P.s_out),
                                QOA_ENCODE_IO_ERROR);
                }
            }
            HIP_TRY(hipEventRecord(P.out_done[bo], P.s_out), QOA_ENCODE_IO_ERROR);
        }
    }
    HIP_TRY(hipStreamSynchronize(P.s_out), QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipStreamSynchronize(P.s_k), QOA_ENCODE_IO_ERROR);
    return QOA_OK;
}

extern "C" int qoa_host_pipeline_create(qoa_host_pipeline **out) try {
    if (!out) return fail_msg("null out pointer", QOA_ENCODE_IO_ERROR);
    *out = nullptr;
    auto *p = new (std::nothrow) qoa_host_pipeline();
    if (!p) return fail_msg("out of memory", QOA_ENCODE_IO_ERROR);
    hipError_t e = p->init();
    if (e != hipSuccess) { delete p; return fail_hip(e, "qoa_host_pipeline_create", QOA_ENCODE_IO_ERROR); }
    *out = p;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" void qoa_host_pipeline_free(qoa_host_pipeline *p) { delete p; }

extern "C" int qoa_host_pipeline_encode(qoa_host_pipeline *p, const qoa_clip *clips, size_t n_clips, const int16_t *h_pcm,
                                        uint8_t *h_out, uint32_t chunk_frames) try {
    if (!p) return fail_msg("null pipeline", QOA_ENCODE_IO_ERROR);
    return host_pipe_encode(*p, clips, n_clips, h_pcm, h_out, chunk_frames);
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// one-shot form: a pipeline that lives for one call (streams, events and staging buffers are set up and torn down)
extern "C" int qoa_encode_batch_host(const qoa_clip *clips, size_t n_clips, const int16_t *h_pcm, uint8_t *h_out,
                                     uint32_t chunk_frames) try {
    qoa_host_pipeline P;
    return host_pipe_encode(P, clips, n_clips, h_pcm, h_out, chunk_frames);
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// QoaEncoder handle
// ------------------------------------------------------------------------------------------------
struct qoa_encoder {
    qoa_desc desc{};
    uint32_t debug_plan_flags = 0; // qoa_debug_encoder_set_plan_flags: QOA_PLAN_* step-form switches for the handle's launches
    hipStream_t stream = nullptr;
    DevBuf lms, chains, pcm, out;
    ~qoa_encoder() { if (stream) (void)hipStreamDestroy(stream); }
};

extern "C" int qoa_encoder_new(const qoa_desc *desc, qoa_encoder **out) try {
    if (!out || !desc) return fail_msg("null pointer", QOA_ENCODE_IO_ERROR);
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    auto *e = new (std::nothrow) qoa_encoder();
    if (!e) return fail_msg("out of memory", QOA_ENCODE_IO_ERROR);
    e->desc = *desc;
    // initial LMS state: history 0, weights [0, 0, -(1<<13), 1<<14]  (src/lib.rs:346-352)
    int32_t init[kMaxChannels * 8];
    for (int c = 0; c < kMaxChannels; c++) {
        int32_t *s = init + c * 8;
        s[0] = s[1] = s[2] = s[3] = 0;
        s[4] = 0; s[5] = 0; s[6] = -(1 << 13); s[7] = 1 << 14;
    }
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = e->lms.reserve(sizeof init);
    if (err == hipSuccess) err = e->chains.reserve(kMaxChannels * sizeof(EncChain));
    if (err == hipSuccess) err = hipMemcpy(e->lms.p, init, sizeof init, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        delete e;
        return fail_hip(err, "qoa_encoder_new", QOA_ENCODE_IO_ERROR);
    }
    *out = e;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" void qoa_encoder_free(qoa_encoder *enc) { delete enc; }

extern "C" int qoa_encoder_write_header(const qoa_encoder *enc, uint8_t out[8]) try { // src/lib.rs:404-408
    if (!enc || !out) return fail_msg("null pointer", QOA_ENCODE_IO_ERROR);
    out[0] = 'q'; out[1] = 'o'; out[2] = 'a'; out[3] = 'f';
    uint32_t s = enc->desc.samples;
    out[4] = (uint8_t)(s >> 24); out[5] = (uint8_t)(s >> 16); out[6] = (uint8_t)(s >> 8); out[7] = (uint8_t)s;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// Shared by encode() and encode_frame(): encode `frame_len_total` sample-times of the handle's stream
// starting from the carried LMS state, with or without the 8-byte file header.
static int encoder_run(qoa_encoder *enc, const int16_t *pcm, size_t n_i16, uint32_t sample_times, bool file_header,
                       uint8_t *out, size_t out_bytes) {
    qoa_clip clip;
    clip.pcm_offset = 0;
    clip.out_offset = 0;
    clip.desc = enc->desc;
    std::vector<EncChain> v;
    make_chains(clip, file_header, sample_times, v);
    HIP_TRY(enc->pcm.reserve(n_i16 * 2), QOA_ENCODE_IO_ERROR);
    HIP_TRY(enc->out.reserve(out_bytes), QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpyAsync(enc->chains.p, v.data(), v.size() * sizeof(EncChain), hipMemcpyHostToDevice, enc->stream),
            QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpyAsync(enc->pcm.p, pcm, n_i16 * 2, hipMemcpyHostToDevice, enc->stream), QOA_ENCODE_IO_ERROR);
    HIP_TRY(launch_encode(pick_cpw(v.size(), enc->debug_plan_flags), enc->desc.channels, (const int16_t *)enc->pcm.p,
                          (uint8_t *)enc->out.p, (const EncChain *)enc->chains.p, (unsigned)v.size(), (int32_t *)enc->lms.p,
                          plan_kernel_flags(enc->debug_plan_flags) | ((enc->debug_plan_flags & QOA_PLAN_LDS_TABLES) ? 16u : 0u),
                          nullptr, enc->stream),
            QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpyAsync(out, enc->out.p, out_bytes, hipMemcpyDeviceToHost, enc->stream), QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipStreamSynchronize(enc->stream), QOA_ENCODE_IO_ERROR);
    return QOA_OK;
}

extern "C" int qoa_encoder_encode(qoa_encoder *enc, const int16_t *pcm, size_t n_i16, uint8_t *out, size_t cap,
                                  size_t *written) try { // src/lib.rs:367-398
    if (!enc) return fail_msg("null encoder", QOA_ENCODE_IO_ERROR);
    size_t channels = enc->desc.channels, total = enc->desc.samples;
    if (n_i16 != total * channels) return QOA_ENCODE_INVALID_SAMPLES; // :368-370
    size_t need = qoa_encoded_size((uint32_t)channels, (uint32_t)total);
    if (!out || cap < need) return fail_msg("output buffer smaller than qoa_encoded_size()", QOA_ENCODE_IO_ERROR);
    if (!pcm) return fail_msg("null pcm", QOA_ENCODE_IO_ERROR);
    int rc = encoder_run(enc, pcm, n_i16, (uint32_t)total, true, out, need);
    if (rc) return rc;
    if (written) *written = need;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_encoder_encode_frame(qoa_encoder *enc, const int16_t *pcm, size_t n_i16, uint8_t *out, size_t cap,
                                        size_t *written, size_t *frame_len_out) try { // src/lib.rs:418-439
    if (!enc) return fail_msg("null encoder", QOA_ENCODE_IO_ERROR);
    size_t channels = enc->desc.channels;
    if (n_i16 == 0 || n_i16 % channels != 0) return QOA_ENCODE_INVALID_SAMPLES; // :424-426
    size_t frame_len = n_i16 / channels;
    if (frame_len > (size_t)kFrameLen) return QOA_ENCODE_INVALID_SAMPLES; // :428-430
    size_t slices = (frame_len + kSliceLen - 1) / kSliceLen;
    size_t frame_size = qoa_frame_size((uint32_t)channels, (uint32_t)slices);
    if (!out || cap < frame_size) return fail_msg("output buffer smaller than the frame", QOA_ENCODE_IO_ERROR);
    if (!pcm) return fail_msg("null pcm", QOA_ENCODE_IO_ERROR);
    int rc = encoder_run(enc, pcm, n_i16, (uint32_t)frame_len, false, out, frame_size);
    if (rc) return rc;
    if (written) *written = frame_size;
    if (frame_len_out) *frame_len_out = frame_len;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_encoder_get_lms(qoa_encoder *enc, int32_t *state) try {
    if (!enc || !state) return fail_msg("null pointer", QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpy(state, enc->lms.p, enc->desc.channels * 8 * sizeof(int32_t), hipMemcpyDeviceToHost),
            QOA_ENCODE_IO_ERROR);
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }
extern "C" int qoa_encoder_set_lms(qoa_encoder *enc, const int32_t *state) try {
    if (!enc || !state) return fail_msg("null pointer", QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpy(enc->lms.p, state, enc->desc.channels * 8 * sizeof(int32_t), hipMemcpyHostToDevice),
            QOA_ENCODE_IO_ERROR);
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// Test hook: the handle's launches take the step-form switches of a plan (QOA_PLAN_* debug flags; QOA_PLAN_COUNTERS is
// ignored) -- the fuzz tests drive every form from hostile LMS states through set_lms + encode_frame.
extern "C" int qoa_debug_encoder_set_plan_flags(qoa_encoder *enc, uint32_t flags) try {
    if (!enc) return fail_msg("null encoder", QOA_ENCODE_IO_ERROR);
    enc->debug_plan_flags = flags & ~(uint32_t)QOA_PLAN_COUNTERS;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// Test hook: the 16 candidates of ONE slice through the device functions of the encode kernel.
extern "C" int qoa_debug_encode_slice(const int32_t *state, const int16_t *samples, int len, int generic,
                                      uint64_t *total_rank, uint64_t *first_rank, uint64_t *slices,
                                      int32_t *end_states) try {
    if (!state || !samples || len < 1 || len > 20) return fail_msg("bad argument", QOA_ENCODE_IO_ERROR);
    int32_t smp[20] = {0};
    for (int i = 0; i < len; i++) smp[i] = samples[i];
    DevBuf d;
    const size_t o_state = 0, o_smp = 64, o_tot = 256, o_fst = 512, o_slice = 768, o_end = 1024, total = 1024 + 512;
    HIP_TRY(d.reserve(total), QOA_ENCODE_IO_ERROR);
    char *b = (char *)d.p;
    HIP_TRY(hipMemcpy(b + o_state, state, 32, hipMemcpyHostToDevice), QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipMemcpy(b + o_smp, smp, 80, hipMemcpyHostToDevice), QOA_ENCODE_IO_ERROR);
    HIP_TRY(launch_debug_slice((const int32_t *)(b + o_state), (const int32_t *)(b + o_smp), len, generic,
                               (unsigned long long *)(b + o_tot), (unsigned long long *)(b + o_fst),
                               (unsigned long long *)(b + o_slice), (int32_t *)(b + o_end), nullptr),
            QOA_ENCODE_IO_ERROR);
    HIP_TRY(hipDeviceSynchronize(), QOA_ENCODE_IO_ERROR);
    if (total_rank) HIP_TRY(hipMemcpy(total_rank, b + o_tot, 128, hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
    if (first_rank) HIP_TRY(hipMemcpy(first_rank, b + o_fst, 128, hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
    if (slices) HIP_TRY(hipMemcpy(slices, b + o_slice, 128, hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
    if (end_states) HIP_TRY(hipMemcpy(end_states, b + o_end, 512, hipMemcpyDeviceToHost), QOA_ENCODE_IO_ERROR);
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_encode_all(const int16_t *pcm, size_t n_i16, const qoa_desc *desc, uint8_t *out, size_t cap,
                              size_t *written) try { // src/lib.rs:760-763
    qoa_encoder *e = nullptr;
    int rc = qoa_encoder_new(desc, &e);
    if (rc) return rc;
    rc = qoa_encoder_encode(e, pcm, n_i16, out, cap, written);
    qoa_encoder_free(e);
    return rc;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// stream walking (host bytes only)
// ------------------------------------------------------------------------------------------------
int Walker::walk(const uint8_t *data, size_t len, size_t *pos_io, std::vector<FrameRec> &frames) {
    size_t pos = *pos_io;
    for (;;) {
        // read_u64_be of the frame header: UnexpectedEof -- even on a partial header -- is the clean end
        // (src/lib.rs:206-216); Cursor::read_exact leaves the cursor at EOF.
        if (len - pos < 8) { pos = len; *pos_io = pos; return kTermEnd; }
        uint64_t fh = be64(data + pos);
        size_t hpos = pos;
        pos += 8;
        uint8_t nch = (uint8_t)(fh >> 56);
        uint32_t rate = (uint32_t)((fh >> 32) & 0xffffff);
        uint16_t nspc = (uint16_t)((fh >> 16) & 0xffff);
        uint16_t fsize = (uint16_t)(fh & 0xffff);
        *pos_io = pos;
        if (nch == 0 || rate == 0) return QOA_DECODE_INVALID_FRAME_HEADER;          // :227-229
        size_t non_sample = kHeaderSize + kLmsLen * 4 * (size_t)nch;                // :231-235
        if ((size_t)fsize <= non_sample) return QOA_DECODE_INVALID_FRAME_HEADER;
        size_t num_slices = ((size_t)fsize - non_sample) / 8;
        if (num_slices % nch != 0) return QOA_DECODE_INVALID_FRAME_HEADER;          // :239-241
        if (num_slices / nch > (size_t)kSlicesPerFrame) return QOA_DECODE_INVALID_FRAME_HEADER; // :242-244
        if (mode == kFixed) {                                                       // :246-259
            if (first) { fixed_channels = nch; fixed_rate = rate; }
            else if (nch != fixed_channels || rate != fixed_rate) return QOA_DECODE_INCOMPATIBLE_FRAME;
        }
        first = false;
        // LMS block: 16 bytes per channel, read as two u64 each (src/lib.rs:271-281)
        size_t lms_bytes = 16 * (size_t)nch;
        if (len - pos < lms_bytes) { pos = len; *pos_io = pos; return QOA_DECODE_IO_ERROR; }
        pos += lms_bytes;
        // slice rows are consumed while num_samples_per_channel_remaining > 0 (src/lib.rs:679-682, :320-328):
        // the header's frame_size is validated above but does NOT drive the reader.
        size_t rows_needed = ((size_t)nspc + kSliceLen - 1) / kSliceLen;
        size_t row_bytes = 8 * (size_t)nch;
        size_t rows_avail = std::min(rows_needed, (len - pos) / row_bytes);
        FrameRec f;
        f.pos = hpos;
        f.channels = nch;
        f.rate = rate;
        f.nspc = nspc;
        f.rows = (uint32_t)rows_avail;
        f.n_out = rows_avail == rows_needed ? nspc : (uint32_t)(rows_avail * kSliceLen);
        frames.push_back(f);
        if (rows_avail < rows_needed) { pos = len; *pos_io = pos; return QOA_DECODE_IO_ERROR; }
        pos += rows_needed * row_bytes;
        *pos_io = pos;
    }
}

// The general kernel's launches (short last frames of clips, odd alignments) run BESIDE K2u's on a side stream: behind it on
// the same stream a few hundred short frames kept the chip for a wavefront's lifetime (0.2-0.6 ms) on their own.  One side
// stream and one event pair serve any number of tables that run on the same caller stream (a pipeline's time slices): a
// stream-wait captures the event's latest record at the time of the call.
struct DecSide {
    hipStream_t s = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DecSide() = default;
    DecSide(const DecSide &) = delete;
    DecSide &operator=(const DecSide &) = delete;
    hipError_t ensure() {
        hipError_t e = hipSuccess;
        if (!s) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e == hipSuccess && !ev_fork) e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
        if (e == hipSuccess && !ev_join) e = hipEventCreateWithFlags(&ev_join, hipEventDisableTiming);
        return e;
    }
    ~DecSide() {
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
};

// Device-side decode tables: frames of 1..8 channels go to K2u / K2 (one entry per frame, grouped by channel count);
// frames with more channels (legal in the format, exotic in practice) go to the generic kernel with one entry per
// (frame, channel) chain.  A table is a VIEW: classify() sorts a slice's records into host vectors, flatten() appends them
// to one host image, bind() points the table at that image's place in a device buffer.  A decode plan owns one table and
// its buffers (upload()); the host pipeline keeps the tables of ALL its time slices in one device buffer with one shared
// counter / sink block and one side stream (advisor, round 4: a 3-minute clip used to cost ~600 hipMalloc / hipFree calls).
struct DecTables {
    const DecChain *d_frames[kMaxChannels + 1] = {nullptr};
    unsigned n_frames[kMaxChannels + 1] = {0};
    bool whole_pieces[kMaxChannels + 1] = {false};
    // K2u (qoa_decode.hip): uniform full frames (256 rows, 5120 sample-times) of 1 .. 8 channels whose PCM sits at a multiple
    // of 4 bytes; everything else stays in frames[] for the general kernel
    const DecChain *d_uframes[kMaxChannels + 1] = {nullptr};
    unsigned n_uframes[kMaxChannels + 1] = {0};
    // The uniform table of a channel count that has the absolute-line variant of K2u (qoa_kernels.h) is ordered
    // [ odd | on the 128-byte grid | off it ]: the LAST n_uabs[n] frames have a PCM offset off the grid but a multiple of 8 bytes
    // (the variant's: whole-line streamed stores instead of write-back half lines), the n_ugrid[n] before them an offset on the
    // grid (either kernel takes them), the first ones ("odd": offsets 4 mod 8) are K2u's alone.
    unsigned n_uabs[kMaxChannels + 1] = {0}, n_ugrid[kMaxChannels + 1] = {0};
    const DecChain *d_chains = nullptr;
    unsigned n_chains = 0;
    unsigned *counter_own = nullptr; // [0]: this table's header-mismatch counter
    uint8_t *sink = nullptr;         // kDecodeScratchBytes: where the general kernel's stores beyond a frame's end go
    DecSide *side = nullptr;

    // what a single-table owner (a decode plan) holds; the host pipeline's tables leave these empty
    DevBuf own_records, own_scratch;
    std::unique_ptr<DecSide> own_side;
    std::vector<DecChain> own_host;  // source of the asynchronous copy: lives until the next upload

    DecTables() = default;
    DecTables(const DecTables &) = delete;
    DecTables &operator=(const DecTables &) = delete;

    // host side of a slice's records, sorted by the kernel that takes them
    std::vector<DecChain> per[kMaxChannels + 1], generic, uni[kMaxChannels + 1], uabs[kMaxChannels + 1], ugrid[kMaxChannels + 1];

    size_t classify(const std::vector<DecChain> &recs) {
        for (auto &v : per) v.clear();
        for (auto &v : uni) v.clear();
        for (auto &v : uabs) v.clear();
        for (auto &v : ugrid) v.clear();
        generic.clear();
        for (const DecChain &d : recs) {
            if (d.rows == 0 || d.n_out == 0) continue;
            if (d.channels <= kMaxChannels) per[d.channels].push_back(d);
            else for (unsigned c = 0; c < d.channels; c++) {
                DecChain e = d;
                e.channel = (uint8_t)c;
                e.out_off = d.out_off + c;
                generic.push_back(e);
            }
        }
        size_t total = generic.size();
        for (int n = 1; n <= kMaxChannels; n++) {
            if (decode_uniform_group(n)) {
                std::vector<DecChain> rest;
                for (const DecChain &d : per[n]) {
                    // (K2u's 16-byte stores want a dword-aligned PCM address.  On the 128-byte grid its runs are whole cache
                    // lines, streamed; off it the kernel switches to write-back stores: as streamed stores they ran 1.4-1.6x
                    // SLOWER than the general kernel, 2.07 vs 1.44 ms on 2048 stereo clips x 64 frames)
                    const bool uniform = d.rows == kSlicesPerFrame && d.n_out == (uint32_t)kFrameLen && (d.out_off * 2) % 4 == 0;
                    const bool abs_n = decode_uniform_abs_supported(n);
                    const bool off_grid = abs_n && (d.out_off * 2) % 128 != 0 && (d.out_off * 2) % 8 == 0;
                    const bool on_grid = abs_n && (d.out_off * 2) % 128 == 0;
                    (!uniform ? rest : off_grid ? uabs[n] : on_grid ? ugrid[n] : uni[n]).push_back(d);
                }
                per[n].swap(rest); // (K2u's last wave may be short of frames: nothing is left over for a launch of its own)
            }
            n_uabs[n] = (unsigned)uabs[n].size();
            n_ugrid[n] = (unsigned)ugrid[n].size();
            uni[n].insert(uni[n].end(), ugrid[n].begin(), ugrid[n].end()); // (one table: odd | on the grid | off it)
            uni[n].insert(uni[n].end(), uabs[n].begin(), uabs[n].end());
            uabs[n].clear();
            ugrid[n].clear();
            n_uframes[n] = (unsigned)uni[n].size();
            n_frames[n] = (unsigned)per[n].size();
            whole_pieces[n] = true;
            for (const DecChain &d : per[n]) if (((uint64_t)d.n_out * n * 2) % 16 != 0) whole_pieces[n] = false;
            total += uni[n].size() + per[n].size();
        }
        n_chains = (unsigned)generic.size();
        return total;
    }
    // appends the classified records to `dst` in bind()'s order and drops the per-kernel host vectors
    void flatten(std::vector<DecChain> &dst) {
        for (int n = 1; n <= kMaxChannels; n++) dst.insert(dst.end(), uni[n].begin(), uni[n].end());
        for (int n = 1; n <= kMaxChannels; n++) dst.insert(dst.end(), per[n].begin(), per[n].end());
        dst.insert(dst.end(), generic.begin(), generic.end());
        for (auto &v : per) { v.clear(); v.shrink_to_fit(); }
        for (auto &v : uni) { v.clear(); v.shrink_to_fit(); }
        generic.clear(); generic.shrink_to_fit();
    }
    // the table's records start at `base` (device memory, flatten()'s order); returns the first record behind them
    const DecChain *bind(const DecChain *base) {
        for (int n = 1; n <= kMaxChannels; n++) { d_uframes[n] = base; base += n_uframes[n]; }
        for (int n = 1; n <= kMaxChannels; n++) { d_frames[n] = base; base += n_frames[n]; }
        d_chains = base;
        return base + n_chains;
    }
    void set_scratch(void *block, DecSide *shared_side) { // block: 64 + kDecodeScratchBytes of device memory
        counter_own = (unsigned *)block;
        sink = (uint8_t *)block + 64;
        side = shared_side;
    }

    // single-table form (decode plans): own buffers.  sync = false: the copy is only enqueued on `stream`; the caller must
    // not call upload() again before it ran
    hipError_t upload(const std::vector<DecChain> &recs, hipStream_t stream, bool sync = true) {
        const size_t total = classify(recs);
        own_host.clear();
        own_host.reserve(total);
        flatten(own_host);
        const bool fresh = own_scratch.p == nullptr;
        hipError_t e = own_scratch.reserve(64 + kDecodeScratchBytes); // [0]: mismatch counter, [64..]: store sink
        if (e == hipSuccess && (fresh || sync)) e = hipMemsetAsync(own_scratch.p, 0, sizeof(unsigned), stream);
        if (e == hipSuccess && total) e = own_records.reserve(total * sizeof(DecChain));
        if (e == hipSuccess && total)
            e = hipMemcpyAsync(own_records.p, own_host.data(), total * sizeof(DecChain), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        if (!own_side) own_side.reset(new (std::nothrow) DecSide());
        if (!own_side) return hipErrorOutOfMemory;
        bind((const DecChain *)own_records.p);
        set_scratch(own_scratch.p, own_side.get());
        if (sync) e = hipStreamSynchronize(stream);
        return e;
    }
    bool empty() const {
        if (n_chains) return false;
        for (int n = 1; n <= kMaxChannels; n++) if (n_frames[n] || n_uframes[n]) return false;
        return true;
    }
    // counter: where header mismatches are counted (null: this table's own counter, reset first -- the count is per run;
    // the host pipeline hands all its time slices one counter that adds up over the call)
    hipError_t run(const uint8_t *d_in, int16_t *d_out, hipStream_t stream, unsigned *counter = nullptr) {
        if (!counter) {
            hipError_t e0 = hipMemsetAsync(counter_own, 0, sizeof(unsigned), stream);
            if (e0 != hipSuccess) return e0;
            counter = counter_own;
        }
        bool has_u = false, has_g = n_chains != 0;
        for (int n = 1; n <= kMaxChannels; n++) {
            has_u = has_u || n_uframes[n];
            // (the general kernel's frames -- and a handful of K2u-only frames beside an absolute-line launch -- go to the side stream)
            has_g = has_g || n_frames[n] || (n_uabs[n] && n_uframes[n] - n_uabs[n] - n_ugrid[n] != 0);
        }
        hipStream_t gs = stream; // where the general kernel's launches go
#ifdef QOA_LAB_DEC_SIDE_KNOB // (tools/k2lab builds only: 0 = one stream, 2 = skip the general kernel -- wrong PCM, timing experiments)
        static const int lab_side = getenv("QOA_LAB_DEC_SIDE") ? atoi(getenv("QOA_LAB_DEC_SIDE")) : 1;
#else
        constexpr int lab_side = 1;
#endif
        if (lab_side == 2) has_g = false;
        if (has_u && has_g && lab_side == 1) {
            hipError_t e = side->ensure();
            if (e == hipSuccess) e = hipEventRecord(side->ev_fork, stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(side->s, side->ev_fork, 0);
            if (e != hipSuccess) return e;
            gs = side->s;
        }
        for (int n = 1; n <= kMaxChannels && lab_side != 2; n++) {
            hipError_t e = launch_decode_frames(n, whole_pieces[n], d_in, d_out, d_frames[n], n_frames[n], counter, sink, gs);
            if (e != hipSuccess) return e;
        }
        if (lab_side != 2) {
            hipError_t e = launch_decode(d_in, d_out, d_chains, n_chains, counter, gs);
            if (e != hipSuccess) return e;
        }
        for (int n = 1; n <= kMaxChannels; n++) {
            if (!n_uframes[n]) continue;
            // K2u stores 16-byte pieces at dword-aligned addresses: with a base pointer off the dword grid (a caller's sub-buffer)
            // the same table goes through the general kernel instead
            hipError_t e = hipSuccess;
            if ((uintptr_t)d_out % 4 != 0)
                e = launch_decode_frames(n, true, d_in, d_out, d_uframes[n], n_uframes[n], counter, sink, stream);
            else if ((uintptr_t)d_out % 128 != 0 || n_uabs[n] == 0)
                // A base pointer off the grid (a caller's sub-buffer) shifts every frame, and a table without off-grid frames has
                // nothing for the absolute-line variant: everything through K2u (streamed / write-back stores per wavefront)
                e = launch_decode_uniform(n, d_in, d_out, d_uframes[n], n_uframes[n], counter, stream);
            else {
                // Offsets mean what they say: the off-grid frames go to the absolute-line variant.  The frames ON the grid join
                // them unless they fill the chip on their own (>= 2048 wavefronts: a K2u launch of their own, which is a few
                // per cent faster on them): a small launch beside a large one runs alone on the chip for a wavefront's lifetime
                // (0.2 ms: this very split cost the short-last-frame batches 0.1 ms when it was first built).
                const unsigned fpw = decode_uniform_group(n);
                const bool grid_own = (n_ugrid[n] + fpw - 1) / fpw >= 2048u;
                const unsigned n_k2u = n_uframes[n] - n_uabs[n] - (grid_own ? 0u : n_ugrid[n]);
                e = launch_decode_uniform(n, d_in, d_out, d_uframes[n], n_k2u, counter, n_k2u * 8u < n_uframes[n] ? gs : stream);
                if (e == hipSuccess)
                    e = launch_decode_uniform_abs(n, d_in, d_out, d_uframes[n] + n_k2u, n_uframes[n] - n_k2u, counter, stream);
            }
            if (e != hipSuccess) return e;
        }
        if (gs != stream) { // join: the caller's stream continues when both are done
            hipError_t e = hipEventRecord(side->ev_join, side->s);
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, side->ev_join, 0);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
};

// One record per frame (the `channel` field is unused, out_off is the frame's first interleaved sample).
static void frames_to_records(const std::vector<FrameRec> &frames, size_t first_frame, uint64_t in_base, uint64_t out_base,
                              std::vector<DecChain> &recs, uint64_t *samples_total) {
    uint64_t out = out_base;
    for (size_t i = first_frame; i < frames.size(); i++) {
        const FrameRec &f = frames[i];
        if (f.rows > 0 && f.n_out > 0) {
            DecChain d;
            d.frame_off = in_base + f.pos;
            d.out_off = out;
            d.n_out = f.n_out;
            d.rows = (uint16_t)std::min<uint32_t>(f.rows, (f.n_out + kSliceLen - 1) / kSliceLen);
            d.channels = f.channels;
            d.channel = 0;
            d.expect_hi = d.expect_lo = 0;
            recs.push_back(d);
        }
        out += (uint64_t)f.n_out * f.channels;
    }
    if (samples_total) *samples_total = out - out_base;
}

extern "C" int qoa_stream_scan(const uint8_t *bytes, size_t len, qoa_stream_info *info) try {
    if (!info) return fail_msg("null pointer", QOA_DECODE_IO_ERROR);
    memset(info, 0, sizeof *info);
    info->in_new = 1;
    if (len < 4) { info->terminal = QOA_DECODE_IO_ERROR; return QOA_OK; }
    if (be32(bytes) != kMagic) { info->terminal = QOA_DECODE_NOT_QOA_FILE; return QOA_OK; }
    if (len < 8) { info->terminal = QOA_DECODE_IO_ERROR; return QOA_OK; }
    Walker w;
    uint32_t s = be32(bytes + 4);
    w.mode = s == 0 ? kStreaming : kFixed;
    w.fixed_samples = s;
    size_t pos = 8;
    std::vector<FrameRec> frames;
    int t = w.walk(bytes, len, &pos, frames);
    info->mode.kind = w.mode == kStreaming ? QOA_MODE_STREAMING : QOA_MODE_FIXED_SAMPLES;
    info->mode.channels = w.fixed_channels;
    info->mode.sample_rate = w.fixed_rate;
    info->mode.samples = s;
    if (w.mode == kFixed && frames.empty()) { // the first frame is parsed inside QoaDecoder::new
        info->terminal = t == kTermEnd ? QOA_DECODE_NO_SAMPLES : t;
        return QOA_OK;
    }
    info->in_new = 0;
    info->terminal = t;
    info->n_frames = frames.size();
    for (auto &f : frames) info->n_samples += (uint64_t)f.n_out * f.channels;
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// batch decode plan
// ------------------------------------------------------------------------------------------------
struct qoa_decode_plan {
    DecTables t;
    uint64_t in_bytes = 0, pcm_bytes = 0, samples = 0;
};

static int plan_upload(qoa_decode_plan *plan, const std::vector<DecChain> &recs) {
    HIP_TRY(plan->t.upload(recs, nullptr), QOA_DECODE_IO_ERROR);
    return QOA_OK;
}

extern "C" int qoa_decode_plan_from_streams(const uint8_t *const *streams, const size_t *len, const uint64_t *in_offset,
                                            const uint64_t *out_offset, size_t n_streams, int *status,
                                            uint64_t *samples_out, qoa_decode_plan **out) try {
    if (!out) return fail_msg("null out pointer", QOA_DECODE_IO_ERROR);
    *out = nullptr;
    std::vector<DecChain> chains;
    uint64_t in_bytes = 0, samples = 0;
    for (size_t i = 0; i < n_streams; i++) {
        int st = QOA_OK;
        uint64_t n_i16 = 0;
        const uint8_t *d = streams[i];
        size_t n = len[i];
        if (in_offset[i] % 8 != 0) return fail_msg("in_offset must be 8-byte aligned", QOA_DECODE_IO_ERROR);
        if (n < 4) st = QOA_DECODE_IO_ERROR;
        else if (be32(d) != kMagic) st = QOA_DECODE_NOT_QOA_FILE;
        else if (n < 8) st = QOA_DECODE_IO_ERROR;
        else {
            Walker w;
            uint32_t s = be32(d + 4);
            w.mode = s == 0 ? kStreaming : kFixed;
            w.fixed_samples = s;
            size_t pos = 8;
            std::vector<FrameRec> frames;
            st = w.walk(d, n, &pos, frames);
            if (w.mode == kFixed && frames.empty() && st == kTermEnd) st = QOA_DECODE_NO_SAMPLES;
            frames_to_records(frames, 0, in_offset[i], out_offset[i], chains, &n_i16);
        }
        if (status) status[i] = st;
        if (samples_out) samples_out[i] = n_i16;
        in_bytes += n;
        samples += n_i16;
    }
    std::unique_ptr<qoa_decode_plan> plan(new (std::nothrow) qoa_decode_plan()); // (released on every way out, a throw included)
    if (!plan) return fail_msg("out of memory", QOA_DECODE_IO_ERROR);
    plan->in_bytes = in_bytes;
    plan->samples = samples;
    plan->pcm_bytes = samples * 2;
    int rc = plan_upload(plan.get(), chains);
    if (rc) return rc;
    *out = plan.release();
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decode_plan_from_clips(const qoa_clip *clips, size_t n_clips, qoa_decode_plan **out) try {
    if (!out) return fail_msg("null out pointer", QOA_DECODE_IO_ERROR);
    *out = nullptr;
    std::vector<DecChain> chains;
    uint64_t in_bytes = 0, samples = 0;
    for (size_t i = 0; i < n_clips; i++) {
        const qoa_desc &d = clips[i].desc;
        if (validate_desc(&d)) return fail_msg("invalid qoa_desc in decode plan", QOA_DECODE_INVALID_FRAME_HEADER);
        if (clips[i].out_offset % 8 != 0) return fail_msg("encoded offset must be 8-byte aligned", QOA_DECODE_IO_ERROR);
        uint64_t pos = clips[i].out_offset + 8, outp = clips[i].pcm_offset;
        uint32_t left = d.samples;
        while (left > 0) {
            uint32_t fl = std::min<uint32_t>(left, kFrameLen);
            uint32_t rows = (fl + kSliceLen - 1) / kSliceLen;
            uint32_t fsize = qoa_frame_size(d.channels, rows);
            {
                DecChain dc;
                dc.frame_off = pos;
                dc.out_off = outp;
                dc.n_out = fl;
                dc.rows = (uint16_t)rows;
                dc.channels = d.channels;
                dc.channel = 0;
                dc.expect_hi = ((uint32_t)d.channels << 24) | d.sample_rate;
                dc.expect_lo = (fl << 16) | fsize;
                chains.push_back(dc);
            }
            pos += fsize;
            outp += (uint64_t)fl * d.channels;
            left -= fl;
        }
        in_bytes += qoa_encoded_size(d.channels, d.samples);
        samples += (uint64_t)d.samples * d.channels;
    }
    std::unique_ptr<qoa_decode_plan> plan(new (std::nothrow) qoa_decode_plan()); // (released on every way out, a throw included)
    if (!plan) return fail_msg("out of memory", QOA_DECODE_IO_ERROR);
    plan->in_bytes = in_bytes;
    plan->samples = samples;
    plan->pcm_bytes = samples * 2;
    int rc = plan_upload(plan.get(), chains);
    if (rc) return rc;
    *out = plan.release();
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" void qoa_decode_plan_free(qoa_decode_plan *plan) { delete plan; }

extern "C" int qoa_decode_plan_bytes(const qoa_decode_plan *plan, uint64_t *in_bytes, uint64_t *pcm_bytes,
                                     uint64_t *samples_i16) try {
    if (!plan) return fail_msg("null plan", QOA_DECODE_IO_ERROR);
    if (in_bytes) *in_bytes = plan->in_bytes;
    if (pcm_bytes) *pcm_bytes = plan->pcm_bytes;
    if (samples_i16) *samples_i16 = plan->samples;
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decode_plan_run(qoa_decode_plan *plan, const uint8_t *d_in, int16_t *d_out, void *stream) try {
    if (!plan) return fail_msg("null plan", QOA_DECODE_IO_ERROR);
    HIP_TRY(plan->t.run(d_in, d_out, (hipStream_t)stream), QOA_DECODE_IO_ERROR);
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decode_plan_mismatches(qoa_decode_plan *plan, uint32_t *count) try {
    if (!plan || !count) return fail_msg("null pointer", QOA_DECODE_IO_ERROR);
    // the runs may sit on non-blocking side streams, which the null stream's copy below does not wait for
    HIP_TRY(hipDeviceSynchronize(), QOA_DECODE_IO_ERROR);
    HIP_TRY(hipMemcpy(count, plan->t.counter_own, sizeof(uint32_t), hipMemcpyDeviceToHost), QOA_DECODE_IO_ERROR);
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// Host-memory batch decode: frame-sliced pipeline (the decode half)
//
// Decode frames are independent (the LMS state is re-seeded from every frame header, src/lib.rs:270-281), so the batch
// is cut along time exactly like the encode pipeline but without any carried state: frames [j*F, (j+1)*F) of all clips
// are copied in (their bytes are contiguous per clip), decoded by K2 / K2u, and their PCM copied out, the three stages of
// neighbouring slices overlapping on three streams.  The copy out moves 2 bytes per sample over PCIe: this pipeline is
// bound by that transfer by construction (LABLOG.md section 5 and bench.py's e2e record have the measured rates), not by the kernel.
// Layout trusted like qoa_decode_plan_from_clips; every frame header is verified on the device.
// ------------------------------------------------------------------------------------------------
struct DecTablesHolder {
    std::vector<std::unique_ptr<DecTables>> t; // one VIEW per time slice of the batch dec_key describes, all of them into ...
    DevBuf records;                             // ... this one buffer: every slice's frame records, uploaded once
    DevBuf scratch;                             // [0]: header mismatches of the whole call, [64..): the general kernel's store sink
    DecSide side;                               // the one side stream the slices' general launches share
};

qoa_host_pipeline::~qoa_host_pipeline() {
    if (s_in) (void)hipStreamSynchronize(s_in);
    if (s_k) (void)hipStreamSynchronize(s_k);
    if (s_out) (void)hipStreamSynchronize(s_out);
    delete dec;
    for (int i = 0; i < 2; i++) {
        if (in_done[i]) (void)hipEventDestroy(in_done[i]);
        if (k_done[i]) (void)hipEventDestroy(k_done[i]);
        if (out_done[i]) (void)hipEventDestroy(out_done[i]);
    }
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_k) (void)hipStreamDestroy(s_k);
    if (s_out) (void)hipStreamDestroy(s_out);
}

static int host_pipe_decode(HostPipe &P, const qoa_clip *clips, size_t n_clips, const uint8_t *h_enc, int16_t *h_pcm,
                            uint32_t chunk_frames) {
    if (!clips || !h_enc || !h_pcm) return fail_msg("null pointer", QOA_DECODE_IO_ERROR);
    if (n_clips == 0) return QOA_OK;
    uint32_t max_frames = 0;
    for (size_t i = 0; i < n_clips; i++) {
        if (validate_desc(&clips[i].desc)) return fail_msg("invalid qoa_desc in decode batch", QOA_DECODE_INVALID_FRAME_HEADER);
        max_frames = std::max<uint32_t>(max_frames, (clips[i].desc.samples + kFrameLen - 1) / kFrameLen);
    }
    // (as in host_pipe_encode: memory of this device is used in place)
    bool src_direct = on_current_device(h_enc) && !P.shared_side_is_remote;
    const bool dst_direct = on_current_device(h_pcm);
    const bool device_memory = dst_direct || on_current_device(h_enc);
    if (src_direct) { // the kernels would read frames in place with 8-byte loads: odd offsets go through the staging buffers
        if ((uintptr_t)h_enc % 8 != 0) src_direct = false;
        for (size_t i = 0; i < n_clips && src_direct; i++)
            if (clips[i].out_offset % 8 != 0) src_direct = false;
    }
    uint32_t F = chunk_frames ? chunk_frames : 8;
    if (src_direct && dst_direct) F = max_frames;
    const uint32_t n_chunks = (max_frames + F - 1) / F;
    HIP_TRY(P.init(), QOA_DECODE_IO_ERROR);
    if (device_memory) HIP_TRY(order_after_device_work(), QOA_DECODE_IO_ERROR);
    PipeDrain drain{P};
    if (!P.dec) P.dec = new (std::nothrow) DecTablesHolder();
    if (!P.dec) return fail_msg("out of memory", QOA_DECODE_IO_ERROR);

    // device staging per buffer: per clip F frames of encoded bytes and F frames of PCM
    std::vector<uint64_t> pcm_reg(n_clips + 1, 0), enc_reg(n_clips + 1, 0);
    bool uniform = n_clips > 1;
    for (size_t i = 0; i < n_clips; i++) {
        const qoa_desc &d = clips[i].desc;
        pcm_reg[i + 1] = pcm_reg[i] + (uint64_t)F * kFrameLen * d.channels;                          // int16 elements
        enc_reg[i + 1] = enc_reg[i] + (uint64_t)F * qoa_frame_size(d.channels, kSlicesPerFrame);     // bytes, 8-aligned
        if (i > 0) {
            const qoa_desc &q = clips[i - 1].desc;
            uniform = uniform && d.channels == q.channels && d.samples == q.samples &&
                      clips[i].pcm_offset - clips[i - 1].pcm_offset == clips[1].pcm_offset - clips[0].pcm_offset &&
                      clips[i].out_offset - clips[i - 1].out_offset == clips[1].out_offset - clips[0].out_offset;
        }
    }
    uniform = uniform && clips[1].pcm_offset > clips[0].pcm_offset && clips[1].out_offset > clips[0].out_offset &&
              clips[1].pcm_offset - clips[0].pcm_offset >= (uint64_t)clips[0].desc.samples * clips[0].desc.channels &&
              clips[1].out_offset - clips[0].out_offset >= qoa_encoded_size(clips[0].desc.channels, clips[0].desc.samples);
    for (int b = 0; b < 2; b++) {
        if (!dst_direct) HIP_TRY(P.pcm[b].reserve(pcm_reg[n_clips] * 2), QOA_DECODE_IO_ERROR);
        if (!src_direct) HIP_TRY(P.enc[b].reserve(enc_reg[n_clips]), QOA_DECODE_IO_ERROR);
    }
    // what time slice j holds of clip d: frames, encoded bytes (whole frames, the last one possibly short), sample-times
    auto slice_of = [&](const qoa_desc &d, uint32_t j, uint64_t *enc_off, uint64_t *enc_bytes, uint64_t *first, uint64_t *now) {
        *first = (uint64_t)j * F * kFrameLen;
        *now = d.samples > *first ? std::min<uint64_t>(d.samples - *first, (uint64_t)F * kFrameLen) : 0;
        const uint64_t fb = qoa_frame_size(d.channels, kSlicesPerFrame);
        const uint64_t full = *now / kFrameLen, rest = *now % kFrameLen;
        *enc_off = 8 + (uint64_t)j * F * fb;
        *enc_bytes = full * fb + (rest ? qoa_frame_size(d.channels, (uint32_t)((rest + kSliceLen - 1) / kSliceLen)) : 0);
    };

    // ---- frame tables of all time slices: built and uploaded once per (clip list, slice length, in-place sides) ----
    if (!P.dec_key.matches(clips, n_clips, F, src_direct, dst_direct)) {
        P.dec_key.valid = false;
        HIP_TRY(hipStreamSynchronize(P.s_k), QOA_DECODE_IO_ERROR); // (no launch of an earlier call may still read the old tables)
        HIP_TRY(P.dec->scratch.reserve(64 + kDecodeScratchBytes), QOA_DECODE_IO_ERROR);
        P.dec->t.clear();
        std::vector<DecChain> recs, all; // all: the records of every time slice, in the order the views are bound below
        for (uint32_t j = 0; j < n_chunks; j++) {
            recs.clear();
            for (size_t i = 0; i < n_clips; i++) {
                const qoa_desc &d = clips[i].desc;
                uint64_t eo, eb, first, now;
                slice_of(d, j, &eo, &eb, &first, &now);
                uint64_t pos = src_direct ? clips[i].out_offset + eo : enc_reg[i];
                uint64_t outp = dst_direct ? clips[i].pcm_offset + first * d.channels : pcm_reg[i], left = now;
                while (left > 0) {
                    const uint32_t fl = (uint32_t)std::min<uint64_t>(left, kFrameLen);
                    const uint32_t rows = (fl + kSliceLen - 1) / kSliceLen;
                    const uint32_t fsize = qoa_frame_size(d.channels, rows);
                    DecChain dc;
                    dc.frame_off = pos; dc.out_off = outp; dc.n_out = fl; dc.rows = (uint16_t)rows;
                    dc.channels = d.channels; dc.channel = 0;
                    dc.expect_hi = ((uint32_t)d.channels << 24) | d.sample_rate;
                    dc.expect_lo = (fl << 16) | fsize;
                    recs.push_back(dc);
                    pos += fsize; outp += (uint64_t)fl * d.channels; left -= fl;
                }
            }
            std::unique_ptr<DecTables> T(new (std::nothrow) DecTables());
            if (!T) return fail_msg("out of memory", QOA_DECODE_IO_ERROR);
            T->classify(recs);
            T->flatten(all);
            P.dec->t.push_back(std::move(T));
        }
        // ONE allocation and ONE upload for all slices (synchronous: `all` may go); every table is a view into it
        HIP_TRY(P.dec->records.reserve(std::max<size_t>(all.size(), 1) * sizeof(DecChain)), QOA_DECODE_IO_ERROR);
        if (!all.empty())
            HIP_TRY(hipMemcpy(P.dec->records.p, all.data(), all.size() * sizeof(DecChain), hipMemcpyHostToDevice), QOA_DECODE_IO_ERROR);
        const DecChain *next = (const DecChain *)P.dec->records.p;
        for (auto &T : P.dec->t) {
            next = T->bind(next);
            T->set_scratch(P.dec->scratch.p, &P.dec->side);
        }
        P.dec_key.set(clips, n_clips, F, src_direct, dst_direct);
    }
    unsigned *const bad_count = (unsigned *)P.dec->scratch.p;
    HIP_TRY(hipMemsetAsync(bad_count, 0, sizeof(unsigned), P.s_k), QOA_DECODE_IO_ERROR);

    for (uint32_t j = 0; j < n_chunks + 1; j++) {
        const int b = (int)(j & 1u);
        if (j < n_chunks) {
            DecTables &T = *P.dec->t[j];
            // ---- stage in: encoded frames of time slice j (buffer b is free once the kernel of slice j-2 has run) ----
            if (!src_direct) {
                if (j >= 2) HIP_TRY(hipStreamWaitEvent(P.s_in, P.k_done[b], 0), QOA_DECODE_IO_ERROR);
                if (uniform) {
                    uint64_t eo, eb, first, now;
                    slice_of(clips[0].desc, j, &eo, &eb, &first, &now);
                    HIP_TRY(hipMemcpy2DAsync(P.enc[b].p, enc_reg[1] - enc_reg[0], h_enc + clips[0].out_offset + eo,
                                             clips[1].out_offset - clips[0].out_offset, eb, n_clips, hipMemcpyDefault, P.s_in),
                            QOA_DECODE_IO_ERROR);
                } else {
                    for (size_t i = 0; i < n_clips; i++) {
                        uint64_t eo, eb, first, now;
                        slice_of(clips[i].desc, j, &eo, &eb, &first, &now);
                        if (now)
                            HIP_TRY(hipMemcpyAsync((uint8_t *)P.enc[b].p + enc_reg[i], h_enc + clips[i].out_offset + eo, eb,
                                                   hipMemcpyDefault, P.s_in), QOA_DECODE_IO_ERROR);
                    }
                }
                HIP_TRY(hipEventRecord(P.in_done[b], P.s_in), QOA_DECODE_IO_ERROR);
                HIP_TRY(hipStreamWaitEvent(P.s_k, P.in_done[b], 0), QOA_DECODE_IO_ERROR);
            }
            // ---- decode time slice j (its PCM buffer is free once slice j-2 has been copied out) ----
            if (j >= 2 && !dst_direct) HIP_TRY(hipStreamWaitEvent(P.s_k, P.out_done[b], 0), QOA_DECODE_IO_ERROR);
            if (!T.empty())
                HIP_TRY(T.run(src_direct ? h_enc : (const uint8_t *)P.enc[b].p, dst_direct ? h_pcm : (int16_t *)P.pcm[b].p, P.s_k, bad_count),
                        QOA_DECODE_IO_ERROR);
            HIP_TRY(hipEventRecord(P.k_done[b], P.s_k), QOA_DECODE_IO_ERROR);
        }
        if (j >= 1 && !dst_direct) {
            // ---- stage out: PCM of time slice j-1 ----
            const uint32_t jj = j - 1;
            const int bo = (int)(jj & 1u);
            HIP_TRY(hipStreamWaitEvent(P.s_out, P.k_done[bo], 0), QOA_DECODE_IO_ERROR);
            if (uniform) {
                uint64_t eo, eb, first, now;
                slice_of(clips[0].desc, jj, &eo, &eb, &first, &now);
                const qoa_desc &d = clips[0].desc;
                HIP_TRY(hipMemcpy2DAsync(h_pcm + clips[0].pcm_offset + first * d.channels, (clips[1].pcm_offset - clips[0].pcm_offset) * 2,
                                         P.pcm[bo].p, (pcm_reg[1] - pcm_reg[0]) * 2, now * d.channels * 2, n_clips,
                                         hipMemcpyDefault, P.s_out), QOA_DECODE_IO_ERROR);
            } else {
                for (size_t i = 0; i < n_clips; i++) {
                    const qoa_desc &d = clips[i].desc;
                    uint64_t eo, eb, first, now;
                    slice_of(d, jj, &eo, &eb, &first, &now);
                    if (now)
                        HIP_TRY(hipMemcpyAsync(h_pcm + clips[i].pcm_offset + first * d.channels, (int16_t *)P.pcm[bo].p + pcm_reg[i],
                                               now * d.channels * 2, hipMemcpyDefault, P.s_out), QOA_DECODE_IO_ERROR);
                }
            }
            HIP_TRY(hipEventRecord(P.out_done[bo], P.s_out), QOA_DECODE_IO_ERROR);
        }
    }
    HIP_TRY(hipStreamSynchronize(P.s_out), QOA_DECODE_IO_ERROR);
    HIP_TRY(hipStreamSynchronize(P.s_k), QOA_DECODE_IO_ERROR);
    uint32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, bad_count, sizeof bad, hipMemcpyDeviceToHost), QOA_DECODE_IO_ERROR);
    if (bad) return fail_msg("frame headers in the batch do not match the clip descriptions", QOA_DECODE_INVALID_FRAME_HEADER);
    return QOA_OK;
}

extern "C" int qoa_host_pipeline_decode(qoa_host_pipeline *p, const qoa_clip *clips, size_t n_clips, const uint8_t *h_enc,
                                        int16_t *h_pcm, uint32_t chunk_frames) try {
    if (!p) return fail_msg("null pipeline", QOA_DECODE_IO_ERROR);
    return host_pipe_decode(*p, clips, n_clips, h_enc, h_pcm, chunk_frames);
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decode_batch_host(const qoa_clip *clips, size_t n_clips, const uint8_t *h_enc, int16_t *h_pcm,
                                     uint32_t chunk_frames) try {
    qoa_host_pipeline P;
    return host_pipe_decode(P, clips, n_clips, h_enc, h_pcm, chunk_frames);
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// Multi-GPU batches behind the C ABI (BASELINE config 4): ONE process, one host thread + one time-sliced
// pipeline per device.
//
// Clips are independent, so a batch shards with no data-path collective: device i takes the i-th block of the block
// partition (qoaudio_amd/dist.py: shard_range -- the first n_clips % n_devices shards hold one clip more).  The one exchange
// the configuration names is the gather of the encoded streams on the root device, devices[0].  In this one-process form it
// is not a separate step: every device runs the host pipeline above on its shard with `out` pointing INTO THE ROOT's
// buffer, so the stage-out copy of time slice j -- a peer copy over the device's own xGMI link, driven by its copy
// engines -- overlaps the kernel of slice j + 1, and the gather is complete one slice after the last kernel.
// Transport: HIP peer copies (hipMemcpy*Async with hipMemcpyDefault between unified addresses; peer access is enabled when
// the devices offer it, otherwise the runtime stages through host memory).  RCCL's grouped ncclSend / ncclRecv is what the
// one-process-per-GPU form uses (qoaudio_amd/dist.py under torch.distributed), where no rank can address a peer's memory;
// a process that owns all devices needs no communicator and no copy kernels on the CUs K1 and K2 are using.
// ------------------------------------------------------------------------------------------------
struct qoa_multi {
    std::vector<int> devices;
    std::vector<qoa_host_pipeline *> pipes; // pipes[i] lives on devices[i]
    ~qoa_multi() {
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        for (size_t i = 0; i < pipes.size(); i++) {
            if (!pipes[i]) continue;
            (void)hipSetDevice(devices[i]);
            delete pipes[i];
        }
        if (have) (void)hipSetDevice(cur);
    }
};

extern "C" int qoa_multi_create_flags(const int *devices, int n_devices, uint32_t flags, qoa_multi **out) try {
    if (!out) return fail_msg("null out pointer", QOA_ENCODE_IO_ERROR);
    *out = nullptr;
    if (!devices || n_devices < 1) return fail_msg("qoa_multi_create: empty device list", QOA_ENCODE_IO_ERROR);
    const bool allow_dup = (flags & QOA_MULTI_ALLOW_DUPLICATE_DEVICES) != 0;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count), QOA_ENCODE_IO_ERROR);
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= count) {
            g_detail = "qoa_multi_create: device " + std::to_string(devices[i]) + " does not exist (" + std::to_string(count) +
                       " visible)";
            return QOA_ENCODE_IO_ERROR;
        }
        for (int k = 0; k < i && !allow_dup; k++)
            if (devices[k] == devices[i]) return fail_msg("qoa_multi_create: a device is listed twice", QOA_ENCODE_IO_ERROR);
    }
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur), QOA_ENCODE_IO_ERROR);
    qoa_multi *m = nullptr;
    try {
        m = new qoa_multi();
        m->devices.assign(devices, devices + n_devices);
        m->pipes.assign((size_t)n_devices, nullptr);
    } catch (const std::exception &) {
        delete m;
        return fail_msg("out of memory", QOA_ENCODE_IO_ERROR);
    }
    int rc = QOA_OK;
    for (int i = 0; i < n_devices && rc == QOA_OK; i++) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) {
            m->pipes[(size_t)i] = new (std::nothrow) qoa_host_pipeline();
            e = m->pipes[(size_t)i] ? m->pipes[(size_t)i]->init() : hipErrorOutOfMemory;
            // debug mode: every shard but the root's treats the shared buffer as a peer's memory (staging + copies)
            if (m->pipes[(size_t)i] && allow_dup && i > 0) m->pipes[(size_t)i]->shared_side_is_remote = true;
        }
        if (e == hipSuccess && i > 0 && devices[i] != devices[0]) { // direct stores into the root's memory where the link allows it
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
                else (void)hipGetLastError(); // "already enabled" is sticky otherwise
            }
        }
        if (e != hipSuccess) rc = fail_hip(e, "qoa_multi_create", QOA_ENCODE_IO_ERROR);
    }
    (void)hipSetDevice(cur);
    if (rc != QOA_OK) { delete m; return rc; }
    *out = m;
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_multi_create(const int *devices, int n_devices, qoa_multi **out) try {
    return qoa_multi_create_flags(devices, n_devices, 0u, out);
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" void qoa_multi_free(qoa_multi *m) { delete m; }

extern "C" int qoa_multi_shard(size_t n_clips, int n_devices, int index, size_t *first, size_t *count) try {
    if (n_devices < 1 || index < 0 || index >= n_devices || !first || !count) return fail_msg("qoa_multi_shard: bad arguments", QOA_ENCODE_IO_ERROR);
    const size_t base = n_clips / (size_t)n_devices, extra = n_clips % (size_t)n_devices, i = (size_t)index;
    *first = i * base + std::min(i, extra);
    *count = base + (i < extra ? 1 : 0);
    return QOA_OK;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

// runs fn(i) for every device that has work, each call bound to its device: on one host thread per device -- a single
// working device runs on the caller's thread -- first failure wins.  Nothing may leave through the C ABI as an exception:
// a thread that cannot be started (std::system_error) ends the call with the IoError code once the threads already
// running have been joined.
template <class Fn>
static int multi_run(qoa_multi *m, size_t n_clips, int io_error, Fn fn) {
    const int n = (int)m->devices.size();
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur), io_error);
    struct Work { int i; size_t first, count; };
    try {
        std::vector<Work> work;
        for (int i = 0; i < n; i++) {
            size_t first = 0, count = 0;
            (void)qoa_multi_shard(n_clips, n, i, &first, &count);
            if (count) work.push_back({i, first, count});
        }
        // the shards' device memory is used in place by kernels on the pipelines' own streams: order the call after
        // everything submitted to these devices so far (see order_after_device_work)
        for (int i = 0; i < n; i++) {
            bool seen = false;
            for (int k = 0; k < i; k++) seen = seen || m->devices[(size_t)k] == m->devices[(size_t)i];
            if (seen) continue;
            hipError_t e = hipSetDevice(m->devices[(size_t)i]);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) { (void)hipSetDevice(cur); return fail_hip(e, "qoa_multi: device synchronize", io_error); }
        }
        std::vector<int> rcs(work.size(), QOA_OK);
        std::vector<std::string> details(work.size());
        auto body = [&](size_t k) {
            const Work &w = work[k];
            const hipError_t e = hipSetDevice(m->devices[(size_t)w.i]);
            rcs[k] = e == hipSuccess ? fn(w.i, w.first, w.count) : fail_hip(e, "hipSetDevice", io_error);
            if (rcs[k] != QOA_OK) details[k] = "device " + std::to_string(m->devices[(size_t)w.i]) + ": " + g_detail;
        };
        if (work.size() == 1) {
            body(0);
        } else {
            std::vector<std::thread> threads;
            bool spawn_failed = false;
            for (size_t k = 0; k < work.size() && !spawn_failed; k++) {
                try {
                    threads.emplace_back(body, k);
                } catch (const std::exception &) {
                    spawn_failed = true;
                }
            }
            for (auto &t : threads) t.join();
            if (spawn_failed) { (void)hipSetDevice(cur); return fail_msg("qoa_multi: could not start a host thread", io_error); }
        }
        (void)hipSetDevice(cur);
        for (size_t k = 0; k < work.size(); k++)
            if (rcs[k] != QOA_OK) { g_detail = details[k]; return rcs[k]; }
        return QOA_OK;
    } catch (const std::exception &) {
        (void)hipSetDevice(cur);
        return fail_msg("qoa_multi: out of memory", io_error);
    }
}

extern "C" int qoa_multi_encode(qoa_multi *m, const qoa_clip *clips, size_t n_clips, const int16_t *const *pcm, uint8_t *out,
                                uint32_t chunk_frames) try {
    if (!m || !clips || !pcm || !out) return fail_msg("qoa_multi_encode: null pointer", QOA_ENCODE_IO_ERROR);
    for (size_t i = 0; i < n_clips; i++) { // the whole batch is validated before any device starts (src/lib.rs:336-344)
        const int rc = validate_desc(&clips[i].desc);
        if (rc) return rc;
    }
    return multi_run(m, n_clips, QOA_ENCODE_IO_ERROR, [&](int i, size_t first, size_t count) {
        if (!pcm[i]) return fail_msg("qoa_multi_encode: null PCM base of a shard", QOA_ENCODE_IO_ERROR);
        return host_pipe_encode(*m->pipes[(size_t)i], clips + first, count, pcm[i], out, chunk_frames);
    });
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_multi_decode(qoa_multi *m, const qoa_clip *clips, size_t n_clips, const uint8_t *enc, int16_t *const *pcm,
                                uint32_t chunk_frames) try {
    if (!m || !clips || !pcm || !enc) return fail_msg("qoa_multi_decode: null pointer", QOA_DECODE_IO_ERROR);
    return multi_run(m, n_clips, QOA_DECODE_IO_ERROR, [&](int i, size_t first, size_t count) {
        if (!pcm[i]) return fail_msg("qoa_multi_decode: null PCM base of a shard", QOA_DECODE_IO_ERROR);
        return host_pipe_decode(*m->pipes[(size_t)i], clips + first, count, enc, pcm[i], chunk_frames);
    });
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

// one-shot forms: the handle lives for one call
extern "C" int qoa_encode_batch_multi(const int *devices, int n_devices, const qoa_clip *clips, size_t n_clips,
                                      const int16_t *const *pcm, uint8_t *out, uint32_t chunk_frames) try {
    qoa_multi *m = nullptr;
    int rc = qoa_multi_create(devices, n_devices, &m);
    if (rc != QOA_OK) return rc;
    rc = qoa_multi_encode(m, clips, n_clips, pcm, out, chunk_frames);
    const std::string keep = g_detail;
    qoa_multi_free(m);
    g_detail = keep;
    return rc;
} catch (...) { return on_exception(QOA_ENCODE_IO_ERROR); }

extern "C" int qoa_decode_batch_multi(const int *devices, int n_devices, const qoa_clip *clips, size_t n_clips, const uint8_t *enc,
                                      int16_t *const *pcm, uint32_t chunk_frames) try {
    qoa_multi *m = nullptr;
    int rc = qoa_multi_create(devices, n_devices, &m);
    if (rc != QOA_OK) return rc == QOA_ENCODE_IO_ERROR ? QOA_DECODE_IO_ERROR : rc;
    rc = qoa_multi_decode(m, clips, n_clips, enc, pcm, chunk_frames);
    const std::string keep = g_detail;
    qoa_multi_free(m);
    g_detail = keep;
    return rc;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

// ------------------------------------------------------------------------------------------------
// QoaDecoder handle
// ------------------------------------------------------------------------------------------------
struct qoa_decoder {
    std::vector<uint8_t> data; // the reader: Cursor<Vec<u8>>
    size_t pos = 0;
    Walker w;
    std::vector<FrameRec> frames;
    std::vector<size_t> frame_pcm; // start of each frame's samples in `pcm`
    std::vector<int16_t> pcm;
    int terminal = kTermEnd;       // what ends the iteration over the bytes seen so far
    // iterator state (src/lib.rs:59-68, :668-692)
    bool returned_first = false;
    size_t it_frame = 0;           // next frame whose header/samples are to be emitted
    bool it_header_done = false;   // header of frames[it_frame] already emitted
    size_t it_sample = 0;          // next sample inside frames[it_frame]
    bool finished_err = false;
    qoa_frame_header current{};    // CurrentFrame.header, :283-286
    hipStream_t stream = nullptr;
    DevBuf d_in, d_out;
    DecTables tables;
    // lazy reader (qoa_decoder_new_reader): bytes are pulled one frame at a time as the iterator runs dry
    qoa_read_fn read = nullptr;
    void *read_ctx = nullptr;
    bool read_eof = false, read_failed = false;
    ~qoa_decoder() { if (stream) (void)hipStreamDestroy(stream); }
};

// read_exact over the callback: appends up to `want` bytes to dec->data, retrying short reads; returns how many arrived.
// 0 bytes from the callback = end of stream, a negative value = an I/O error of the caller's reader.
static size_t reader_pull(qoa_decoder *dec, size_t want) {
    const size_t base = dec->data.size();
    dec->data.resize(base + want);
    size_t got = 0;
    while (got < want && !dec->read_eof && !dec->read_failed) {
        const int64_t n = dec->read(dec->read_ctx, dec->data.data() + base + got, want - got);
        if (n < 0) dec->read_failed = true;
        else if (n == 0) dec->read_eof = true;
        else got += (size_t)std::min<uint64_t>((uint64_t)n, want - got);
    }
    dec->data.resize(base + got);
    return got;
}

// Pull the next frame from the reader: its 8-byte header first, then -- when the header passes the checks the reference
// makes before it reads on (src/lib.rs:227-244) -- the LMS block and the slice rows the header announces (:271-281, :679-682).
// The reference issues these reads slice by slice as the iterator advances; here they happen together at the frame's
// start, so the reader is never more than one frame ahead of the items handed out.
static void reader_pull_frame(qoa_decoder *dec) {
    const size_t at = dec->data.size();
    if (reader_pull(dec, 8) < 8) return;
    const uint64_t fh = be64(dec->data.data() + at);
    const size_t nch = (size_t)(fh >> 56), rate = (size_t)((fh >> 32) & 0xffffff), nspc = (size_t)((fh >> 16) & 0xffff),
                 fsize = (size_t)(fh & 0xffff);
    const size_t non_sample = kHeaderSize + kLmsLen * 4 * nch;
    if (nch == 0 || rate == 0 || fsize <= non_sample || ((fsize - non_sample) / 8) % nch != 0 ||
        (fsize - non_sample) / 8 / nch > (size_t)kSlicesPerFrame)
        return; // Walker::walk reports InvalidFrameHeader from the header alone, as the reference does
    (void)reader_pull(dec, 16 * nch + (nspc + kSliceLen - 1) / kSliceLen * 8 * nch);
}

// Decode frames [first_frame, end) on the GPU and append their PCM to dec->pcm.
static int decoder_decode_new_frames(qoa_decoder *dec, size_t first_frame) {
    std::vector<DecChain> recs;
    uint64_t n_i16 = 0;
    frames_to_records(dec->frames, first_frame, 0, 0, recs, &n_i16);
    size_t base = dec->pcm.size();
    {
        size_t off = base;
        for (size_t i = first_frame; i < dec->frames.size(); i++) {
            dec->frame_pcm.push_back(off);
            off += (size_t)dec->frames[i].n_out * dec->frames[i].channels;
        }
    }
    dec->pcm.resize(base + n_i16);
    if (recs.empty() || n_i16 == 0) return QOA_OK;
    if (!dec->stream) HIP_TRY(hipStreamCreateWithFlags(&dec->stream, hipStreamNonBlocking), QOA_DECODE_IO_ERROR);
    // upload only the byte range the new frames span (frame positions are multiples of 8 by construction)
    size_t lo = dec->frames[first_frame].pos & ~(size_t)7, hi = dec->data.size();
    for (auto &c : recs) c.frame_off -= lo;
    HIP_TRY(dec->d_in.reserve(hi - lo + 16), QOA_DECODE_IO_ERROR);
    HIP_TRY(dec->d_out.reserve(n_i16 * 2), QOA_DECODE_IO_ERROR);
    HIP_TRY(hipMemcpyAsync(dec->d_in.p, dec->data.data() + lo, hi - lo, hipMemcpyHostToDevice, dec->stream),
            QOA_DECODE_IO_ERROR);
    HIP_TRY(dec->tables.upload(recs, dec->stream), QOA_DECODE_IO_ERROR);
    HIP_TRY(dec->tables.run((const uint8_t *)dec->d_in.p, (int16_t *)dec->d_out.p, dec->stream), QOA_DECODE_IO_ERROR);
    HIP_TRY(hipMemcpyAsync(dec->pcm.data() + base, dec->d_out.p, n_i16 * 2, hipMemcpyDeviceToHost, dec->stream),
            QOA_DECODE_IO_ERROR);
    HIP_TRY(hipStreamSynchronize(dec->stream), QOA_DECODE_IO_ERROR);
    return QOA_OK;
}

// Walk whatever bytes are unread, decode the frames found.  Returns 0 or a HIP failure code.
static int decoder_advance(qoa_decoder *dec) {
    size_t first_new = dec->frames.size();
    dec->terminal = dec->w.walk(dec->data.data(), dec->data.size(), &dec->pos, dec->frames);
    // a reader that FAILED (not: ended) at a frame boundary is an IoError, not the clean end a short header read means
    // (src/lib.rs:206-216 treats only UnexpectedEof as the end)
    if (dec->read_failed && dec->terminal == kTermEnd) dec->terminal = QOA_DECODE_IO_ERROR;
    return decoder_decode_new_frames(dec, first_new);
}

// lazy mode: the iterator has used up the frames it has -- fetch one more frame, if the reader has one
static bool decoder_refill(qoa_decoder *dec, int *rc) {
    *rc = QOA_OK;
    if (!dec->read || dec->read_eof || dec->read_failed || dec->terminal != kTermEnd) return false;
    const size_t before = dec->data.size();
    reader_pull_frame(dec);
    if (dec->data.size() == before && !dec->read_failed) return false; // clean end of stream
    *rc = decoder_advance(dec);
    return true;
}

extern "C" int qoa_decoder_new(const uint8_t *bytes, size_t len, qoa_decoder **out) try { // src/lib.rs:127-169
    if (!out) return fail_msg("null out pointer", QOA_DECODE_IO_ERROR);
    *out = nullptr;
    if (len < 4) return fail_msg("failed to fill whole buffer", QOA_DECODE_IO_ERROR); // read_u32_be -> UnexpectedEof
    if (be32(bytes) != kMagic) return QOA_DECODE_NOT_QOA_FILE;
    if (len < 8) return fail_msg("failed to fill whole buffer", QOA_DECODE_IO_ERROR);
    auto *dec = new (std::nothrow) qoa_decoder();
    if (!dec) return fail_msg("out of memory", QOA_DECODE_IO_ERROR);
    std::unique_ptr<qoa_decoder> guard(dec); // (released on every way out, a throw included)
    dec->data.assign(bytes, bytes + len);
    dec->pos = 8;
    uint32_t samples = be32(bytes + 4);
    dec->w.mode = samples == 0 ? kStreaming : kFixed;
    dec->w.fixed_samples = samples;
    int rc = decoder_advance(dec);
    if (rc) { guard.reset(); return rc; }
    if (dec->w.mode == kFixed) {
        // the first frame header + LMS are parsed inside new(): its errors are new()'s errors (:161-166)
        if (dec->frames.empty()) {
            int t = dec->terminal;
            guard.reset();
            if (t == kTermEnd) return QOA_DECODE_NO_SAMPLES;
            if (t == QOA_DECODE_IO_ERROR) g_detail = "failed to fill whole buffer";
            return t;
        }
        dec->current.num_channels = dec->frames[0].channels;
        dec->current.sample_rate = dec->frames[0].rate;
        dec->current.num_samples_per_channel = dec->frames[0].nspc;
    }
    *out = guard.release();
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

// QoaDecoder::new over a reader that is pulled lazily (src/lib.rs:127-169): the magic and the sample count now, the first
// frame inside new() when the count is non-zero (the reference parses its header + LMS there, :161-166), every later frame
// when Iterator::next runs out of samples.
extern "C" int qoa_decoder_new_reader(qoa_read_fn read, void *ctx, qoa_decoder **out) try {
    if (!out) return fail_msg("null out pointer", QOA_DECODE_IO_ERROR);
    *out = nullptr;
    if (!read) return fail_msg("null reader callback", QOA_DECODE_IO_ERROR);
    auto *dec = new (std::nothrow) qoa_decoder();
    if (!dec) return fail_msg("out of memory", QOA_DECODE_IO_ERROR);
    std::unique_ptr<qoa_decoder> guard(dec); // (released on every way out, a throw included)
    dec->read = read;
    dec->read_ctx = ctx;
    auto fail = [&](int code, const char *msg) { if (msg) g_detail = msg; guard.reset(); return code; };
    if (reader_pull(dec, 4) < 4) return fail(QOA_DECODE_IO_ERROR, dec->read_failed ? "reader callback failed" : "failed to fill whole buffer");
    if (be32(dec->data.data()) != kMagic) return fail(QOA_DECODE_NOT_QOA_FILE, nullptr);
    if (reader_pull(dec, 4) < 4) return fail(QOA_DECODE_IO_ERROR, dec->read_failed ? "reader callback failed" : "failed to fill whole buffer");
    dec->pos = 8;
    const uint32_t samples = be32(dec->data.data() + 4);
    dec->w.mode = samples == 0 ? kStreaming : kFixed;
    dec->w.fixed_samples = samples;
    if (dec->w.mode == kFixed) {
        reader_pull_frame(dec);
        const int rc = decoder_advance(dec);
        if (rc) return fail(rc, nullptr);
        if (dec->frames.empty()) {
            const int t = dec->terminal;
            if (t == kTermEnd) return fail(QOA_DECODE_NO_SAMPLES, nullptr);
            return fail(t, t == QOA_DECODE_IO_ERROR ? (dec->read_failed ? "reader callback failed" : "failed to fill whole buffer") : nullptr);
        }
        dec->current.num_channels = dec->frames[0].channels;
        dec->current.sample_rate = dec->frames[0].rate;
        dec->current.num_samples_per_channel = dec->frames[0].nspc;
    }
    *out = guard.release();
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decoder_new_streaming(qoa_decoder **out) try { // src/lib.rs:632-638
    static const uint8_t hdr[8] = {'q', 'o', 'a', 'f', 0, 0, 0, 0};
    return qoa_decoder_new(hdr, 8, out);
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" void qoa_decoder_free(qoa_decoder *dec) { delete dec; }

extern "C" int qoa_decoder_mode(const qoa_decoder *dec, qoa_mode *mode) try {
    if (!dec || !mode) return fail_msg("null pointer", QOA_DECODE_IO_ERROR);
    mode->kind = dec->w.mode == kStreaming ? QOA_MODE_STREAMING : QOA_MODE_FIXED_SAMPLES;
    mode->channels = dec->w.fixed_channels;
    mode->sample_rate = dec->w.fixed_rate;
    mode->samples = dec->w.fixed_samples;
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decoder_current_frame_header(const qoa_decoder *dec, qoa_frame_header *h) try {
    if (!dec || !h) return fail_msg("null pointer", QOA_DECODE_IO_ERROR);
    *h = dec->current;
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" double qoa_decoder_total_duration(const qoa_decoder *dec) { // src/lib.rs:190-201
    if (!dec || dec->w.mode == kStreaming) return -1.0;
    return (double)dec->w.fixed_samples / (double)dec->w.fixed_rate;
}

// Iterator::next (src/lib.rs:668-692) replayed over the pre-decoded frames.
extern "C" int qoa_decoder_next(qoa_decoder *dec, qoa_item *item) try {
    if (!dec || !item) return fail_msg("null pointer", QOA_DECODE_IO_ERROR);
    memset(item, 0, sizeof *item);
    if (dec->finished_err) { item->kind = QOA_ITEM_END; return QOA_OK; }
    for (;;) {
        bool have = dec->it_frame < dec->frames.size();
        // pending samples of the current frame come first (:669-673)
        if (have && dec->it_header_done) {
            const FrameRec &f = dec->frames[dec->it_frame];
            size_t n = (size_t)f.n_out * f.channels;
            if (dec->it_sample < n) {
                item->kind = QOA_ITEM_SAMPLE;
                item->sample = dec->pcm[dec->frame_pcm[dec->it_frame] + dec->it_sample++];
                return QOA_OK;
            }
        }
        if (!dec->returned_first) { // :674-678 -- the header parsed in new(), or the zero default in streaming mode
            dec->returned_first = true;
            item->kind = QOA_ITEM_FRAME_HEADER;
            item->header = dec->current;
            if (dec->w.mode == kFixed) dec->it_header_done = true; // frame 0's header is this item
            return QOA_OK;
        }
        if (have && dec->it_header_done) { // frame exhausted: move on
            dec->it_frame++;
            dec->it_header_done = false;
            dec->it_sample = 0;
            continue;
        }
        if (have) { // next frame header (:684-685)
            const FrameRec &f = dec->frames[dec->it_frame];
            dec->current.num_channels = f.channels;
            dec->current.sample_rate = f.rate;
            dec->current.num_samples_per_channel = f.nspc;
            dec->it_header_done = true;
            item->kind = QOA_ITEM_FRAME_HEADER;
            item->header = dec->current;
            return QOA_OK;
        }
        {   // lazy reader: the next frame is fetched only now (src/lib.rs:684)
            int rc;
            if (decoder_refill(dec, &rc)) {
                if (rc) { item->kind = QOA_ITEM_ERROR; item->error = rc; dec->finished_err = true; return QOA_OK; }
                continue;
            }
        }
        if (dec->terminal == kTermEnd) { item->kind = QOA_ITEM_END; return QOA_OK; } // :686
        item->kind = QOA_ITEM_ERROR;
        item->error = dec->terminal;
        if (dec->terminal == QOA_DECODE_IO_ERROR) g_detail = dec->read_failed ? "reader callback failed" : "failed to fill whole buffer";
        dec->finished_err = true;
        return QOA_OK;
    }
} catch (...) { // (next() reports through the item: Some(Err(IoError)), after which the iterator is finished -- src/lib.rs:668-692)
    const int rc = on_exception(QOA_DECODE_IO_ERROR);
    if (item) { item->kind = QOA_ITEM_ERROR; item->error = rc; }
    if (dec) dec->finished_err = true;
    return QOA_OK;
}

extern "C" size_t qoa_decoder_read_samples(qoa_decoder *dec, int16_t *out, size_t cap) {
    if (!dec || !out || dec->finished_err) return 0;
    if (!(dec->it_frame < dec->frames.size() && dec->it_header_done)) return 0;
    const FrameRec &f = dec->frames[dec->it_frame];
    size_t n = (size_t)f.n_out * f.channels;
    size_t take = std::min(cap, n - dec->it_sample);
    if (take) memcpy(out, dec->pcm.data() + dec->frame_pcm[dec->it_frame] + dec->it_sample, take * sizeof(int16_t));
    dec->it_sample += take;
    return take;
}

extern "C" int qoa_decoder_decode_frame(qoa_decoder *dec, const uint8_t *frame, size_t len, int16_t *out, size_t cap,
                                        size_t *n_out) try { // src/lib.rs:641-651
    if (!dec) return fail_msg("null decoder", QOA_DECODE_IO_ERROR);
    if (dec->finished_err) { if (n_out) *n_out = 0; return QOA_OK; }
    if (len) {
        dec->data.insert(dec->data.end(), frame, frame + len);
        int rc = decoder_advance(dec);
        if (rc) return rc;
    }
    // drain the iterator: count first, so that a too-small buffer loses nothing
    size_t total = 0;
    {
        size_t fi = dec->it_frame;
        bool hd = dec->it_header_done;
        size_t si = dec->it_sample;
        for (; fi < dec->frames.size(); fi++, hd = false, si = 0) {
            (void)hd;
            total += (size_t)dec->frames[fi].n_out * dec->frames[fi].channels - si;
        }
    }
    if (n_out) *n_out = total;
    if (!out || cap < total) {
        if (total == 0 && dec->terminal != kTermEnd) { dec->finished_err = true; return dec->terminal; }
        return total ? fail_msg("output buffer too small; call again with len == 0", QOA_DECODE_IO_ERROR) : QOA_OK;
    }
    size_t w = 0;
    qoa_item it;
    for (;;) {
        qoa_decoder_next(dec, &it);
        if (it.kind == QOA_ITEM_END) break;
        if (it.kind == QOA_ITEM_ERROR) return it.error; // `item?` -- samples gathered so far are dropped
        if (it.kind == QOA_ITEM_SAMPLE) {
            out[w++] = it.sample;
            w += qoa_decoder_read_samples(dec, out + w, cap - w);
        }
    }
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }

extern "C" int qoa_decode_all(const uint8_t *bytes, size_t len, uint8_t *channels, uint32_t *sample_rate, int16_t *out,
                              size_t cap, size_t *n_out) try { // src/lib.rs:713-747
    qoa_decoder *dec = nullptr;
    int rc = qoa_decoder_new(bytes, len, &dec);
    if (rc) return rc;
    qoa_item it;
    qoa_decoder_next(dec, &it); // always the first FrameHeader (:724-731)
    uint8_t nch = it.header.num_channels;
    uint32_t sr = it.header.sample_rate;
    size_t n = 0;
    for (;;) {
        qoa_decoder_next(dec, &it);
        if (it.kind == QOA_ITEM_END) break;
        if (it.kind == QOA_ITEM_ERROR) { rc = it.error; break; }
        if (it.kind == QOA_ITEM_SAMPLE) {
            if (out && n < cap) out[n] = it.sample;
            n++;
            if (out && n < cap) n += qoa_decoder_read_samples(dec, out + n, cap - n);
            else {
                // count-only pass: skip the rest of this frame
                int16_t tmp[256];
                size_t k;
                while ((k = qoa_decoder_read_samples(dec, tmp, 256)) != 0) n += k;
            }
        } else if (it.header.num_channels != nch || it.header.sample_rate != sr) { // :735-738
            rc = QOA_DECODE_INCOMPATIBLE_FRAME;
            break;
        }
    }
    qoa_decoder_free(dec);
    if (rc) return rc;
    if (channels) *channels = nch;
    if (sample_rate) *sample_rate = sr;
    if (n_out) *n_out = n;
    return QOA_OK;
} catch (...) { return on_exception(QOA_DECODE_IO_ERROR); }
