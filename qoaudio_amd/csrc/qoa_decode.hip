// qoa_decode.hip -- the gfx950 slice decoders (reference: decode_one_slice_per_channel src/lib.rs:291-330, LMS read
// :271-281).  Three kernels, in the order of this file:
//   qoa_decode_kernel          any channel count, one lane per chain, 2-byte stores: only for streams of more than 8 channels
//   qoa_decode_frames_kernel   "K2", the general kernel for 1 .. 8 channels: short last frames, PCM off the dword grid; slices
//                              fetched per lane into AccVGPRs, 16-byte pieces out of an LDS image
//   qoa_decode_uniform_kernel  "K2u", the fast path that carries almost every frame: full 256-row frames of 1 .. 8 channels,
//                              slices staged by LDS-DMA (no AccVGPRs), whole 128-byte lines out of an LDS ring, generated rows
// Split from qoa_kernels.hip (K1) so that the kernels build independently.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qoa_device.h"
#include "qoa_kernels.h"
#include <cstdlib>
#ifndef QOA_DEC_ASM_INC
#define QOA_DEC_ASM_INC "qoa_decode_asm.inc"
#endif
#include QOA_DEC_ASM_INC

#ifndef QOA_DEC_DEPTH
#define QOA_DEC_DEPTH 2
#endif

namespace qoa {


// ------------------------------------------------------------------------------------------------
// K2 (qoa_decode_frames_kernel; the description of the I/O below is this kernel's, K2u has its own further down)
//
// A decode chain is one (stream, frame, channel): the LMS state is re-seeded from every frame header
// (src/lib.rs:270-281), so frames are independent.  One lane per chain, 20 steps per slice
// (src/lib.rs:306-317): predict on full 32-bit multiplies (a hostile stream can drive decoder weights
// anywhere), dequantise with the slice's four magnitudes (read once per slice from a 16-entry LDS table)
// through v_perm_b32, clamp, LMS update.
//
// I/O.  Input: each lane reads its own 8-byte slice per row (channels of a frame are adjacent lanes, so a
// wave touches 64/NCH lines per row and re-uses each line for 128/(8 NCH) rows out of L1/L2); the next
// row's slice is loaded before the current one is decoded.  Output: int16 samples at stride NCH are
// scattered 2 bytes at a time, which a global store cannot coalesce -- so every lane writes its samples
// into an LDS image of the frame's interleaved PCM (R rows = R*20 sample-times per frame group), and the
// wave then copies the image out as 16-byte pieces of contiguous global memory.  The group stride in LDS
// is an odd number of dwords, so the 2-byte scatter is bank-conflict free across frames.
// ------------------------------------------------------------------------------------------------
// K2 packed row.  A frame starts from i16 weights (they come from the header, src/lib.rs:276-281) and the history
// is always i16, so the decoder can use the same i16-pair arithmetic as K1's fast path: 2 v_dot2_i32_i16 for the
// prediction, 2 v_pk_mad_i16 for the update.  Exactness is a per-row guard here: one sample moves a weight by at
// most |dequant >> 4| <= ceil(mag3(sf) / 16), so a row of 20 samples whose weights all satisfy
// |w| + 20 * ceil(mag3 / 16) <= 32767 at its start cannot leave i16; the products are then exact and the i32
// accumulation of v_dot2 is the reference's wrapping sum (src/lib.rs:799-805).  Rows that fail the guard (a hostile
// stream can drive decoder weights anywhere) run the 32-bit step from the unpacked state, for the whole wave.
// ------------------------------------------------------------------------------------------------
typedef short pk_i16 __attribute__((ext_vector_type(2)));

struct DecPk {
    unsigned w01, w23;   // (w0, w1), (w2, w3)
    unsigned pa, pm, pb; // history (h0, h1), (h1, h2), (h2, h3)
    unsigned sa, sm, sb; // their +-1 pairs
};

__device__ __forceinline__ unsigned pk_signs(unsigned pair) {
    const pk_i16 v = __builtin_bit_cast(pk_i16, pair);
    const pk_i16 m = v >> (pk_i16){15, 15};
    return __builtin_bit_cast(unsigned, m) | 0x00010001u;
}

__device__ __forceinline__ void dec_pack(const Lms &l, DecPk &p) {
    auto pack = [](int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); };
    p.w01 = pack(l.w0, l.w1); p.w23 = pack(l.w2, l.w3);
    p.pa = pack(l.h0, l.h1); p.pm = pack(l.h1, l.h2); p.pb = pack(l.h2, l.h3);
    p.sa = pk_signs(p.pa); p.sm = pk_signs(p.pm); p.sb = pk_signs(p.pb);
}

__device__ __forceinline__ void dec_unpack(const DecPk &p, Lms &l, int &s0, int &s1, int &s2, int &s3) {
    l.w0 = (int)(p.w01 << 16) >> 16; l.w1 = (int)p.w01 >> 16;
    l.w2 = (int)(p.w23 << 16) >> 16; l.w3 = (int)p.w23 >> 16;
    l.h0 = (int)(p.pa << 16) >> 16; l.h1 = (int)p.pa >> 16;
    l.h2 = (int)(p.pb << 16) >> 16; l.h3 = (int)p.pb >> 16;
    sign_mults(l, s0, s1, s2, s3);
}

// |w_i| <= limit for the four packed weights (limit in 0..32767)
__device__ __forceinline__ bool dec_weights_within(const DecPk &p, int limit) {
    const pk_i16 a = __builtin_bit_cast(pk_i16, p.w01), b = __builtin_bit_cast(pk_i16, p.w23);
    const pk_i16 hi = __builtin_elementwise_max(a, b), lo = __builtin_elementwise_min(a, b);
    return hi.x <= limit && hi.y <= limit && lo.x >= -limit && lo.y >= -limit;
}

// 32767 - max |w_i| of the four packed weights (a weight of -32768 gives -1)
__device__ __forceinline__ int dec_slack(const DecPk &p) {
    const int w0 = (int)(p.w01 << 16) >> 16, w1 = (int)p.w01 >> 16, w2 = (int)(p.w23 << 16) >> 16, w3 = (int)p.w23 >> 16;
    const int m = max(max(abs(w0), abs(w1)), max(abs(w2), abs(w3)));
    return 32767 - m;
}

__device__ __forceinline__ bool fits_i16(int v) { return (unsigned)(v + 32768) < 65536u; }

// One decoded sample on the packed state.  `entry` comes from the LDS dequant table: low half = the dequantised
// residual (src/lib.rs:309), high half = that value >> 4, the weight step of src/lib.rs:822 -- so the sample costs no
// code-to-magnitude arithmetic at all, and the 20 lookups of a row are issued ahead of its serial chain.
// B_i = (h[i-2], h[i-1]); the three live history pairs rotate by name in the unrolled caller.
// `before` = the unclamped reconstruction of the sample before (h3 at a row's first sample), updated to this one's: the next
// history pair (h3, recon) is ONE v_cvt_pk_i16_i32 of the two -- it clamps both (:310; clamping the older value again gives
// the same i16) and packs them (:827).  Returns the pair; the new sample is its high half.
__device__ __forceinline__ unsigned decode_step_pk(unsigned &w01, unsigned &w23, unsigned A, unsigned B, unsigned SA, unsigned SB,
                                                   int &before, unsigned entry) {
    int p = __builtin_amdgcn_sdot2(__builtin_bit_cast(pk_i16, w01), __builtin_bit_cast(pk_i16, A), 0, false);
    p = __builtin_amdgcn_sdot2(__builtin_bit_cast(pk_i16, w23), __builtin_bit_cast(pk_i16, B), p, false);
    const int deq = (int)(short)(entry & 0xffffu);
    const int u = (p >> 13) + deq;
    const short delta = (short)(entry >> 16);
    const pk_i16 d2 = {delta, delta};
    w01 = __builtin_bit_cast(unsigned, (pk_i16)(__builtin_bit_cast(pk_i16, SA) * d2 + __builtin_bit_cast(pk_i16, w01)));
    w23 = __builtin_bit_cast(unsigned, (pk_i16)(__builtin_bit_cast(pk_i16, SB) * d2 + __builtin_bit_cast(pk_i16, w23)));
    const unsigned Bn = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(before, u));
    before = u;
    return Bn;
}

// LDS dequant table for the packed rows: [sf][code] = (dequant & 0xffff) | ((dequant >> 4) << 16)
__device__ __forceinline__ void init_deq_table(unsigned (*s_deq)[8]) {
    for (unsigned t = threadIdx.x; t < 128; t += blockDim.x) { // (a block may be a single wavefront)
        const unsigned sf = t >> 3, code = t & 7u;
        const int mag = c_mags[sf][code >> 1];
        const int deq = (code & 1u) ? -mag : mag; // QOA_DEQUANT_TAB, src/lib.rs:847-864
        s_deq[sf][code] = ((unsigned)deq & 0xffffu) | ((unsigned)(deq >> 4) << 16);
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_init_lms(const uint8_t *f, unsigned channel, Lms &l) {
    // 4 x i16 history, 4 x i16 weights, big-endian, sign-extended (src/lib.rs:271-281)
    const unsigned *s = (const unsigned *)(f + 8 + 16 * (size_t)channel);
    unsigned a = bswap32(s[0]), b = bswap32(s[1]), c = bswap32(s[2]), d = bswap32(s[3]);
    l.h0 = (int)a >> 16; l.h1 = (int)(a << 16) >> 16; l.h2 = (int)b >> 16; l.h3 = (int)(b << 16) >> 16;
    l.w0 = (int)c >> 16; l.w1 = (int)(c << 16) >> 16; l.w2 = (int)d >> 16; l.w3 = (int)(d << 16) >> 16;
}

// One decoded sample (src/lib.rs:307-316).  `code` is the 3-bit quantised residual.
__device__ __forceinline__ int decode_step(Lms &l, int &s0, int &s1, int &s2, int &s3, unsigned code, unsigned dq_lo,
                                           unsigned dq_hi) {
    unsigned long long acc = (unsigned long long)(unsigned)mul32(l.w0, l.h0);
    acc = mad32(l.w1, l.h1, acc);
    acc = mad32(l.w2, l.h2, acc);
    acc = mad32(l.w3, l.h3, acc);
    int pred = (int)(unsigned)acc >> 13;
    int sneg = -(int)(code & 1u);
    unsigned sel = (code & 6u) * 0x0101u + 0x0c0c0100u;
    int mag = (int)__builtin_amdgcn_perm(dq_hi, dq_lo, sel);
    int deq = (mag ^ sneg) - sneg;
    int recon = med3(pred + deq, -32768, 32767);
    int delta = deq >> 4;
    l.w0 = mad24(s0, delta, l.w0); l.w1 = mad24(s1, delta, l.w1);
    l.w2 = mad24(s2, delta, l.w2); l.w3 = mad24(s3, delta, l.w3);
    l.h0 = l.h1; l.h1 = l.h2; l.h2 = l.h3; l.h3 = recon;
    s0 = s1; s1 = s2; s2 = s3; s3 = (recon >> 31) | 1;
    return recon;
}

__device__ __forceinline__ void init_mag_table(uint2 *s_mags) {
    if (threadIdx.x < 16) {
        s_mags[threadIdx.x] = make_uint2((unsigned)c_mags[threadIdx.x][0] | ((unsigned)c_mags[threadIdx.x][1] << 16),
                                         (unsigned)c_mags[threadIdx.x][2] | ((unsigned)c_mags[threadIdx.x][3] << 16));
    }
    __syncthreads();
}

#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 1 // (the generic kernel: part 1 of the object; see the launchers)
// K2 generic: any channel count (1..255), one lane per chain, direct 2-byte global stores.  Used for the
// exotic streams only (more than 8 channels); everything else goes through qoa_decode_frames_kernel.
__global__ void __launch_bounds__(256) qoa_decode_kernel(const uint8_t *__restrict__ in, int16_t *__restrict__ out,
                                                         const DecChain *__restrict__ chains, unsigned n_chains,
                                                         unsigned *__restrict__ mismatches) {
    __shared__ uint2 s_mags[16];
    init_mag_table(s_mags);
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_chains) return;
    const DecChain ch = chains[id];
    const uint8_t *f = in + ch.frame_off;
    const unsigned nch = ch.channels;
    if ((ch.expect_hi | ch.expect_lo) != 0 && ch.channel == 0) {
        unsigned hi = bswap32(*(const unsigned *)f), lo = bswap32(*(const unsigned *)(f + 4));
        if (hi != ch.expect_hi || lo != ch.expect_lo) atomicAdd(mismatches, 1u);
    }
    Lms l;
    decode_init_lms(f, ch.channel, l);
    int s0, s1, s2, s3;
    sign_mults(l, s0, s1, s2, s3);
    const uint8_t *sp = f + 8 + 16 * (size_t)nch + 8 * (size_t)ch.channel;
    int16_t *op = out + ch.out_off;
    unsigned remaining = ch.n_out;
    for (unsigned row = 0; row < ch.rows; row++) {
        const unsigned hi = bswap32(*(const unsigned *)sp), lo = bswap32(*(const unsigned *)(sp + 4));
        sp += 8 * (size_t)nch;
        const unsigned long long slice = ((unsigned long long)hi << 32) | lo;
        const uint2 dq = s_mags[hi >> 28];
        int16_t rec[20];
#pragma unroll
        for (int i = 0; i < 20; i++)
            rec[i] = (int16_t)decode_step(l, s0, s1, s2, s3, (unsigned)(slice >> (57 - 3 * i)) & 7u, dq.x, dq.y);
        const unsigned n = min(remaining, 20u); // tail: only `remaining` samples are exposed (src/lib.rs:320-328)
#pragma unroll
        for (int i = 0; i < 20; i++) if ((unsigned)i < n) op[(size_t)i * nch] = rec[i];
        op += (size_t)20 * nch;
        remaining -= n;
    }
}

#endif // part 1
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 1 || QOA_DECODE_PART == 4 // (the general kernel: its whole-piece / PARTIAL variants)
// K2: NCH in 1..8, one table entry per FRAME, 64/NCH frames per wavefront, LDS-transposed output.
// `scratch` points at kDecodeScratchBytes of device memory that absorbs the 16-byte stores of pieces that lie
// beyond a frame's end: making every flush store unconditional keeps their count static, so hipcc can wait
// for the prefetched slice with a counted vmcnt instead of draining the stores.
template <int NCH, int R, int WPB, bool PARTIAL>
__global__ void __launch_bounds__(64 * WPB) qoa_decode_frames_kernel(const uint8_t *__restrict__ in, int16_t *__restrict__ out,
                                                                const DecChain *__restrict__ frames, unsigned n_frames,
                                                                unsigned *__restrict__ mismatches,
                                                                uint8_t *__restrict__ scratch) {
    constexpr int FPW = 64 / NCH;               // frame groups per wavefront
    constexpr int GS = R * 20 * NCH * 2;        // bytes of one group's PCM image (R rows)
    constexpr int GSP = GS + 4;                 // odd dword stride: conflict-free 2-byte scatter across groups
    constexpr int PPG = GS / 16;                // 16-byte pieces per group image
    constexpr int NP = (FPW * PPG + 63) / 64;   // pieces per lane per flush
    static_assert(GS % 16 == 0, "group image must be a whole number of 16-byte pieces");
    // When NCH does not divide 64 the last few lanes of a wave belong to no frame group (g == FPW).  They run the same
    // code -- they serve flush pieces, ballots and bpermutes -- so they need an image slot of their own to scribble
    // into: without it their row writes land in the NEXT wave's image (found by fuzzing multi-wave 3/5/6/7-channel
    // batches; tests/test_gpu_parity.py::test_decode_channel_counts_that_do_not_divide_64).
    constexpr int IMG_GROUPS = FPW + ((64 % NCH) ? 1 : 0);
    __shared__ __attribute__((aligned(16))) uint8_t s_img[WPB][IMG_GROUPS * GSP + 12];
    __shared__ uint2 s_mags[16];
    __shared__ unsigned s_deq[16][8];
    init_deq_table(s_deq);
    init_mag_table(s_mags); // (ends with the block barrier)

    // This kernel's launches usually run BESIDE a K2u launch that fills the chip (short last frames of the same batch, on the
    // plan's side stream): its few wavefronts are serial chains of up to 256 rows, and sharing a SIMD's issue slots evenly with
    // three K2u wavefronts stretched them from 0.14 ms to 1-2 ms -- the whole decode then waited for them.  They issue first.
    __builtin_amdgcn_s_setprio(2);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned g = lane / NCH, c = lane % NCH;
    const unsigned frame_id = (blockIdx.x * (unsigned)WPB + wave) * FPW + g;
    const bool valid = (g < (unsigned)FPW) && frame_id < n_frames;
    DecChain fr;
    if (valid) fr = frames[frame_id];
    else { fr.frame_off = 0; fr.out_off = 0; fr.n_out = 0; fr.rows = 0; fr.channels = NCH; fr.channel = 0; fr.expect_hi = fr.expect_lo = 0; }
    const unsigned rows = valid ? fr.rows : 0u;
    const uint8_t *f = in + fr.frame_off;
    Lms l = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        if ((fr.expect_hi | fr.expect_lo) != 0 && c == 0) {
            unsigned hi = bswap32(*(const unsigned *)f), lo = bswap32(*(const unsigned *)(f + 4));
            if (hi != fr.expect_hi || lo != fr.expect_lo) atomicAdd(mismatches, 1u);
        }
        decode_init_lms(f, c, l);
    }
    int s0, s1, s2, s3;
    sign_mults(l, s0, s1, s2, s3);
    DecPk pk;          // the chain's state while `packed` (l is then stale)
    bool packed = true; // header values are i16
    dec_pack(l, pk);
    // Slice loads are UNCONDITIONAL and clamped (a branch around a load makes hipcc wait for it at the join,
    // i.e. immediately): lanes without a frame, and rows past a frame's end, re-read a valid slice of lane 0's
    // frame -- every table entry has rows >= 1 -- and ignore it.
    const uint8_t *sp = f + 8 + 16 * NCH + 8 * c;
    unsigned rows_ld = rows;
    {
        const unsigned long long p0 = (unsigned long long)sp; // lane 0 of the wave holds the wave's first frame
        const unsigned lo0 = __builtin_amdgcn_readfirstlane((unsigned)p0);
        const unsigned hi0 = __builtin_amdgcn_readfirstlane((unsigned)(p0 >> 32));
        const unsigned rows0 = __builtin_amdgcn_readfirstlane(rows);
        if (rows0 == 0) return; // whole wave beyond the table
        if (!valid) {
            sp = (const uint8_t *)(((unsigned long long)hi0 << 32) | lo0);
            rows_ld = rows0;
        }
    }
    uint8_t *img = s_img[wave];
    const unsigned my_img = g * GSP + c * 2;

    // Flush plan, fixed for the whole kernel: piece t of this lane is 16 bytes at byte k*16 of group gg's image.
    // ptr[t] walks the group's output by GS bytes per flush; left[t] = bytes of the frame's output that remain
    // at and after this piece (<= 0: nothing to store).
    unsigned long long ptr[NP];
    int left[NP];
    unsigned src_off[NP];
    {
        const unsigned long long my_out = (unsigned long long)(out + fr.out_off);
        const int my_bytes = (int)(fr.n_out * (unsigned)(NCH * 2));
#pragma unroll
        for (int t = 0; t < NP; t++) {
            const unsigned p = lane + 64u * t;
            const unsigned gg = min(p / PPG, (unsigned)(FPW - 1)), k = p % PPG;
            const int from = (int)((gg * NCH) << 2);
            const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)my_out);
            const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(my_out >> 32));
            const int bytes = __builtin_amdgcn_ds_bpermute(from, my_bytes);
            ptr[t] = ((((unsigned long long)hi) << 32) | lo) + k * 16;
            left[t] = (p < (unsigned)(FPW * PPG)) ? bytes - (int)(k * 16) : 0;
            src_off[t] = gg * GSP + k * 16;
        }
    }
    uint8_t *const dummy = scratch + lane * 16;

    // Slice loads.  vmcnt retires loads and stores in issue order, so a load issued AFTER a flush's stores cannot be
    // waited for without waiting for those stores to reach memory first (a ~6 us stall per flush when the next
    // row's slice was fetched one row ahead).  Hence the R slices of the NEXT group are requested together, right
    // before the current group's flush stores, and row r of the next group waits with vmcnt(NP + R-1-r): its own
    // load retired, the NP stores behind it still in flight.  hipcc cannot express that wait -- at the loop header
    // it merges the pre-loop state (loads only) with the back edge (loads + stores) into a near-zero count -- so
    // the loads and their waits are asm here, invisible to its vmcnt bookkeeping (the loop has no other loads).
    // (This AccVGPR protocol is K2's only; K2u stages its slices by LDS-DMA and holds no load in a register.)
    // The destination of a load in flight must not be a register hipcc knows about: it believes an asm output is ready
    // when the statement ends and may copy or spill it before the data has arrived.  The slices therefore land in
    // AccVGPRs a[2k : 2k+1] (k = slot of (group set, row)), which hipcc never allocates in this kernel, and reach ordinary
    // registers only inside the statement that has waited for them (tests/test_k2_disasm.py checks the code object).
    auto load_row = [&](auto slotc, unsigned row) {
        constexpr int slot = decltype(slotc)::value;
        const uint8_t *a = sp + (size_t)min(row, rows_ld - 1) * (8 * NCH);
        // "memory": hipcc must not move its own loads/stores across this one -- the counted waits below rely on
        // exactly NP (or more) stores being issued between these loads and the next group's first wait
        asm volatile("global_load_dwordx2 a[%1:%2], %0, off" : : "v"(a), "n"(2 * slot), "n"(2 * slot + 1)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14",
                       "a15");
    };
    // Two groups ahead: the CU's waves share one path into L1, so a load queues behind every store the other waves
    // have in flight; requested a whole group earlier it has that group's decode time to come back.  Two register
    // sets, used alternately (a set is refilled by the group that just consumed it).
    constexpr int D = QOA_DEC_DEPTH; // groups of slices in flight
    static_assert(D * R * 2 <= 16, "AccVGPR slots of the slice prefetch");
    static_for<0, D * R>([&](auto kc) { load_row(kc, (unsigned)decltype(kc)::value); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the first two groups: nothing is behind these loads yet
    for (unsigned row00 = 0; __builtin_amdgcn_ballot_w64(row00 < rows) != 0ull; row00 += D * R) {
      static_for<0, D>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const unsigned row0 = row00 + (unsigned)(p * R);
        if (p > 0 && __builtin_amdgcn_ballot_w64(row0 < rows) == 0ull) return; // the wave's frames ended earlier in this round
        static_for<0, R>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            // in flight, oldest first: loads r..R-1 of this group, the NP stores of the flush before last, the next
            // group's R loads, the last flush's NP stores
            unsigned raw_lo, raw_hi;
            asm volatile("s_waitcnt vmcnt(%2)\n\tv_accvgpr_read_b32 %0, a%3\n\tv_accvgpr_read_b32 %1, a%4"
                         : "=v"(raw_lo), "=v"(raw_hi) : "n"(D * NP + D * R - 1 - r), "n"(2 * (p * R + r)), "n"(2 * (p * R + r) + 1)
                         : "memory");
            const unsigned hi = bswap32(raw_lo), lo = bswap32(raw_hi);
            const unsigned *deq_row = s_deq[hi >> 28];
            const unsigned long long slice = ((unsigned long long)hi << 32) | lo;
            unsigned entry[20];
#pragma unroll
            for (int i = 0; i < 20; i++) entry[i] = deq_row[(unsigned)(slice >> (57 - 3 * i)) & 7u];
            // the largest weight step this row's scalefactor allows: |(-mag3) >> 4| = ceil(mag3 / 16), the code-7 entry
            const int dmax = -((int)deq_row[7] >> 16);
            // lanes without a frame, and lanes past their frame's last row, decode a borrowed slice whose result is never
            // stored: they must not vote the wave onto the 32-bit step
            const bool live = valid && row0 + (unsigned)r < rows;
            const bool row_ok = !live || (packed && dec_weights_within(pk, 32767 - 20 * dmax));
            if (__builtin_amdgcn_ballot_w64(!row_ok) == 0ull) {
                // pair ring as in K1: B_i = (h[i-2], h[i-1]) lives in P[i % 4]; step i reads P[(i+2)%4] and P[i%4]
                unsigned P[4] = {pk.pb, 0u, pk.pa, pk.pm}, S[4] = {pk.sb, 0u, pk.sa, pk.sm};
                int before = (int)pk.pb >> 16; // h3
#pragma unroll
                for (int i = 0; i < 20; i++) {
                    P[(i + 1) % 4] = decode_step_pk(pk.w01, pk.w23, P[(i + 2) % 4], P[i % 4], S[(i + 2) % 4], S[i % 4], before,
                                                    entry[i]);
                    S[(i + 1) % 4] = pk_signs(P[(i + 1) % 4]);
                    *(int16_t *)(img + my_img + (r * 20 + i) * (NCH * 2)) = (int16_t)(P[(i + 1) % 4] >> 16);
                }
                pk.pa = P[2]; pk.pm = P[3]; pk.pb = P[0]; // 20 % 4 == 0: the ring is back in place
                pk.sa = S[2]; pk.sm = S[3]; pk.sb = S[0];
            } else {
                if (packed) dec_unpack(pk, l, s0, s1, s2, s3);
                const uint2 dq = s_mags[hi >> 28];
#pragma unroll
                for (int i = 0; i < 20; i++) {
                    int rec = decode_step(l, s0, s1, s2, s3, (unsigned)(slice >> (57 - 3 * i)) & 7u, dq.x, dq.y);
                    *(int16_t *)(img + my_img + (r * 20 + i) * (NCH * 2)) = (int16_t)rec;
                }
                packed = fits_i16(l.w0) && fits_i16(l.w1) && fits_i16(l.w2) && fits_i16(l.w3);
                dec_pack(l, pk);
            }
        });
        static_for<0, R>([&](auto rc) { // D groups ahead, before the stores below
            load_row(std::integral_constant<int, p * R + decltype(rc)::value>{}, row0 + D * R + (unsigned)decltype(rc)::value);
        });
        // flush R rows of every frame group: LDS image -> contiguous global memory, 16 bytes per lane-step.
        // All LDS reads first (one wait), then NP unconditional stores.
        typedef unsigned u4_unaligned __attribute__((ext_vector_type(4), aligned(2)));
        typedef unsigned u32_alias __attribute__((may_alias));
        asm volatile("" ::: "memory"); // the image was written as int16: keep the dword reads below behind those stores
        u4_unaligned v[NP];
#pragma unroll
        for (int t = 0; t < NP; t++) {
            const u32_alias *src = (const u32_alias *)(img + src_off[t]);
            v[t] = u4_unaligned{src[0], src[1], src[2], src[3]};
        }
        asm volatile("" ::: "memory");
        bool any_partial = false;
#pragma unroll
        for (int t = 0; t < NP; t++) {
            // global address space spelled out: a generic pointer here becomes flat_store, which hipcc can only
            // wait for with vmcnt(0) & lgkmcnt(0)
            typedef __attribute__((address_space(1))) u4_unaligned *gptr_t;
            const unsigned long long dst = left[t] >= 16 ? ptr[t] : (unsigned long long)dummy;
// Non-temporal stores measured on MI355X (config-3 shapes): +13 % at 2 channels, +3 % at 4 and 8, but
            // -8 % / -1.5 % at 1 / 3 channels (short odd-sized pieces) -> used for even channel counts only.
            if (NCH % 2 == 0) __builtin_nontemporal_store(v[t], (gptr_t)dst);
            else *(gptr_t)dst = v[t];
            any_partial |= (left[t] > 0 && left[t] < 16);
        }
        // PARTIAL = false is launched when every frame's PCM is a whole number of 16-byte pieces (always true for
        // full 5120-sample frames): with no store under a branch hipcc can count the stores and waits for the
        // next prefetched slice with vmcnt(NP) instead of vmcnt(0), i.e. it never drains the write queue.
        if (PARTIAL && __builtin_amdgcn_ballot_w64(any_partial) != 0ull) { // last piece of a short frame: 2-byte stores
#pragma unroll
            for (int t = 0; t < NP; t++) {
                if (left[t] > 0 && left[t] < 16) {
                    const unsigned n16 = (unsigned)left[t] >> 1;
                    const int16_t *src16 = (const int16_t *)(img + src_off[t]);
                    for (unsigned q = 0; q < n16; q++) ((int16_t *)ptr[t])[q] = src16[q];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NP; t++) { ptr[t] += GS; left[t] -= GS; }
      });
    }
}

#endif // part 1
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART >= 2 // (the K2u kernel template: parts 2 and 3 instantiate different channel counts)
// ------------------------------------------------------------------------------------------------
// K2u: the fast path for UNIFORM frames -- rows == 256 and n_out == 5120 (every frame of a clip but its last), 1 .. 8
// channels (round 4: also 3 / 5 / 6 / 7, whose frames do not fill a wavefront: K2uCfg, "mirror" lanes), PCM on a dword grid
// (streamed stores on a 128-byte grid, write-back stores off it).  Short last frames and PCM off the dword grid stay on
// qoa_decode_frames_kernel above; the plan splits its frame table accordingly (qoa_capi.cpp, DecTables) and launches that
// kernel beside this one.
//
// What bounds it (profiles/r03_k2lab_log.md): the MEMORY SKELETON.  The same kernel with the sample arithmetic removed
// (slice loads, stage, flush reads and stores only) takes as long as the whole kernel (1.5 ms on config 3); with the slices
// served from L2 or the Infinity Cache it takes 0.95-1.05 ms (the rate of a pure fill of the 5.4 GB), i.e. the 1.1 GB of
// HBM reads mixed into the store stream cost 0.3-0.5 ms whatever their granularity (64 / 128 bytes per frame and request),
// alignment, depth of prefetch (one or two units ahead) or cache policy.  The 11.4 VALU per sample are hidden behind it:
// a row with 13 % fewer VALU (16-byte table entries + v_mad_i64_i32) measured the same time.  Three things got it here:
//
//  * Whole-line stores.  A ROUND is 16/NCH slice rows of every frame of the wave: exactly 128 bytes of slices in and
//    640 bytes = 5 whole 128-byte lines of PCM out per frame.  Lanes scatter their samples into a per-frame LDS RING;
//    after each phase of PR rows the wave stores the lines that are complete (8 consecutive lanes = one aligned line of
//    one frame, NLS = FPW/8 store instructions per line; one ds_read_b128 per lane and line), the odd part of a phase
//    stays in the ring for the next one.  The old kernel's 320-byte pieces made every second store a half line: 4.4 vs
//    5.7 TB/s in tools/ubench/membench.
//  * Coalesced, asynchronous slice loads.  One 8-byte load per lane per row touched 32 different cache lines per
//    instruction and, queued behind the CU's own stores, cost 0.6 ms.  Now the wave fetches a staging UNIT (HR rows = 64
//    bytes per frame) with 16-byte-per-lane LDS-DMA loads (global_load_lds_dwordx4), 4 consecutive lanes per frame, one
//    unit ahead, straight into its LDS stage; every lane reads its own 8-byte slices back from there at the unit top.  No
//    register holds a load in flight (round 2 parked them in AccVGPRs).  vmcnt waits are exact and static: behind a unit's
//    loads there are only the stores of the unit before (K2uCfg::behind).
//  * The row as ONE generated asm statement (qoa_decode_asm.inc, tools/gen_decode_asm.py): 9 VALU per sample + 1.5 for
//    the lookups in a PAIR table (two codes per LDS read, both weight steps from one packed shift); hipcc's version of
//    the same row spends 14.4.  Its range guard (see the generator) keeps a running lower bound of 32767 - max|w|
//    instead of recomputing it every row; a row it cannot prove exact executes nothing and reports flag = 1, the caller
//    then checks exactly and either re-enters or takes the 32-bit step (rolled loop, cold).
// LDS: 8.7 KB ring + 2 KB stage per wave (stereo), 16.6 + 2 KB (mono), + 4.3 KB of tables per block
// -> 4-wave blocks, 12 (stereo) / 8 (mono) waves per CU.  At 4 and 8 channels a round is a single staging unit and a single
// phase of 640 bytes.
// ------------------------------------------------------------------------------------------------
// ABS = true (round 5): the ABSOLUTE-LINE variant for frames whose PCM is off the 128-byte grid (on an 8-byte grid).  The
// image ring stays frame-relative, but the flush works on the frame's absolute cache lines: a lane's 16-byte piece of line m
// sits at ring offset (128 m - delta + 16 q) mod RING with delta = (frame's PCM address) mod 128, every store instruction
// again covers whole lines (streamed), and only the frame's first and last line are partial (masked pieces + one 8-byte
// store each where delta is no multiple of 16).  What changes against the aligned kernel:
//  * the ring divides the round's bytes (RING = 320 | 640), so the up to 127 bytes a round leaves behind for the next round's
//    first line are where that line's pieces look for them; its first 8 bytes are mirrored behind its end (generated rows) so
//    that a piece that starts 8 bytes before the ring's end is contiguous;
//  * a line is flushed as soon as it is complete -- whether a phase completes one line or two depends on delta, a per-frame
//    value -- in TWO store slots per phase that are always issued (static vmcnt bookkeeping), the second under a lane mask;
//    per-lane state: the round-relative end of the next line (`lend`), from which the ring offset and the address follow.
// Write-back stores on such frames -- every 128-byte run written as two half lines that meet in L2 -- ran at 0.31 of the HBM
// peak against 0.42 for the same frames on the grid (bench.py k2_extras.general, round 4).
template <int NCH, bool ABS = false>
struct K2uCfg {
    static constexpr bool POW2 = (NCH & (NCH - 1)) == 0; // 1, 2, 4, 8: the channels of a frame fill whole wavefronts
    static_assert(!ABS || NCH == 2 || NCH == 4 || NCH == 8, "absolute-line rows are generated for 2 / 4 / 8 channels (tools/gen_decode_asm.py: GEOMETRY_ABS)");
    // bytes per flushed "line" and lanes per line.  128 / 8 everywhere in the product; the lab's mono variant (QOA_K2U_MONO_HALF:
    // 64-byte half lines, a 128-byte ring per frame -> 12 instead of 8 wavefronts per CU) is measured in profiles/r04_k2lab_log.md
#ifdef QOA_K2U_MONO_HALF
    static constexpr int LNB = NCH == 1 ? 64 : 128;
#else
    static constexpr int LNB = 128;
#endif
    static constexpr int LPL = LNB / 16, FPS = 64 / LPL; // lanes per line, frames per store instruction
    static constexpr int FPW = 64 / NCH;                 // frames (groups) per wave; the 64 - FPW * NCH lanes beyond them (3, 5, 6, 7
                                                         // channels) mirror lanes of the last frame and write to a spare image
    static constexpr int LIVE = FPW * NCH;
    static constexpr int gcd16() { int g = 16, n = NCH; while (n) { int t = g % n; g = n; n = t; } return g; }
    // rows per round: the fewest whose PCM is whole 128-byte lines of every frame (40 NCH bytes per row):
    // 1 / 2 / 4 / 8 channels 16 / 8 / 4 / 2 rows = 640 B; 6 channels 8 rows = 1920 B; 3 / 5 / 7 channels 16 rows
    static constexpr int RR = 16 / gcd16();
#ifdef QOA_K2U_HR2 // (tools/k2lab: rows per staging unit of the stereo geometry)
    static constexpr int HR = NCH == 2 ? QOA_K2U_HR2 : NCH == 1 ? 4 : POW2 ? RR : 4;
#else
    // rows per staging unit.  3 / 5 / 6 / 7 channels: 4 rows = 32 NCH bytes per frame and request (round 4; 2 rows -- one load
    // instruction per unit -- were 2 .. 3.5 % slower on all four, 8 rows 1 .. 3 % slower: profiles/r04_k2lab_log.md section 7)
#ifndef QOA_K2G_HR // (tools/k2lab: rows per staging unit of the 3 / 5 / 6 / 7-channel geometry)
#define QOA_K2G_HR 4
#endif
    static constexpr int HR = NCH == 1 ? 4 : NCH == 2 ? 4 : POW2 ? RR : (QOA_K2G_HR <= RR ? QOA_K2G_HR : RR);
#endif
#ifndef QOA_K2G_PR // (tools/k2lab: rows per phase of the 3 / 5 / 6 / 7-channel geometry)
#define QOA_K2G_PR 2
#endif
    // rows per phase (flush granularity).  ABS at 4 / 8 channels: half a round (320 bytes) instead of the whole one, so that a phase
    // + the longest unfinished line (127 bytes) fit the 640-byte ring
    static constexpr int PR = ABS ? (NCH == 8 ? 1 : 2)
                                  : NCH == 1 ? (LNB == 64 ? 2 : 4) : NCH == 2 ? 2 : POW2 ? (RR < 4 ? RR : 4) : QOA_K2G_PR;
    static constexpr int PB = PR * 40 * NCH;             // PCM bytes per frame per phase
    static constexpr int ring() { // whole lines, >= one phase + the largest carried partial line, and a whole number of
                                  // sample-times (2 NCH bytes): a lane's channel offset is added to a wrapped sample-time offset,
                                  // so no sample-time may straddle the ring's end (1 / 2 / 4 / 8 channels: any line multiple)
        if (ABS) { // divides the round's bytes, holds a phase + the longest unfinished line (127 bytes)
            int r = RR * 40 * NCH;
            while (r % 2 == 0 && (r / 2) % (2 * NCH) == 0 && r / 2 >= PB + 127) r /= 2;
            return r;
        }
        int carry = 0, k = 0, pos = 0;
        for (k = 0; k < RR / PR; k++) { pos += PB; int c = pos % LNB; carry = c > carry ? c : carry; }
        int unit = LNB;
        while (unit % (2 * NCH)) unit += LNB;
        return (PB + carry + unit - 1) / unit * unit;
    }
    static constexpr int RING = ring();
    // Ring stride per frame.  Mono: an odd number of dwords -- 32 frames of a half-wave scatter into 32 different banks.
    // Otherwise RING + 16: the flush reads whole 16-byte pieces with ONE ds_read_b128 each (4 x ds_read_b32 before), and the
    // 2-byte scatter of a half-wave (at most 16 frames, stride 17 or 41 sixteen-byte units) meets at most 2 frames per bank,
    // which a ds_write hides behind its own data transfer (MI355X_MICROARCH.md, LDS: "2-way on ds_write_b32 costs nothing").
    // bytes of the ring's start written again behind its end (generated rows): 8 are needed (a frame may sit 8 bytes off the
    // 16-byte grid even at 8 channels: offsets count int16 elements), whole sample-times are written
    static constexpr int MIRROR = ABS ? (NCH == 8 ? 16 : 8) : 0;
#ifdef QOA_K2U_GSP_PAD
    static constexpr int GSP = RING + QOA_K2U_GSP_PAD;
#else
    // (ABS: RING + 8 = 82 dwords for stereo: the 16 frames of a half-wave's 2-byte scatter fall into 16 different banks, and
    // three 4-wavefront workgroups still fit a compute unit's LDS only without further padding)
    static constexpr int GSP = ABS ? RING + MIRROR : RING + (NCH == 1 ? 4 : 16);
#endif
    static constexpr int IMG_GROUPS = FPW + (LIVE < 64 ? 1 : 0); // + the spare image the mirror lanes scribble into
    // staging units in flight: the slices of unit n + NSLOT are requested at the top of unit n (each slot is one unit's
    // lane-linear LDS-DMA image)
#ifdef QOA_K2U_NSLOT
    static constexpr int NSLOT = QOA_K2U_NSLOT;
#else
    static constexpr int NSLOT = 1;
#endif
#ifdef QOA_K2U_FLUSH_PRIO // (tools/k2lab)
    static constexpr int FLUSH_PRIO = QOA_K2U_FLUSH_PRIO;
#else
    static constexpr int FLUSH_PRIO = NCH == 8 ? 0 : 3;   // issue priority while a wavefront flushes its lines (see the flush)
#endif
    // waves per block (multiples of 4 only: other counts load the four SIMDs unevenly -- profiles/r02_k2lab_log.md, run 20)
#ifndef QOA_K2U_ABS_WPB // (tools/k2lab: wavefronts per workgroup of the absolute-line variant)
#define QOA_K2U_ABS_WPB 4
#endif
    static constexpr int WPB = ABS ? QOA_K2U_ABS_WPB : 4;
    // vector-memory operations younger than unit n's loads at the top of unit n (steady state): the loads of the NSLOT - 1
    // units requested since, and the stores of the NSLOT units since
    static constexpr int behind(int u) {
        int n = (NSLOT - 1) * NLD;
        constexpr int units = RR / HR;
        for (int k = 1; k <= NSLOT; k++) n += unit_stores(((u - k) % units + units) % units);
        return n;
    }
    static constexpr int NLS = (FPW + FPS - 1) / FPS;    // store instructions per line (8 lanes = one 128-byte line of one frame);
                                                         // when FPW is no multiple of 8 the last one runs under a lane mask
    static constexpr int CB = HR * 8 * NCH;              // slice bytes per frame per staging unit (64 or 128; 16 NCH)
    static constexpr int LPC = CB / 16;                  // lanes per frame chunk
    static constexpr int FPI = 64 / LPC;                 // whole frame chunks per load instruction (its other lanes land in spare room)
    static constexpr int NLD = (FPW + FPI - 1) / FPI;    // load instructions per staging unit
    static constexpr int SSP = CB;                       // stage stride per frame within an instruction's 1024 bytes: lane-linear, as the LDS-DMA writes it
    static constexpr int SLOT = NLD * 1024;              // one unit's stage: every lane of every load instruction lands 16 bytes
    static constexpr int lines_done(int rows) { return (rows * 40 * NCH) / LNB; } // lines complete after `rows` rows of the round
    // store instructions issued during staging unit u (rows [u*HR, (u+1)*HR) of the round)
    static constexpr int ABS_SLOTS = (PB + LNB - 1) / LNB; // ABS: store slots per phase (the most lines a phase can complete)
    static constexpr int unit_stores(int u) {
        int n = 0;
        for (int row = u * HR; row < (u + 1) * HR; row++)
            if ((row + 1) % PR == 0) n += (ABS ? ABS_SLOTS : lines_done(row + 1) - lines_done(row + 1 - PR)) * NLS;
        return n;
    }
    static_assert(RR % HR == 0 && (HR % PR == 0 || PR % HR == 0) && NLD >= 1 && CB % 16 == 0 && (RR * 40 * NCH) % LNB == 0 &&
                      (POW2 ? (FPW * CB) % 1024 == 0 : LPC % NCH == 0) && LPC <= 64,
                  "K2u geometry");
};

#ifndef QOA_K2U_LOAD_POLICY
#define QOA_K2U_LOAD_POLICY ""
#endif
#ifndef QOA_K2U_WPB_OVERRIDE
#define QOA_K2U_WPB_OVERRIDE 0
#endif
template <int NCH, int WPB, bool ABS = false>
__global__ void __launch_bounds__(64 * WPB) qoa_decode_uniform_kernel(const uint8_t *__restrict__ in, int16_t *__restrict__ out,
                                                                     const DecChain *__restrict__ frames, unsigned n_frames,
                                                                     unsigned *__restrict__ mismatches) {
    typedef K2uCfg<NCH, ABS> C;
    constexpr int FPW = C::FPW, RR = C::RR, HR = C::HR, PR = C::PR, RING = C::RING, GSP = C::GSP, NLS = C::NLS, NLD = C::NLD,
                  SSP = C::SSP, LPC = C::LPC, CB = C::CB, UNITS = RR / HR, NSLOT = C::NSLOT, SLOT = C::SLOT, LIVE = C::LIVE,
                  LNB = C::LNB, LPL = C::LPL, FPS = C::FPS;
    __shared__ __attribute__((aligned(16))) uint8_t s_img[WPB][C::IMG_GROUPS * GSP + (ABS ? 8 : NCH == 1 ? 12 : 16)];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[WPB][NSLOT * SLOT];
    __shared__ uint2 s_mags[16];
    // pair table of the generated row (qoa_decode_asm.inc): [sf][code a * 8 + code b] = deq(a) | deq(b) << 16
    __shared__ __attribute__((aligned(256))) unsigned s_deq2[16][64];
    __shared__ int s_grow[16]; // 20 * ceil(mag3(sf) / 16): the most one row can move a weight
    for (unsigned t = threadIdx.x; t < 1024; t += blockDim.x) {
        const unsigned sf = t >> 6, a = (t >> 3) & 7u, b = t & 7u;
        const int da = (a & 1u) ? -c_mags[sf][a >> 1] : c_mags[sf][a >> 1]; // QOA_DEQUANT_TAB, src/lib.rs:847-864
        const int db = (b & 1u) ? -c_mags[sf][b >> 1] : c_mags[sf][b >> 1];
        s_deq2[sf][t & 63u] = ((unsigned)da & 0xffffu) | ((unsigned)db << 16);
    }
    if (threadIdx.x < 16) s_grow[threadIdx.x] = 20 * ((c_mags[threadIdx.x][3] + 15) >> 4);
    init_mag_table(s_mags); // (ends with the block barrier)

    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // lanes beyond the wave's FPW frames (3 / 5 / 6 / 7 channels) MIRROR the lane NCH below them: the same frame, channel,
    // slices, table lookups and guard votes -- they add nothing and change nothing -- but scatter into a spare image and
    // serve no store role
    const bool mirror = lane >= (unsigned)LIVE;
    const unsigned g = (mirror ? lane - NCH : lane) / NCH, c = lane % NCH;
    // XCD-aware order: the dispatcher deals workgroups to the chip's 8 XCDs round-robin (workgroup b runs on XCD b % 8:
    // tools/k1lab/small_launch.py stamps it), so with the identity mapping every XCD -- its L2, its translation caches -- walks
    // the whole batch interleaved with the others.  Here XCD x takes the x-th contiguous eighth of the blocks instead (the
    // first nb % 8 eighths hold one block more, exactly what round-robin dealing gives XCD x): -2 .. -4 % on config 3's decode,
    // -1 .. -3 % mono, same buffers, alternating builds (tools/k2lab/ab_inproc.py, profiles/r04_k2lab_log.md).
#if defined(QOA_K2U_XCD_CHUNK) // (tools/k2lab: XCD x takes chunks of QOA_K2U_XCD_CHUNK consecutive blocks, the XCDs move through the batch together)
    const unsigned loc = blockIdx.x / 8u, xcd = blockIdx.x % 8u;
    const unsigned bid = (loc / QOA_K2U_XCD_CHUNK) * (8u * QOA_K2U_XCD_CHUNK) + xcd * QOA_K2U_XCD_CHUNK + loc % QOA_K2U_XCD_CHUNK;
#elif !defined(QOA_K2U_NO_XCD_ORDER)
    const unsigned nb = gridDim.x, xcd = blockIdx.x % 8u, base = nb / 8u, extra = nb % 8u;
    const unsigned bid = xcd * base + min(xcd, extra) + blockIdx.x / 8u;
#else
    const unsigned bid = blockIdx.x;
#endif
    const unsigned frame0 = (bid * (unsigned)WPB + wave) * FPW;
    if (frame0 >= n_frames) return; // a wave of the last block beyond the table (behind the only barrier)
#ifdef QOA_K2U_LAB_STAGGER // (tools/k2lab: do the wavefronts of the first generation run in lockstep?  staggered starts, QOA_K2U_LAB_STAGGER x 3.4 us apart at most 8 steps)
    for (unsigned k = (bid * 5u + wave * 3u) % 8u * QOA_K2U_LAB_STAGGER; k > 0; k--) __builtin_amdgcn_s_sleep(127);
#endif
    // The table's last wave may hold fewer than FPW frames: its spare frame slots decode the wave's last frame once more --
    // the same slices, the same image, the same line stores to the same addresses (identical bytes: benign) -- so that
    // whatever is left of a batch does not need a launch of its own (it ran alone on the chip for a wave's lifetime: + 0.2 ms)
    const unsigned n_here = min((unsigned)FPW, n_frames - frame0);
    const DecChain fr = frames[frame0 + min(g, n_here - 1u)];
    const uint8_t *f = in + fr.frame_off;
    Lms l;
    if ((fr.expect_hi | fr.expect_lo) != 0 && c == 0 && !mirror && g < n_here) {
        unsigned hi = bswap32(*(const unsigned *)f), lo = bswap32(*(const unsigned *)(f + 4));
        if (hi != fr.expect_hi || lo != fr.expect_lo) atomicAdd(mismatches, 1u);
    }
    decode_init_lms(f, c, l);
    int s0, s1, s2, s3;
    sign_mults(l, s0, s1, s2, s3);
    DecPk pk;
    bool packed = true; // header values are i16
    dec_pack(l, pk);

    uint8_t *img = s_img[wave];
    uint8_t *stage = s_stage[wave];
    const unsigned my_img = (mirror ? (unsigned)FPW : g) * GSP + c * 2;
    const unsigned my_stage = (g / (unsigned)C::FPI) * 1024u + (g % (unsigned)C::FPI) * SSP + c * 8; // (= g * SSP where FPI chunks fill the 1024 bytes)
    // load roles: lane fetches 16-byte piece (lane % LPC) of frame q*(64/LPC) + lane/LPC, q in 0..NLD-1
    // store roles: lane writes 16-byte piece (lane % 8) of a line of frame o*8 + lane/8, o in 0..NLS-1
    unsigned long long in_ptr[NLD], out_ptr[NLS];
    {
        const unsigned long long my_in = (unsigned long long)(f + 8 + 16 * NCH);
        const unsigned long long my_out = (unsigned long long)(out + fr.out_off);
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            // (lanes beyond the last frame's chunk -- only where NCH does not divide 64 -- fetch the last frame's piece again:
            // it lands in the stage's spare lanes)
            const unsigned lf = min((unsigned)(q * C::FPI) + lane / (unsigned)LPC, (unsigned)(FPW - 1));
            const int from = (int)(lf * NCH) << 2;
            const unsigned ilo = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)my_in);
            const unsigned ihi = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(my_in >> 32));
            in_ptr[q] = ((((unsigned long long)ihi) << 32) | ilo) + (lane % (unsigned)LPC) * 16u;
        }
#pragma unroll
        for (int o = 0; o < NLS; o++) {
            const int from = (int)(min((unsigned)(o * FPS) + lane / (unsigned)LPL, (unsigned)(FPW - 1)) * NCH) << 2; // (masked roles: any valid address)
            const unsigned olo = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)my_out);
            const unsigned ohi = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(my_out >> 32));
            out_ptr[o] = ((((unsigned long long)ohi) << 32) | olo) + (lane % (unsigned)LPL) * 16u;
        }
    }
    // ABS: per store role o, this lane's piece q = lane % 8 of the role's frame.  lend = round-relative end of the frame's next
    // unflushed absolute line (128 (m + 1) - delta), rp = ring offset of this lane's piece of that line; the piece's address is
    // out_ptr[o] + lend (out_ptr[o] = frame address + 16 q - 128, advanced by the round's bytes per round like the aligned
    // kernel's).  f_lo[o] = the frame's address: pieces of its first line that lie in front of it are not written.
    unsigned lend[NLS]; // (always positive: 1 .. the round's bytes + 128)
    unsigned rp[NLS], edge_q[NLS];
    unsigned long long f_lo[NLS];
    (void)lend; (void)rp; (void)edge_q; (void)f_lo;
    if constexpr (ABS) {
#pragma unroll
        for (int o = 0; o < NLS; o++) {
            const unsigned long long fa = out_ptr[o] - (lane % (unsigned)LPL) * 16u; // the role's frame address
            const unsigned delta = (unsigned)fa & 127u;
            f_lo[o] = fa;
            lend[o] = 128u - delta;
            rp[o] = (unsigned)(RING - (int)delta + 16 * (int)(lane % (unsigned)LPL)) % (unsigned)RING; // piece q of the line that ends at lend
            // the piece that straddles the frame's start (and, 20480 NCH bytes on, its end) when delta is no multiple of 16:
            // this lane's if q == delta / 16; 8 = none
            edge_q[o] = (delta & 15u) ? delta >> 4 : 8u;
            out_ptr[o] = out_ptr[o] - 128u;
        }
    }
    // do all of the wave's frames start on a 128-byte line?  (decides the store policy of the flush, wave-uniform)
#ifdef QOA_K2U_FORCE_WRITE_BACK // (tools/k2lab)
    const bool lines_aligned = __builtin_amdgcn_ballot_w64(((unsigned long long)(out + fr.out_off)) != 0ull) == 0ull;
#else
    const bool lines_aligned = __builtin_amdgcn_ballot_w64((((unsigned long long)(out + fr.out_off)) & 127ull) != 0ull) == 0ull;
#endif
    const unsigned oct_img = (lane / (unsigned)LPL) * GSP + (lane % (unsigned)LPL) * 16u; // + o*FPS*GSP + ring offset of the line
    typedef __attribute__((address_space(3))) uint8_t *lds_ptr;

    // Asynchronous slice loads: LDS-DMA (global_load_lds_dwordx4) straight into the wave's stage -- lane L of instruction q
    // lands at stage + q * 1024 + L * 16, which with the load roles above IS the frame-major stage [frame][CB bytes].  No
    // register holds a load in flight, so nothing the compiler does between issue and wait can break it, and the kernel
    // uses no AccVGPRs (with them hipcc gives a 768-thread block only half of its 168 registers as VGPRs).  M0 = destination
    // base; it is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md, inline-asm notes).
    const unsigned stage_m0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)stage);
    auto issue_loads = [&](unsigned slot_off) { // slot_off: wave-uniform byte offset of the destination slot
        static_for<0, NLD>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" QOA_K2U_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(in_ptr[q]), "s"(stage_m0 + slot_off + (unsigned)(q * 1024)) : "memory");
            in_ptr[q] += CB;
        });
    };
    uint2 unit_raw[HR]; // the unit's slices of this chain, read back from the stage at the unit top
    (void)unit_raw;
    const unsigned tab_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)&s_deq2[0][0];
    const unsigned grow_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int *)&s_grow[0];
    const unsigned img_lds = (unsigned)(size_t)(lds_ptr)img + my_img;
    int slack = dec_slack(pk);
    (void)tab_lds; (void)grow_lds; (void)img_lds; (void)slack;
    constexpr unsigned kRounds = 256 / RR, kUnits = kRounds * UNITS;
    static_assert(kUnits > 2 * NSLOT, "staging depth");
    // unit top (n = the unit's number in the frame): its slices have landed in slot n % NSLOT -- behind them in the vmcnt
    // order are only the C::behind(u) younger operations of the steady state; the first and last NSLOT units, where fewer
    // are in flight, simply drain -- every lane takes its own rows out, and once those reads are back the DMA of unit
    // n + NSLOT may overwrite the slot
    auto unit_top = [&](auto uc, unsigned n) {
        constexpr int u = decltype(uc)::value;
        // (ABS: the frame's first 8 bytes may go out in unit 0 -- one store more than the static count -- so unit 1 drains too)
        if (n < (unsigned)NSLOT + (ABS ? 1u : 0u) || n + NSLOT > kUnits) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(C::behind(u)) : "memory");
        const unsigned slot_off = NSLOT == 1 ? 0u : (n % NSLOT) * SLOT;
#pragma unroll
        for (int r = 0; r < HR; r++) unit_raw[r] = *(const uint2 *)(stage + slot_off + my_stage + r * NCH * 8);
        if (n + NSLOT < kUnits) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // WAR: the reads above before the DMA below
            issue_loads(slot_off);
        }
    };
    for (unsigned n = 0; n < (unsigned)NSLOT; n++) issue_loads(n * SLOT);
    for (unsigned round = 0; round < kRounds; round++) {
        static_for<0, RR>([&](auto rc) {
            constexpr int row = decltype(rc)::value; // row within the round
            if constexpr (row % HR == 0)
                unit_top(std::integral_constant<int, row / HR>{}, round * UNITS + row / HR);
            {
                static_assert(ABS || (RING == (NCH == 1 ? QOA_DEC_ASM_RING_1 : NCH == 2 ? QOA_DEC_ASM_RING_2 : NCH == 3 ? QOA_DEC_ASM_RING_3
                                       : NCH == 4 ? QOA_DEC_ASM_RING_4 : NCH == 5 ? QOA_DEC_ASM_RING_5 : NCH == 6 ? QOA_DEC_ASM_RING_6
                                       : NCH == 7 ? QOA_DEC_ASM_RING_7 : QOA_DEC_ASM_RING_8) &&
                                  RR == (NCH == 1 ? QOA_DEC_ASM_ROWS_1 : NCH == 2 ? QOA_DEC_ASM_ROWS_2 : NCH == 3 ? QOA_DEC_ASM_ROWS_3
                                         : NCH == 4 ? QOA_DEC_ASM_ROWS_4 : NCH == 5 ? QOA_DEC_ASM_ROWS_5 : NCH == 6 ? QOA_DEC_ASM_ROWS_6
                                         : NCH == 7 ? QOA_DEC_ASM_ROWS_7 : QOA_DEC_ASM_ROWS_8)),
                              "generated image offsets");
                static_assert(!ABS || (RING == (NCH == 2 ? QOA_DEC_ASM_ABS_RING_2 : NCH == 4 ? QOA_DEC_ASM_ABS_RING_4 : QOA_DEC_ASM_ABS_RING_8) &&
                                       C::MIRROR == (NCH == 2 ? QOA_DEC_ASM_ABS_MIRROR_2 : NCH == 4 ? QOA_DEC_ASM_ABS_MIRROR_4 : QOA_DEC_ASM_ABS_MIRROR_8)),
                              "generated image offsets (absolute-line rows)");
                const uint2 raw = unit_raw[row % HR];
                unsigned force = 0, flag = 0;
                bool need_slow = false;
#ifdef QOA_K2U_LAB_NOCOMPUTE // lab: the memory skeleton alone (the slices are consumed by one LDS write so that they are not dead)
                *(volatile unsigned *)(img + my_img) = raw.x ^ raw.y;
                if (false)
#endif
#pragma unroll 1
                for (;;) {
#define QOA_DEC_ROW_CASE(N, R)                                                                                         \
    if constexpr (NCH == N && row == R)                                                                                \
        asm volatile(QOA_DEC_ROW_TEXT_##N##_##R                                                                        \
                     : [w01] "+v"(pk.w01), [w23] "+v"(pk.w23), [pa] "+v"(pk.pa), [pm] "+v"(pk.pm), [pb] "+v"(pk.pb),   \
                       [sa] "+v"(pk.sa), [sm] "+v"(pk.sm), [sb] "+v"(pk.sb), [slack] "+v"(slack), [flag] "=s"(flag)    \
                     : [rx] "v"(raw.x), [ry] "v"(raw.y), [tab] "s"(tab_lds), [grow] "s"(grow_lds), [img] "v"(img_lds),  \
                       [bsw] "s"(0x00010203u), [one2] "s"(0x00010001u), [force] "s"(force)                            \
                     : "memory", "vcc", "scc", QOA_DEC_ROW_CLOBBERS);
#define QOA_DEC_ROW_CASE_ABS(N, R)                                                                                     \
    if constexpr (NCH == N && row == R)                                                                                \
        asm volatile(QOA_DEC_ROW_TEXT_A##N##_##R                                                                       \
                     : [w01] "+v"(pk.w01), [w23] "+v"(pk.w23), [pa] "+v"(pk.pa), [pm] "+v"(pk.pm), [pb] "+v"(pk.pb),   \
                       [sa] "+v"(pk.sa), [sm] "+v"(pk.sm), [sb] "+v"(pk.sb), [slack] "+v"(slack), [flag] "=s"(flag)    \
                     : [rx] "v"(raw.x), [ry] "v"(raw.y), [tab] "s"(tab_lds), [grow] "s"(grow_lds), [img] "v"(img_lds),  \
                       [bsw] "s"(0x00010203u), [one2] "s"(0x00010001u), [force] "s"(force)                            \
                     : "memory", "vcc", "scc", QOA_DEC_ROW_CLOBBERS);
                    if constexpr (ABS) {
                        QOA_DEC_ROW_CASES_ABS // (generated: the rows of the absolute-line variant)
                    } else {
                        QOA_DEC_ROW_CASES // (generated: one case per (channel count, row of the round))
                    }
#undef QOA_DEC_ROW_CASE_ABS
#undef QOA_DEC_ROW_CASE
                    if (flag == 0) break;
                    // the cheap guard tripped (nothing was executed): exact test on the current weights
                    const int grow = s_grow[bswap32(raw.x) >> 28];
                    const bool ok = packed && dec_weights_within(pk, 32767 - grow);
                    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) { need_slow = true; break; }
                    slack = dec_slack(pk) - grow;
                    force = 1;
                }
                if (need_slow) {
                    const unsigned hi = bswap32(raw.x), lo = bswap32(raw.y);
                    const unsigned long long slice = ((unsigned long long)hi << 32) | lo;
                    if (packed) dec_unpack(pk, l, s0, s1, s2, s3);
                    const uint2 dq = s_mags[hi >> 28];
#pragma unroll 1
                    for (int i = 0; i < 20; i++) {
                        int rec = decode_step(l, s0, s1, s2, s3, (unsigned)(slice >> (57 - 3 * i)) & 7u, dq.x, dq.y);
                        const unsigned ro = (unsigned)((row * 20 + i) * NCH * 2) % (unsigned)RING;
                        *(int16_t *)(img + my_img + ro) = (int16_t)rec;
                        if (ABS && ro < (unsigned)C::MIRROR) *(int16_t *)(img + my_img + ro + RING) = (int16_t)rec; // (as the generated rows)
                    }
                    packed = fits_i16(l.w0) && fits_i16(l.w1) && fits_i16(l.w2) && fits_i16(l.w3);
                    dec_pack(l, pk);
                    slack = packed ? dec_slack(pk) : -0x40000000;
                }
            }
            if constexpr (ABS && (row + 1) % PR == 0) {
                // ---- absolute-line flush: the lines of every frame that are complete now, as whole cache lines ----
                constexpr int P = (row + 1) * 40 * NCH; // bytes of the round written so far
                // (pieces sit on an 8-byte grid in the ring: one ds_read_b128 each -- as 4-byte aligned vectors hipcc read them
                // with two ds_read2_b32 and an address add apiece)
                typedef unsigned u4_unaligned __attribute__((ext_vector_type(4), aligned(8)));
                typedef unsigned u2_unaligned __attribute__((ext_vector_type(2), aligned(8)));
                asm volatile("" ::: "memory"); // the image was written as int16: keep the reads below behind those stores
                u4_unaligned vv[C::ABS_SLOTS][NLS];
                unsigned long long ga[C::ABS_SLOTS][NLS], ok_mask[C::ABS_SLOTS][NLS];
                const unsigned frame_img = (lane / (unsigned)LPL) * GSP;
                static_for<0, C::ABS_SLOTS>([&](auto sc) {
                    constexpr int sl = decltype(sc)::value;
                    static_for<0, NLS>([&](auto oc) {
                        constexpr int o = decltype(oc)::value;
                        const bool valid = lend[o] <= (unsigned)P; // the frame's next line is complete
                        vv[sl][o] = *(const u4_unaligned *)(img + frame_img + o * FPS * GSP + rp[o]);
                        ga[sl][o] = out_ptr[o] + lend[o];
                        // (a piece of the frame's first line that lies in front of the frame, or straddles its start, is not written;
                        // two ballots and a scalar AND: the ballot of the combined condition costs two VALU more)
                        ok_mask[sl][o] = __builtin_amdgcn_ballot_w64(valid) & __builtin_amdgcn_ballot_w64(ga[sl][o] >= f_lo[o]);
                        const unsigned rn = rp[o] + (unsigned)LNB;
                        rp[o] = valid ? min(rn, rn - (unsigned)RING) : rp[o]; // (+ 128) mod RING
                        lend[o] += valid ? (unsigned)LNB : 0u;
                    });
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // the issue priority is raised for the store burst only (as in the aligned kernel), not for the line bookkeeping in
                // front of it: 1-2 % on under-filled launches (profiles/r05_k2u_offgrid_prio.txt)
                if constexpr (C::FLUSH_PRIO != 0) __builtin_amdgcn_s_setprio(C::FLUSH_PRIO);
                asm volatile("; K2u absolute-line flush, streamed stores" ::: "memory");
                static_for<0, C::ABS_SLOTS>([&](auto sc) {
                    constexpr int sl = decltype(sc)::value;
                    static_for<0, NLS>([&](auto oc) {
                        constexpr int o = decltype(oc)::value;
                        // always issued (the static vmcnt bookkeeping of the slice loads counts it), lanes without a complete line
                        // off -- possibly ALL lanes: a store under EXEC = 0 still passes through the vmcnt order (measured:
                        // tools/ubench/exec0_vmcnt.hip, profiles/r05_ubench_gfx950_exec0_vmcnt.txt -- behind [cold load, 4 stores
                        // with EXEC = 0] `s_waitcnt vmcnt(4)` lasts as long as behind 4 real stores, 248 vs 249 ticks; 56 with nothing)
                        unsigned long long keep;
                        const unsigned long long a_ = ga[sl][o], m_ = ok_mask[sl][o]; // (copies: an asm operand alone does not
                        const u4_unaligned v_ = vv[sl][o];                            // capture the enclosing lambda's arrays)
                        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dwordx4 %1, %2, off nt\n\t"
                                     "s_mov_b64 exec, %0"
                                     : "=&s"(keep) : "v"(a_), "v"(v_), "s"(m_) : "memory", "scc");
                    });
                });
                if constexpr (row + 1 == PR) {
                    // the frame's first bytes where its address is no multiple of 16: the 8 bytes of the straddling piece that
                    // belong to it (ring [0, 8)), once per frame -- one store more in unit 0 than the static count (unit_top)
                    if (round == 0) {
                        static_for<0, NLS>([&](auto oc) {
                            constexpr int o = decltype(oc)::value;
                            const u2_unaligned h = *(const u2_unaligned *)(img + frame_img + o * FPS * GSP);
                            const unsigned long long m = __builtin_amdgcn_ballot_w64((lane % (unsigned)LPL) == edge_q[o]);
                            unsigned long long keep;
                            const unsigned long long fa_ = f_lo[o];
                            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\t"
                                         "global_store_dwordx2 %1, %2, off\n\ts_mov_b64 exec, %0"
                                         : "=&s"(keep) : "v"(fa_), "v"(h), "s"(m) : "memory", "scc");
                        });
                    }
                }
                if constexpr (row + 1 == RR) {
#pragma unroll
                    for (int o = 0; o < NLS; o++) lend[o] -= (unsigned)(RR * 40 * NCH); // the next round counts from its own start
                }
                asm volatile("; end of the absolute-line burst" ::: "memory");
                if constexpr (C::FLUSH_PRIO != 0) __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (!ABS && (row + 1) % PR == 0) {
                // flush the lines this phase completed
                constexpr int l0 = C::lines_done(row + 1 - PR), l1 = C::lines_done(row + 1);
                typedef unsigned u4_unaligned __attribute__((ext_vector_type(4), aligned(4)));
                typedef unsigned u4_piece16 __attribute__((ext_vector_type(4), aligned(16), may_alias));
                typedef unsigned u4_piece8 __attribute__((ext_vector_type(4), aligned(8), may_alias));
                typedef std::conditional_t<GSP % 16 == 0, u4_piece16, u4_piece8> u4_piece;
                typedef unsigned u32_alias __attribute__((may_alias));
                asm volatile("" ::: "memory"); // the image was written as int16: keep the dword reads below behind those stores
                // The flushing wavefront issues ahead of the SIMD's other wavefronts (s_setprio): its LDS reads and line stores go
                // out as one burst instead of being interleaved with the others' row arithmetic.  Measured in ONE process on the
                // SAME buffers (tools/k2lab/ab_inproc.py -- the kernel's time depends on where its buffers lie, see
                // profiles/r03_k2lab_log.md): stereo 1.53-1.59 -> 1.44-1.45 ms in the slow placements, 1.39-1.40 -> 1.38 in
                // the fast ones; 1 / 4 channels -1 %; 8 channels +-2 % (left alone).  It is the STORES that want it (priority for the
                // flush's LDS reads alone does nothing); priority for the slice loads as well undoes it.
                if constexpr (C::FLUSH_PRIO != 0) __builtin_amdgcn_s_setprio(C::FLUSH_PRIO);
                // ALL of the flush's LDS reads first, one wait, then all of its stores as one burst (mono, 8 store instructions
                // per line: 1.67 -> 1.56 ms on 2048 clips; 4 channels -1 %; stereo unchanged -- tools/k2lab/ab_inproc.py)
                u4_unaligned vv[NLS][l1 - l0 > 0 ? l1 - l0 : 1];
                static_for<0, NLS>([&](auto oc) {
                    constexpr int o = decltype(oc)::value;
#pragma unroll
                    for (int ln = l0; ln < l1; ln++) {
                        if constexpr (GSP % 8 == 0) { // (16-byte aligned in the product; the lab's 8-byte ring pads read unaligned)
                            const u4_piece x = *(const u4_piece *)(img + oct_img + o * FPS * GSP + (ln * LNB) % RING);
                            vv[o][ln - l0] = u4_unaligned{x[0], x[1], x[2], x[3]};
                        } else {
                            const u32_alias *src = (const u32_alias *)(img + oct_img + o * FPS * GSP + (ln * LNB) % RING);
                            vv[o][ln - l0] = u4_unaligned{src[0], src[1], src[2], src[3]};
                        }
                    }
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // Frames on the 128-byte grid: every store instruction covers whole cache lines -- streamed past L2 (nt).  Off it
                // every 128-byte run straddles two lines, each half written by a different instruction: as write-back stores the
                // halves meet in L2 (63 full stereo frames per clip, PCM 48 bytes off the grid: 0.98 -> 0.61 ms; on the grid nt is
                // the faster one, 0.44 vs 0.45 ms -- tools/k2lab/short_frames.py).  Two copies of the burst, one wave-uniform
                // branch in front: each stays straight-line code inside the raised-priority window (tests/test_k2_disasm.py).
                auto store_burst = [&](auto streamed) {
                constexpr bool kStreamed = decltype(streamed)::value;
                // (the two copies differ only in the stores' !nontemporal metadata: LLVM would merge them into one -- dropping the
                // metadata, i.e. ALL stores write-back -- were it not for these comment-only statements that differ)
                if constexpr (kStreamed) asm volatile("; K2u flush, streamed stores" ::: "memory");
                else asm volatile("; K2u flush, write-back stores" ::: "memory");
                if constexpr (C::FLUSH_PRIO != 0) __builtin_amdgcn_s_setprio(C::FLUSH_PRIO); // (again: each copy is a window of its own)
                static_for<0, NLS>([&](auto oc) {
                    constexpr int o = decltype(oc)::value;
                    constexpr int valid = FPW - o * FPS < FPS ? FPW - o * FPS : FPS; // frames this role group really has
#pragma unroll
                    for (int ln = l0; ln < l1; ln++) {
                        if constexpr (valid == FPS) {
                            typedef __attribute__((address_space(1))) u4_unaligned *gptr_t;
                            if constexpr (kStreamed) __builtin_nontemporal_store(vv[o][ln - l0], (gptr_t)(out_ptr[o] + ln * LNB));
#ifdef QOA_K2U_OFFGRID_POLICY // (tools/k2lab: cache-policy bits of the off-grid flush's stores, e.g. "sc1" or "sc0 sc1")
                            else {
                                const unsigned long long a_ = out_ptr[o] + (unsigned)(ln * LNB / 4096 * 4096);
                                const u4_unaligned v_ = vv[o][ln - l0];
                                asm volatile("global_store_dwordx4 %0, %1, off offset:%2 " QOA_K2U_OFFGRID_POLICY
                                             : : "v"(a_), "v"(v_), "n"(ln * LNB % 4096) : "memory");
                            }
#else
                            else *(gptr_t)(out_ptr[o] + ln * LNB) = vv[o][ln - l0];
#endif
                        } else {
                            // the last role group of a frame count that is no multiple of 8: the same store under a lane mask
                            // (always issued -- the vmcnt bookkeeping of the slice loads counts it -- with the idle octets off)
                            unsigned long long keep;
                            asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dwordx4 %1, %2, off offset:%4 nt\n\t"
                                         "s_mov_b64 exec, %0"
                                         : "=&s"(keep)
                                         : "v"(out_ptr[o] + (unsigned)(ln * LNB / 4096 * 4096)), "v"(vv[o][ln - l0]),
                                           "s"((1ull << (LPL * valid)) - 1ull), "n"(ln * LNB % 4096) // (13-bit signed immediate)
                                         : "memory", "scc"); // (always nt: a handful of lanes; s_and_b64 writes SCC)
                        }
                    }
                });
                if constexpr (kStreamed) asm volatile("; end of the streamed burst" ::: "memory");
                else asm volatile("; end of the write-back burst" ::: "memory");
                if constexpr (C::FLUSH_PRIO != 0) __builtin_amdgcn_s_setprio(0);
                };
                if (lines_aligned) store_burst(std::true_type{});
                else store_burst(std::false_type{});
            }
        });
#pragma unroll
        for (int o = 0; o < NLS; o++) out_ptr[o] += RR * 40 * NCH; // the round's PCM per frame (whole lines: 640 B, or 1920 .. 4480 B)
    }
    if constexpr (ABS) {
        // ---- the frame's last line: what is left in the ring (delta bytes).  lend counts from the frame's END now: this lane's
        // piece starts at lend - 128 + 16 q <= 0; whole pieces in front of the end go out as one masked line, the piece that
        // straddles the end (delta no multiple of 16) gives its first 8 bytes ----
        typedef unsigned u4_unaligned __attribute__((ext_vector_type(4), aligned(8)));
        typedef unsigned u2_unaligned __attribute__((ext_vector_type(2), aligned(8)));
        asm volatile("" ::: "memory");
        const unsigned frame_img = (lane / (unsigned)LPL) * GSP;
        static_for<0, NLS>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            const int start = (int)lend[o] - LNB + 16 * (int)(lane % (unsigned)LPL);
            const u4_unaligned v = *(const u4_unaligned *)(img + frame_img + o * FPS * GSP + rp[o]);
            const unsigned long long a = out_ptr[o] + lend[o];
            const unsigned long long m16 = __builtin_amdgcn_ballot_w64(start + 16 <= 0);
            const unsigned long long m8 = __builtin_amdgcn_ballot_w64(start == -8);
            const u2_unaligned v8 = u2_unaligned{v[0], v[1]};
            unsigned long long keep;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dwordx4 %1, %2, off nt\n\t"
                         "s_mov_b64 exec, %0\n\ts_and_b64 exec, exec, %5\n\tglobal_store_dwordx2 %1, %4, off\n\ts_mov_b64 exec, %0"
                         : "=&s"(keep) : "v"(a), "v"(v), "s"(m16), "v"(v8), "s"(m8) : "memory", "scc");
        });
    }
}


#endif // parts 2 and 3
// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
// The object is built in four parts side by side (hipcc spends 2.5 minutes on this file as one unit): qoa_decode.o =
// QOA_DECODE_PART 1, the general kernels and their launchers; qoa_decode_u.o (qoa_decode_u.hip) = part 2, K2u for 1 / 2 / 4 / 8
// channels + the absolute-line variant; qoa_decode_v.o = part 3, K2u for 3 / 5 / 6 / 7 channels; qoa_decode_w.o = part 4, the general
// kernel's PARTIAL variants.  Without
// the macro (tools/k2lab/build_lib.sh) the file is one unit with everything.
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 1
hipError_t launch_decode(const uint8_t *in, int16_t *out, const DecChain *chains, unsigned n, unsigned *mismatches,
                         hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(qoa_decode_kernel, g, b, 0, stream, in, out, chains, n, mismatches);
    return hipGetLastError();
}

#endif // part 1
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 1 || QOA_DECODE_PART == 4
template <int R, int WPB, bool PARTIAL>
static hipError_t launch_decode_frames_r(int nch, const uint8_t *in, int16_t *out, const DecChain *frames,
                                         unsigned n_frames, unsigned *mismatches, uint8_t *scratch, hipStream_t stream) {
    dim3 b(64 * WPB);
#define QOA_LAUNCH(N)                                                                                              \
    case N: {                                                                                                      \
        unsigned per_block = (unsigned)WPB * (64u / N);                                                            \
        dim3 g((n_frames + per_block - 1) / per_block);                                                            \
        hipLaunchKernelGGL((qoa_decode_frames_kernel<N, R, WPB, PARTIAL>), g, b, 0, stream, in, out, frames, n_frames, \
                           mismatches, scratch);                                                                   \
        break;                                                                                                     \
    }
    switch (nch) {
        QOA_LAUNCH(1) QOA_LAUNCH(2) QOA_LAUNCH(3) QOA_LAUNCH(4) QOA_LAUNCH(5) QOA_LAUNCH(6) QOA_LAUNCH(7) QOA_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef QOA_LAUNCH
    return hipGetLastError();
}

#endif // part 1
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART >= 2
template <int NCH, bool ABS = false>
static hipError_t launch_decode_uniform_n(const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                          unsigned *mismatches, hipStream_t stream) {
    constexpr unsigned FPW = 64 / NCH;
    constexpr unsigned W = QOA_K2U_WPB_OVERRIDE ? QOA_K2U_WPB_OVERRIDE : K2uCfg<NCH, ABS>::WPB;
    const unsigned waves = (n_frames + FPW - 1) / FPW; // (the last wave's spare frame slots repeat its last frame)
    // ONE launch: the last block may be short of waves (they leave behind the table barrier).  A second, small launch for a
    // remainder runs after the first on the stream, alone on the chip, for as long as a full wave takes (+0.25 ms measured).
    // Work that does not give every CU a block (single clips) runs one wavefront per block, spread over the CUs.
#ifdef QOA_K2U_LAB_DYNLDS // (tools/k2lab/generations.py: unused dynamic LDS per workgroup = fewer workgroups per CU)
    const char *lab_lds = getenv("QOA_LAB_K2U_DYN_LDS");
    const unsigned dyn_lds = lab_lds ? (unsigned)atoi(lab_lds) : 0u;
#else
    const unsigned dyn_lds = 0u;
#endif
    if (waves >= 256u * W)
        hipLaunchKernelGGL((qoa_decode_uniform_kernel<NCH, W, ABS>), dim3((waves + W - 1) / W), dim3(64 * W), dyn_lds, stream, in, out,
                           frames, n_frames, mismatches);
    else
        hipLaunchKernelGGL((qoa_decode_uniform_kernel<NCH, 1, ABS>), dim3(waves), dim3(64), 0, stream, in, out, frames, n_frames,
                           mismatches);
    return hipGetLastError();
}

#endif // parts 2 and 3: the launcher template
// (part 3 of the object, qoa_decode_v.hip: the channel counts that do not divide a wavefront -- their 16-row rounds make them
// the slowest instantiations to compile)
hipError_t launch_decode_uniform_odd(int nch, const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                     unsigned *mismatches, hipStream_t stream);
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 3
hipError_t launch_decode_uniform_odd(int nch, const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                     unsigned *mismatches, hipStream_t stream) {
    switch (nch) {
    case 3: return launch_decode_uniform_n<3>(in, out, frames, n_frames, mismatches, stream);
    case 5: return launch_decode_uniform_n<5>(in, out, frames, n_frames, mismatches, stream);
    case 6: return launch_decode_uniform_n<6>(in, out, frames, n_frames, mismatches, stream);
    case 7: return launch_decode_uniform_n<7>(in, out, frames, n_frames, mismatches, stream);
    default: return hipErrorInvalidValue;
    }
}
#endif
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 2
hipError_t launch_decode_uniform(int nch, const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                 unsigned *mismatches, hipStream_t stream) {
    if (n_frames == 0) return hipSuccess;
    switch (nch) {
    case 1: return launch_decode_uniform_n<1>(in, out, frames, n_frames, mismatches, stream);
    case 2: return launch_decode_uniform_n<2>(in, out, frames, n_frames, mismatches, stream);
    case 4: return launch_decode_uniform_n<4>(in, out, frames, n_frames, mismatches, stream);
    case 8: return launch_decode_uniform_n<8>(in, out, frames, n_frames, mismatches, stream);
    default: return launch_decode_uniform_odd(nch, in, out, frames, n_frames, mismatches, stream);
    }
}

hipError_t launch_decode_uniform_abs(int nch, const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                     unsigned *mismatches, hipStream_t stream) {
    if (n_frames == 0) return hipSuccess;
    switch (nch) {
    case 2: return launch_decode_uniform_n<2, true>(in, out, frames, n_frames, mismatches, stream);
    case 4: return launch_decode_uniform_n<4, true>(in, out, frames, n_frames, mismatches, stream);
    case 8: return launch_decode_uniform_n<8, true>(in, out, frames, n_frames, mismatches, stream);
    default: return hipErrorInvalidValue;
    }
}

#endif // part 2
hipError_t launch_decode_frames_partial(int nch, const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                        unsigned *mismatches, uint8_t *scratch, hipStream_t stream);
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 1
hipError_t launch_decode_frames(int nch, bool whole_pieces, const uint8_t *in, int16_t *out, const DecChain *frames,
                                unsigned n_frames, unsigned *mismatches, uint8_t *scratch, hipStream_t stream) {
    if (n_frames == 0) return hipSuccess;
    if (!whole_pieces) return launch_decode_frames_partial(nch, in, out, frames, n_frames, mismatches, scratch, stream);
    // few wavefronts (the short last frames of a batch): one per workgroup, so that they spread over the compute units
    const unsigned waves = (n_frames + (64u / (unsigned)nch) - 1) / (64u / (unsigned)nch);
    if (waves < 1024u) return launch_decode_frames_r<4, 1, false>(nch, in, out, frames, n_frames, mismatches, scratch, stream);
    return launch_decode_frames_r<4, 4, false>(nch, in, out, frames, n_frames, mismatches, scratch, stream);
}
#endif // part 1
#if !defined(QOA_DECODE_PART) || QOA_DECODE_PART == 4
// (part 4, qoa_decode_w.hip: the PARTIAL variants -- frames whose PCM is no whole number of 16-byte pieces)
hipError_t launch_decode_frames_partial(int nch, const uint8_t *in, int16_t *out, const DecChain *frames, unsigned n_frames,
                                        unsigned *mismatches, uint8_t *scratch, hipStream_t stream) {
    const unsigned waves = (n_frames + (64u / (unsigned)nch) - 1) / (64u / (unsigned)nch);
    if (waves < 1024u) return launch_decode_frames_r<4, 1, true>(nch, in, out, frames, n_frames, mismatches, scratch, stream);
    return launch_decode_frames_r<4, 4, true>(nch, in, out, frames, n_frames, mismatches, scratch, stream);
}
#endif // part 4
} // namespace qoa
