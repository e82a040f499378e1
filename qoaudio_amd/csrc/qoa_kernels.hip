// qoa_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the QOA hot path.
//
// K1  qoa_encode_kernel : the 16-scalefactor brute-force slice search with the 4-tap sign-sign LMS
//     recurrence (reference: encode_slice src/lib.rs:495-596, encode_frame_to_buf :441-493,
//     QoaLms :797-829, qoa_div :613-617).
// (K2 / K2u, the slice decoder, live in qoa_decode.hip.)
//
// Mapping (K1).  A chain = one (clip, channel) is strictly serial over its whole length because the
// winner's end state of slice k is the entry state of slice k+1 (src/lib.rs:482) and the encoder
// never re-seeds it.  What is parallel inside a chain is the 16 scalefactor candidates of a slice:
// one wavefront lane per (chain, scalefactor), 16 lanes = one DPP row per chain, CPW = 1, 2 or 4
// chains per 64-wide wavefront.  Every lane runs the 20-sample recurrence for its own scalefactor
// out of registers (its reciprocal and its four dequant magnitudes are lane constants, so the
// quantise/dequantise "tables" are two packed VGPRs read with v_perm_b32 -- an LDS lookup would put
// ~64 cycles of latency on the serial path 20x per slice).  The winner is the lexicographic minimum
// of (total rank, first-sample rank, scalefactor) -- the closed form of the reference's rank-sorted
// order with its two early exits -- found with two 32-bit row_ror DPP min-reductions, and its end
// state reaches all 16 lanes through the chain's LDS slots (ds_bpermute on the per-slice fall-back paths).
//
// All arithmetic reproduces the reference's 32-bit wrapping integers exactly.  The kernel is bound by VALU
// instruction ISSUE (one wave per SIMD, ~4 cycles per instruction whatever it is; a branch costs 20-27), so everything is
// about the instruction count of the serial slice step -- 22 instructions in the cheap-quotient speculative form that
// ordinary audio runs on, 26 / 27 / 35 in the exact-quotient / penalty / full forms on i16-packed weights and history where
// those are provably exact, 41 on 32-bit weights otherwise -- and about keeping everything else off the compute
// wavefront: 16 full slices at a time run inside one generated asm loop with one branch per slice (qoa_slice_asm.inc,
// tools/gen_slice_asm.py), and launches that leave SIMDs idle give every compute wavefront a helper wavefront that does
// the memory work of the block top.  See DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "qoa_format.h"
#include "qoa_kernels.h"
#ifndef QOA_SLICE_ASM_INC // (tools/k1lab builds timing variants of the generated texts)
#define QOA_SLICE_ASM_INC "qoa_slice_asm.inc"
#endif
#include QOA_SLICE_ASM_INC

#include "qoa_device.h"

namespace qoa {

struct CandResult {
    Lms lms;                 // end state of this candidate
    unsigned long long e;    // sum of err^2
    unsigned pen;            // sum of penalty^2 (fits u32: 20 * 5888^2)
    unsigned aerr0;          // |err| of the first sample: orders the first-sample ranks (the first-sample
                             // penalty is the same for all 16 candidates, they share the entry weights)
    unsigned s_hi, s_lo;     // packed 64-bit slice, tail already left-aligned
};

// One candidate-sample step (src/lib.rs:566-584) as ONE hand-written instruction block:
// 41 VALU instructions.  The kernel is issue-bound at one wave per SIMD (~4.4 cycles per instruction whatever
// it is, tools/ubench), so instruction COUNT is the cost; written in C++ the same step compiles to 43-44
// instructions plus an s_nop after every one of its 6-7 inline-asm helpers (hipcc pads each asm boundary).
//
//   predict   ((w0h0 + w1h1) + (w2h2 + w3h3)) >> 13, wrapping                      src/lib.rs:811-813
//   penalty   max(0, ((w0w0 + w1w1 + w2w2 + w3w3) >> 18) - 0x8ff), squared, summed    src/lib.rs:814-817
//   qoa_div   n = (res * recip + 2^15) >> 16; scaled = n + sgn(res) - sgn(n)       src/lib.rs:613-617
//             computed as med3(n, n + 2a, a) with a = sgn(res): for a = +1 that is n (n > 0), 1 (n = 0) and
//             n + 2 (n < 0, which only the i32 wrap of res * recip at sf 0/1 produces), mirrored for a = -1,
//             and 0 for a = 0 (then n = 0) -- the same value in every case, one instruction fewer
//   quantiser c = clamp(scaled,-8,8); level = min(|c|>>1,3); code = 2*level + (c<0); dequant = +-mag[level]
//             (closed form of QOA_QUANT_TAB / QOA_DEQUANT_TAB, verified exhaustively in tests/test_capi_host.py;
//             the lane's four magnitudes are packed in dq_lo/dq_hi and selected by v_perm_b32)
//   update    w_i += sign(h_i) * (dequant >> 4), s_i = +-1 travels with h_i         src/lib.rs:821-828
//
// Dot products use v_mul_lo_u32 + v_mad_u64_u32 (low dword = full 32-bit wrapping multiply-add; same issue cost
// as the 24-bit forms on gfx950), so weights need no range guard.  v_mad_i32_i24 is used only where the operands
// provably fit 24 bits: residual (|r| < 2^19) x reciprocal (<= 2^16), penalty (<= 5888), +-1 x (dequant >> 4).
// v[124:127] are scratch pairs for the two dot products (a 64-bit asm operand has no sub-register syntax).
struct StepConst {
    int recip;          // this lane's reciprocal
    unsigned dq_lo, dq_hi; // this lane's dequant magnitudes, 2 x 16 bit each
    unsigned selbase;   // 0x0c0c0100: v_perm selector for level 0
    int hi;             // 32767
};

__device__ __forceinline__ void cand_step(int smp, const StepConst &k, Lms &l, int &s0, int &s1, int &s2, int &s3,
                                          unsigned long long &e, unsigned &pen_sum, unsigned &acc, int &err_out) {
    int nw0, nw1, nw2, nw3, recon, ns3, err, t, u, sneg, d;
    // Order: a lone wave issues an instruction that depends on its predecessor ~0.5 cycle later than an
    // independent one (tools/ubench), so the predict and penalty chains are interleaved and the off-path work
    // (code packing, weight update, rank sum) is slotted between the dependent steps of the quantiser.
    asm("v_mul_lo_u32 v124, %[w0], %[h0]\n\t"
        "v_mul_lo_u32 v126, %[w0], %[w0]\n\t"
        "v_mad_u64_u32 v[124:125], vcc, %[w1], %[h1], v[124:125]\n\t"
        "v_mad_u64_u32 v[126:127], vcc, %[w1], %[w1], v[126:127]\n\t"
        "v_mad_u64_u32 v[124:125], vcc, %[w2], %[h2], v[124:125]\n\t"
        "v_mad_u64_u32 v[126:127], vcc, %[w2], %[w2], v[126:127]\n\t"
        "v_mad_u64_u32 v[124:125], vcc, %[w3], %[h3], v[124:125]\n\t"
        "v_mad_u64_u32 v[126:127], vcc, %[w3], %[w3], v[126:127]\n\t"
        "v_ashrrev_i32 v124, 13, v124\n\t"                     // v124 = predicted
        "v_ashrrev_i32 v126, 18, v126\n\t"
        "v_sub_u32 %[t], %[smp], v124\n\t"                     // residual
        "v_max_i32 v126, 0x8ff, v126\n\t"
        "v_mad_i32_i24 %[u], %[t], %[recip], %[rnd]\n\t"
        "v_add_u32 v126, 0xfffff701, v126\n\t"                 // penalty = max(x - 0x8ff, 0)
        "v_med3_i32 %[t], %[t], -1, 1\n\t"                     // sgn(residual)
        "v_ashrrev_i32 %[u], 16, %[u]\n\t"                     // n
        "v_mad_i32_i24 %[pen], v126, v126, %[pen]\n\t"
        "v_lshl_add_u32 %[sneg], %[t], 1, %[u]\n\t"            // n + 2 sgn(residual)
        "v_med3_i32 %[u], %[u], %[sneg], %[t]\n\t"             // scaled
        "v_sub_u32 %[t], 0, %[u]\n\t"
        "v_ashrrev_i32 %[sneg], 31, %[u]\n\t"
        "v_max_i32 %[t], %[u], %[t]\n\t"                       // |scaled|
        "v_min_i32 %[t], 7, %[t]\n\t"
        "v_and_b32 %[t], 6, %[t]\n\t"                          // 2*level
        "v_mad_u32_u24 %[d], %[t], %[c101], %[selb]\n\t"
        "v_lshl_add_u32 %[acc], %[acc], 3, %[t]\n\t"
        "v_perm_b32 %[d], %[dqhi], %[dqlo], %[d]\n\t"          // magnitude
        "v_sub_u32 %[acc], %[acc], %[sneg]\n\t"                // acc = acc<<3 | code
        "v_xor_b32 %[d], %[d], %[sneg]\n\t"
        "v_sub_u32 %[d], %[d], %[sneg]\n\t"                    // dequantised
        "v_add_u32 %[recon], %[d], v124\n\t"
        "v_ashrrev_i32 %[d], 4, %[d]\n\t"                      // delta
        "v_med3_i32 %[recon], %[recon], %[lo], %[hi]\n\t"      // reconstructed
        "v_mad_i32_i24 %[nw0], %[s0], %[d], %[w0]\n\t"
        "v_sub_u32 %[err], %[smp], %[recon]\n\t"
        "v_mad_i32_i24 %[nw1], %[s1], %[d], %[w1]\n\t"
        "v_ashrrev_i32 %[ns3], 31, %[recon]\n\t"
        "v_mad_i32_i24 %[nw2], %[s2], %[d], %[w2]\n\t"
        "v_mad_i64_i32 %[e], vcc, %[err], %[err], %[e]\n\t"    // rank += err^2
        "v_mad_i32_i24 %[nw3], %[s3], %[d], %[w3]\n\t"
        "v_or_b32 %[ns3], 1, %[ns3]"
        : [nw0] "=&v"(nw0), [nw1] "=&v"(nw1), [nw2] "=&v"(nw2), [nw3] "=&v"(nw3), [recon] "=&v"(recon),
          [ns3] "=&v"(ns3), [err] "=&v"(err), [t] "=&v"(t), [u] "=&v"(u), [sneg] "=&v"(sneg), [d] "=&v"(d), [e] "+v"(e),
          [pen] "+v"(pen_sum), [acc] "+v"(acc)
        : [w0] "v"(l.w0), [w1] "v"(l.w1), [w2] "v"(l.w2), [w3] "v"(l.w3), [h0] "v"(l.h0), [h1] "v"(l.h1),
          [h2] "v"(l.h2), [h3] "v"(l.h3), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [smp] "v"(smp),
          [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi), [selb] "v"(k.selbase), [hi] "v"(k.hi),
          [rnd] "s"(1 << 15), [c101] "s"(0x0101), [lo] "s"(-32768)
        : "vcc", "v124", "v125", "v126", "v127");
    l.w0 = nw0; l.w1 = nw1; l.w2 = nw2; l.w3 = nw3;
    l.h0 = l.h1; l.h1 = l.h2; l.h2 = l.h3; l.h3 = recon;
    s0 = s1; s1 = s2; s2 = s3; s3 = ns3;
    err_out = err;
}

// A full 20-sample slice, fully unrolled (the history shift is register renaming).
__device__ __forceinline__ void run_slice_full(const int (&smp)[20], int sf, const StepConst &k, const Lms &entry,
                                               CandResult &r) {
    Lms l = entry;
    int s0, s1, s2, s3;
    sign_mults(l, s0, s1, s2, s3);
    unsigned long long e = 0;
    unsigned pen = 0, acc_a = 0, acc_b = 0;
#pragma unroll
    for (int i = 0; i < 20; i++) {
        int err;
        cand_step(smp[i], k, l, s0, s1, s2, s3, e, pen, i < 10 ? acc_a : acc_b, err);
        if (i == 0) r.aerr0 = uabs(err);
    }
    r.lms = l;
    r.e = e;
    r.pen = pen;
    // slice = sf<<60 | codes 0..9 <<30 | codes 10..19   (src/lib.rs:549,560,584)
    r.s_hi = ((unsigned)sf << 28) | (acc_a >> 2);
    r.s_lo = (acc_a << 30) | acc_b;
}

// The production path for a full slice: all 20 steps in ONE asm statement (text generated by
// tools/gen_slice_asm.py from the same 41-instruction step as cand_step()).  Compared with 20 cand_step() statements
// this drops the s_nop and s_waitcnt hipcc inserts at every asm boundary and the register copies around them;
// the 20 samples are read from the LDS tile inside the block (`tile_addr` = LDS byte address of the slice).
__device__ __forceinline__ void run_slice_asm(unsigned tile_addr, int sf, const StepConst &k, const Lms &entry,
                                              CandResult &r) {
    int w0 = entry.w0, w1 = entry.w1, w2 = entry.w2, w3 = entry.w3;
    int h0 = entry.h0, h1 = entry.h1, h2 = entry.h2, h3 = entry.h3;
    unsigned long long e;
    unsigned pen, acc_a, acc_b, aerr0;
    asm volatile(QOA_SLICE_ASM_TEXT
                 : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [h0] "+v"(h0), [h1] "+v"(h1),
                   [h2] "+v"(h2), [h3] "+v"(h3), [e] "=&v"(e), [pen] "=&v"(pen), [acca] "=&v"(acc_a),
                   [accb] "=&v"(acc_b), [aerr0] "=&v"(aerr0)
                 : [tile] "v"(tile_addr), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi),
                   [selb] "v"(k.selbase), [hi] "v"(k.hi), [rnd] "s"(1 << 15), [c101] "s"(0x0101), [lo] "s"(-32768)
                 : "vcc", "memory", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
                   "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                   "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    r.lms.w0 = w0; r.lms.w1 = w1; r.lms.w2 = w2; r.lms.w3 = w3;
    r.lms.h0 = h0; r.lms.h1 = h1; r.lms.h2 = h2; r.lms.h3 = h3; // 20 % 4 == 0: the ring is back in place
    r.e = e;
    r.pen = pen;
    r.aerr0 = aerr0;
    r.s_hi = ((unsigned)sf << 28) | (acc_a >> 2);
    r.s_lo = (acc_a << 30) | acc_b;
}

// ------------------------------------------------------------------------------------------------
// Packed fast path.  In practice every weight fits i16 (the penalty term exists to keep them there), and
// the history always does (reconstructed samples are clamped).  Carrying them as i16 pairs turns the two
// 4-term dot products into 2 + 2 v_dot2_i32_i16 and the four weight updates into 2 v_pk_mad_i16: 35
// instructions per step instead of 41 (QOA_SLICE_PK_ASM_TEXT, tools/gen_slice_asm.py).
//
// Exactness.  With all |w_i| <= 32767 the products are exact and v_dot2_i32_i16 accumulates them modulo 2^32,
// which is the reference's wrapping i32 sum (src/lib.rs:811-817).  The penalty is computed as
// max(0, (S - (0x8ff << 18)) >> 18) with S = sum w_i^2; that equals max(0, (S >> 18) - 0x8ff) while S < 2^31.
// Invariant: S < kPkEnergyLimit = 3875 << 18 at a step  =>  every |w_i| <= 31871  =>  after the update
// (|delta| <= 14336 >> 4 = 896) every |w_i| <= 32767 and S <= 1.134e9 < 2^31, so the next step computes its S
// and its penalty exactly, and if that S is no longer below the limit its penalty is >= 3875 - 0x8ff = 1572.
// Hence: "penalty sum of a candidate < 1572^2" proves that all 20 steps of that candidate were exact, and that
// its end state again satisfies what the next slice's first step needs.  A slice in which any candidate of the
// wave fails the test is evaluated again by the 32-bit path from the same entry state (it never happens on
// real audio or on the benchmark signals: qoa_encode_plan_stats() counts it, tests/test_gpu_parity.py checks the
// reference's fixture and bench.py reports the workload's count).  After a 32-bit slice the state is
// re-admitted by pk_try_pack(), which checks the ranges and the energy directly.
// On top of this form sits the speculative one (run_block_spec): the same packed steps without the penalty arithmetic,
// proven slice by slice from a sampled energy maximum; this full form is what a speculation miss falls back to.
// ------------------------------------------------------------------------------------------------
struct PkLms {
    unsigned w01, w23; // (w0, w1), (w2, w3): lower index in the low half
    unsigned pa, pb;   // history (h0, h1), (h2, h3)
};
constexpr unsigned kPkPenLimit = 1572u * 1572u;
constexpr long long kPkEnergyLimit = 3875ll << 18;

struct PkResult {
    PkLms lms;
    unsigned long long e;
    unsigned pen, aerr0, s_hi, s_lo;
};

__device__ __forceinline__ void pk_unpack(const PkLms &p, Lms &l) {
    l.w0 = (int)(p.w01 << 16) >> 16; l.w1 = (int)p.w01 >> 16;
    l.w2 = (int)(p.w23 << 16) >> 16; l.w3 = (int)p.w23 >> 16;
    l.h0 = (int)(p.pa << 16) >> 16; l.h1 = (int)p.pa >> 16;
    l.h2 = (int)(p.pb << 16) >> 16; l.h3 = (int)p.pb >> 16;
}

// Can this state enter the packed path?  (Same answer in the 16 lanes of a chain: they hold the same state.)
__device__ __forceinline__ bool pk_try_pack(const Lms &l, PkLms &p) {
    auto fits = [](int v) { return (unsigned)(v + 32768) < 65536u; };
    auto pack = [](int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); };
    const bool in_range = fits(l.w0) && fits(l.w1) && fits(l.w2) && fits(l.w3) && fits(l.h0) && fits(l.h1) &&
                          fits(l.h2) && fits(l.h3);
    const int c0 = med3(l.w0, -32768, 32767), c1 = med3(l.w1, -32768, 32767), c2 = med3(l.w2, -32768, 32767),
              c3 = med3(l.w3, -32768, 32767); // clamped: the 64-bit sum below cannot overflow
    const long long energy = (long long)c0 * c0 + (long long)c1 * c1 + (long long)c2 * c2 + (long long)c3 * c3;
    p.w01 = pack(l.w0, l.w1); p.w23 = pack(l.w2, l.w3);
    p.pa = pack(l.h0, l.h1); p.pb = pack(l.h2, l.h3);
    return in_range && energy < kPkEnergyLimit;
}

__device__ __forceinline__ void run_slice_pk(unsigned tile_addr, int sf, const StepConst &k, const PkLms &entry,
                                             PkResult &r) {
    unsigned w01 = entry.w01, w23 = entry.w23, pa = entry.pa, pb = entry.pb;
    unsigned long long e;
    unsigned pen, acc_a, acc_b, aerr0;
    asm volatile(QOA_SLICE_PK_ASM_TEXT
                 : [w01] "+v"(w01), [w23] "+v"(w23), [pa] "+v"(pa), [pb] "+v"(pb), [e] "=&v"(e), [pen] "=&v"(pen),
                   [acca] "=&v"(acc_a), [accb] "=&v"(acc_b), [aerr0] "=&v"(aerr0)
                 : [tile] "v"(tile_addr), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi),
                   [selb] "v"(k.selbase), [hi] "v"(k.hi), [rnd] "s"(1 << 15), [c101] "s"(0x0101), [lo] "s"(-32768),
                   [negc] "s"(-(0x8ff << 18))
                 : "vcc", "memory", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
                   "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
                   "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    r.lms.w01 = w01; r.lms.w23 = w23; r.lms.pa = pa; r.lms.pb = pb;
    r.e = e;
    r.pen = pen;
    r.aerr0 = aerr0;
    r.s_hi = ((unsigned)sf << 28) | (acc_a >> 2);
    r.s_lo = (acc_a << 30) | acc_b;
}

// `n` consecutive full slices of a block on the packed path in ONE statement (QOA_BLOCK_PK_ASM_TEXT): candidate steps,
// winner election, exchange of the winner's state through the chain's 16 LDS slots (`cand` = this lane's slot, slots
// are 32 bytes apart, a row's 16 slots are 512-byte aligned) and staging of the winner's slice words at `outb + 8 * i`
// (the others write to `dummy`).  Returns how many slices were completed; fewer than n means slice number `done`
// failed the kPkPenLimit test -- `pk` then holds that slice's entry state and the caller takes the 32-bit path.
__device__ __forceinline__ unsigned run_block_pk(unsigned tile_addr, unsigned cand, unsigned outb, unsigned dummy, int sf,
                                                 const StepConst &k, unsigned n, PkLms &pk) {
    register unsigned w01 asm("v84") = pk.w01;
    register unsigned w23 asm("v85") = pk.w23;
    register unsigned pa asm("v86") = pk.pa;
    register unsigned pb asm("v87") = pk.pb;
    unsigned done;
    asm volatile(QOA_BLOCK_PK_ASM_TEXT
                 : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done)
                 : [tile] "v"(tile_addr), [cand] "v"(cand), [outb] "v"(outb), [dummy] "v"(dummy),
                   [sf4] "v"((unsigned)sf << 5), [sfsh] "v"((unsigned)sf << 28), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo),
                   [dqhi] "v"(k.dq_hi), [selb] "v"(k.selbase), [hi] "v"(k.hi), [rnd] "s"(1 << 15), [c101] "s"(0x0101),
                   [lo] "s"(-32768), [negc] "s"(-(0x8ff << 18)), [n] "s"(n), [limit] "s"(kPkPenLimit), [mf0] "s"(0x1e0),
                   [hiw] "s"(0x07060302u) // (the penalties of two steps side by side: pk_step, pair_pen)
                 : "vcc", "scc", "memory", "s95", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v88", "v89", "v90", "v91",
                   "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
                   "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120",
                   "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    pk.w01 = w01; pk.w23 = w23; pk.pa = pa; pk.pb = pb;
    return done;
}

// The speculative form of run_block_pk (QOA_BLOCK_SPEC_ASM_TEXT): steps without any penalty arithmetic (26 instructions,
// 29 at the five steps that sample the weight energy), proven after each slice by two tests on the sampled maximum --
// limq: every candidate stayed inside the i16 window (so all 64 ranks are exact err^2 sums); winq: the WINNER's penalty
// was zero at every step, hence its rank is its true rank while every other candidate's true rank is >= its computed
// one, and the reference's winner is this winner.  Margins for the three unsampled steps are folded into limq / winq
// (spec_limits).  The rank is a saturating 32-bit sum (tools/gen_slice_asm.py: spec_rank_pair): a third test demands
// that the winner's is below 32767^2, where it is provably the true rank.  All tests are folded into ONE per-lane
// comparison at the end of the slice (tools/gen_slice_asm.py: pk_block_spec -- a lone wavefront pays 20-27 cycles per
// branch); a slice that fails it ends the call (done < n) with its entry state restored -- the candidate slots are
// double-buffered, `pp` = byte distance between the two buffers -- and the caller runs it through the next form down.
//
// CHEAP selects QOA_BLOCK_CHEAP_ASM_TEXT, the same loop with the cheap quotient (tools/gen_slice_asm.py: cheap_quotient_note):
// 22 / 25 instructions per step.  The quantiser's level is taken from |n| and its sign from the residual, which is what the
// reference computes unless the wrapped quotient n and the residual have opposite signs (scalefactors 0 and 1 only); such a
// candidate is marked and takes part in the election with the rank it provably costs at least (tile record dword 0,
// marked_bound(); real ranks count from 1 there): if that wins -- the best unmarked rank is not below the bound -- the slice is declined like a
// speculation miss.
template <bool CHEAP, bool COUNT = false>
__device__ __forceinline__ unsigned run_block_spec(unsigned tile_addr, unsigned cand, unsigned pp, unsigned outb, unsigned dummy,
                                                   int sf, const StepConst &k, int limq, int winq, unsigned n, PkLms &pk,
                                                   unsigned *marked_pairs = nullptr, unsigned long long *lab = nullptr) {
    register unsigned w01 asm("v84") = pk.w01;
    register unsigned w23 asm("v85") = pk.w23;
    register unsigned pa asm("v86") = pk.pa;
    register unsigned pb asm("v87") = pk.pb;
    unsigned done, tmp;
#define QOA_SPEC_CLOBBERS                                                                                                  \
    "vcc", "scc", "memory", "s94", "s95", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",  \
        "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",  \
        "v97", "v98", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",          \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", \
        "v127"
#define QOA_SPEC_INPUTS                                                                                                    \
    [tile] "v"(tile_addr), [cand] "v"(cand), [outb] "v"(outb), [dummy] "v"(dummy), [sf4] "v"((unsigned)sf << 5),           \
        [sfsh] "v"((unsigned)sf << 28), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi), [selb] "v"(k.selbase), \
        [hi] "v"(k.hi), [limq] "v"(limq), [winq] "v"(winq), [rnd] "s"(1 << 15), [c101] "s"(0x0101), [lo] "s"(-32768),       \
        [negc] "s"(-(0x8ff << 18)), [n] "s"(n), [mf0] "s"(0x1e0), [pp] "s"(pp)
    if constexpr (CHEAP) {
#ifdef QOA_LAB_STAMPS // timing experiment (tools/k1lab/stamps.py): cycles inside the 20 steps / in the rest of the slice loop
        unsigned tb, tt;
        asm volatile(QOA_BLOCK_CHEAP_ASM_TEXT
                     : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp), [tb] "=&s"(tb), [tt] "=&s"(tt)
                     : QOA_SPEC_INPUTS
                     : QOA_SPEC_CLOBBERS, "v99", "s86", "s87", "s88", "s89", "s90", "s91", "s92");
        if (lab) { lab[0] += tb; lab[1] += tt; }
        if (marked_pairs) *marked_pairs = 0;
#else
        if constexpr (COUNT) { // plans with counters: the text that also counts this lane's marked (candidate, slice) pairs
            register unsigned marks asm("v99");
            asm volatile(QOA_BLOCK_CHEAP_COUNT_ASM_TEXT
                         : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp), "=v"(marks)
                         : QOA_SPEC_INPUTS
                         : QOA_SPEC_CLOBBERS);
            if (marked_pairs) *marked_pairs = marks;
        } else {
            asm volatile(QOA_BLOCK_CHEAP_ASM_TEXT
                         : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp)
                         : QOA_SPEC_INPUTS
                         : QOA_SPEC_CLOBBERS, "v99");
        }
#endif
    } else {
        asm volatile(QOA_BLOCK_SPEC_ASM_TEXT
                     : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp)
                     : QOA_SPEC_INPUTS
                     : QOA_SPEC_CLOBBERS, "v99");
    }
#undef QOA_SPEC_INPUTS
#undef QOA_SPEC_CLOBBERS
    pk.w01 = w01; pk.w23 = w23; pk.pa = pa; pk.pb = pb;
    return done;
}

// The wide form (QOA_BLOCK_WIDE_ASM_TEXT, tools/gen_slice_asm.py: pk_block(wide=True), round 4): exact-quotient speculative
// steps with the TRUE 64-bit rank (one error term per step instead of the saturating pair sum) -- 28 instructions per step.  For material whose winners' ranks exceed the saturating rank's exact range
// (32767^2) and whose scalefactor-0/1 candidates wrap without a provable bound, but whose winners carry NO penalty:
// full-scale noise (tools/k1lab/leaver_study.c: 92 % of its winners' ranks are beyond 32767^2, a candidate is marked in 48 % of
// its slices with no bound at all, and no winner carries a penalty).  Before, that class ran on the 35-instruction full form.
// The energy tests are those of run_block_spec (a sample at every 4th step, spec_limits' margins); until late in round 4 this form
// sampled at every step (30 per step) on the belief that the margins made noise miss on the other forms -- measured: with the
// margins it declines 516 of white noise's 2,097,152 wavefront-slices and runs 6 % faster.  Declines (done < n, entry state
// restored) when a winner's penalty is not provably zero or a candidate's energy may have left the packed window.
__device__ __forceinline__ unsigned run_block_wide(unsigned tile_addr, unsigned cand, unsigned pp, unsigned outb, unsigned dummy,
                                                   int sf, const StepConst &k, int limq, int winq, unsigned n, PkLms &pk) {
    register unsigned w01 asm("v84") = pk.w01;
    register unsigned w23 asm("v85") = pk.w23;
    register unsigned pa asm("v86") = pk.pa;
    register unsigned pb asm("v87") = pk.pb;
    unsigned done, tmp;
    asm volatile(QOA_BLOCK_WIDE_ASM_TEXT
                 : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp)
                 : [tile] "v"(tile_addr), [cand] "v"(cand), [outb] "v"(outb), [dummy] "v"(dummy), [sf4] "v"((unsigned)sf << 5),
                   [sfsh] "v"((unsigned)sf << 28), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi),
                   [selb] "v"(k.selbase), [hi] "v"(k.hi), [limq] "v"(limq), [winq] "v"(winq), [rnd] "s"(1 << 15),
                   [c101] "s"(0x0101), [lo] "s"(-32768), [negc] "s"(-(0x8ff << 18)), [n] "s"(n), [mf0] "s"(0x1e0), [pp] "s"(pp)
                 : "vcc", "scc", "memory", "s94", "s95", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
                   "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
                   "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
                   "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
                   "v124", "v125", "v126", "v127");
    pk.w01 = w01; pk.w23 = w23; pk.pa = pa; pk.pb = pb;
    return done;
}

// The penalty form (QOA_BLOCK_PEN_ASM_TEXT, tools/gen_slice_asm.py: pk_block(pen=True)): cheap-quotient steps WITH the penalty
// arithmetic on every step; the rank is err^2 + penalty as the full form's, nothing about the penalty is speculated.  What
// the block loop runs where the speculative forms miss (winners that carry a penalty: tonal passages) -- 26 instead of the
// full form's 35 instructions per step (the penalty's square is taken two steps at a time).  Declines (done < n, entry state restored) like run_block_spec<true> (marked
// candidate not provably a loser, rank not provably exact) and like run_block_pk (a candidate's weight energy left the
// packed window: penalty sum >= kPkPenLimit).
__device__ __forceinline__ unsigned run_block_pen(unsigned tile_addr, unsigned cand, unsigned pp, unsigned outb, unsigned dummy,
                                                  int sf, const StepConst &k, unsigned n, PkLms &pk) {
    register unsigned w01 asm("v84") = pk.w01;
    register unsigned w23 asm("v85") = pk.w23;
    register unsigned pa asm("v86") = pk.pa;
    register unsigned pb asm("v87") = pk.pb;
    unsigned done, tmp;
    asm volatile(QOA_BLOCK_PEN_ASM_TEXT
                 : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp)
                 : [tile] "v"(tile_addr), [cand] "v"(cand), [outb] "v"(outb), [dummy] "v"(dummy), [sf4] "v"((unsigned)sf << 5),
                   [sfsh] "v"((unsigned)sf << 28), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi),
                   [selb] "v"(k.selbase), [hi] "v"(k.hi), [limitv] "v"(kPkPenLimit), [rnd] "s"(1 << 15), [c101] "s"(0x0101),
                   [lo] "s"(-32768), [negc] "s"(-(0x8ff << 18)), [n] "s"(n), [mf0] "s"(0x1e0), [pp] "s"(pp),
                   [hiw] "s"(0x07060302u) // v_perm selector: the high words of two energies side by side (pk_step, penalty form)
                 : "vcc", "scc", "memory", "s94", "s95", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
                   "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
                   "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
                   "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
                   "v124", "v125", "v126", "v127");
    pk.w01 = w01; pk.w23 = w23; pk.pa = pa; pk.pb = pb;
    return done;
}

// `n` consecutive full slices on the 32-bit step (38.5 instructions: the penalty is squared two steps at a time) in ONE statement
// (QOA_BLOCK_32_ASM_TEXT, tools/gen_slice_asm.py: b32_block, step_pair_pen):
// full i32 weights, so it cannot decline -- the block loop's last resort for wavefronts with a chain whose state, or whose
// losing candidates, leave the packed forms' window slice after slice.  Election and state exchange as in run_block_pk, with
// both halves of the 32-byte slot (history, weights).
__device__ __forceinline__ void run_block_32(unsigned tile_addr, unsigned cand, unsigned outb, unsigned dummy, int sf,
                                             const StepConst &k, unsigned n, Lms &st) {
    register int h0 asm("v84") = st.h0;
    register int h1 asm("v85") = st.h1;
    register int h2 asm("v86") = st.h2;
    register int h3 asm("v87") = st.h3;
    register int w0 asm("v88") = st.w0;
    register int w1 asm("v89") = st.w1;
    register int w2 asm("v90") = st.w2;
    register int w3 asm("v91") = st.w3;
    unsigned done;
    asm volatile(QOA_BLOCK_32_ASM_TEXT
                 : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), [done] "=&s"(done)
                 : [tile] "v"(tile_addr), [cand] "v"(cand), [outb] "v"(outb), [dummy] "v"(dummy), [sf4] "v"((unsigned)sf << 5),
                   [sfsh] "v"((unsigned)sf << 28), [recip] "v"(k.recip), [dqlo] "v"(k.dq_lo), [dqhi] "v"(k.dq_hi),
                   [selb] "v"(k.selbase), [hi] "v"(k.hi), [rnd] "s"(1 << 15), [c101] "s"(0x0101), [lo] "s"(-32768), [n] "s"(n),
                   [mf0] "s"(0x1e0), [hiw] "s"(0x07060302u), [c8ff2] "s"(0x08ff08ffu) // (the pair-wise penalty: step_pair_pen)
                 : "vcc", "scc", "memory", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v92",
                   "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
                   "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",
                   "v122", "v123", "v124", "v125", "v126", "v127");
    (void)done;
    st.h0 = h0; st.h1 = h1; st.h2 = h2; st.h3 = h3;
    st.w0 = w0; st.w1 = w1; st.w2 = w2; st.w3 = w3;
}

// run_block_spec with the quantiser / dequantiser in an LDS table (QOA_BLOCK_SPEC_LDSQ_ASM_TEXT, tools/gen_slice_asm.py:
// pk_step_ldsq): 22 VALU + 1 LDS read per step instead of 30 VALU.  `qtab` = LDS byte address of this lane's table entry for
// scaled = 0 (entries for scaled = -8..8 sit at qtab - 32 .. qtab + 32).  Used when the launch gives every SIMD several
// wavefronts, which hide the read's latency; a lone wavefront would wait for it 20 times per slice.
__device__ __forceinline__ unsigned run_block_spec_ldsq(unsigned tile_addr, unsigned cand, unsigned pp, unsigned outb, unsigned dummy,
                                                        int sf, const StepConst &k, int limq, int winq, unsigned qtab, unsigned n,
                                                        PkLms &pk) {
    register unsigned w01 asm("v84") = pk.w01;
    register unsigned w23 asm("v85") = pk.w23;
    register unsigned pa asm("v86") = pk.pa;
    register unsigned pb asm("v87") = pk.pb;
    unsigned done, tmp;
    asm volatile(QOA_BLOCK_SPEC_LDSQ_ASM_TEXT
                 : "+v"(w01), "+v"(w23), "+v"(pa), "+v"(pb), [done] "=&s"(done), [tmp] "=&s"(tmp)
                 : [tile] "v"(tile_addr), [cand] "v"(cand), [outb] "v"(outb), [dummy] "v"(dummy),
                   [sf4] "v"((unsigned)sf << 5), [sfsh] "v"((unsigned)sf << 28), [recip] "v"(k.recip), [qtab] "v"(qtab),
                   [hi] "v"(k.hi), [limq] "v"(limq), [winq] "v"(winq),
                   [rnd] "s"(1 << 15), [lo] "s"(-32768), [negc] "s"(-(0x8ff << 18)), [n] "s"(n),
                   [mf0] "s"(0x1e0), [pp] "s"(pp)
                 : "vcc", "scc", "memory", "s94", "s95", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",
                   "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
                   "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
                   "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
                   "v124", "v125", "v126", "v127");
    pk.w01 = w01; pk.w23 = w23; pk.pa = pa; pk.pb = pb;
    return done;
}

// Thresholds of the speculative form for scalefactor sf, in units of (weight energy - (0x8ff << 18)) as the steps sample
// it.  One step changes the energy S = sum w^2 by at most 2 d sum|w| + 4 d^2 <= 4 d sqrt(S) + 4 d^2 with
// d = ceil(mag3(sf) / 16) the largest weight step (sum|w| <= 2 sqrt(S)).  Three steps go unsampled after each sample:
//   limq: S + 3 G < kPkEnergyLimit, G = 4 d 31872 + 4 d^2  -> S stays below the limit (sqrt < 31872) until the next sample,
//         which is run_slice_pk's exactness invariant;
//   winq: S + 3 G' < 0x900 << 18, G' = 4 d 24576 + 4 d^2    -> the penalty (S >> 18) - 0x8ff is <= 0 at every step.
__device__ __forceinline__ void spec_limits(int sf, int &limq, int &winq) {
    const long long d = (c_mags[sf][3] + 15) >> 4;
    const long long c = 0x8ffll << 18;
    limq = (int)(kPkEnergyLimit - 3 * (4 * d * 31872 + 4 * d * d) - c);
    winq = (int)((0x900ll << 18) - 3 * (4 * d * 24576 + 4 * d * d) - c);
}

// Any slice length 1..20 (the last slice of a clip).  Rolled loop; the sample window is shifted through
// registers so nothing is runtime-indexed (a runtime index would put the array in scratch).  Codes are
// placed at their final bit positions, which is the reference's tail shift (src/lib.rs:485-487) for free.
__device__ __forceinline__ void run_slice_tail(const int (&smp)[20], int len, int sf, const StepConst &k,
                                               const Lms &entry, CandResult &r) {
    Lms l = entry;
    int s0, s1, s2, s3;
    sign_mults(l, s0, s1, s2, s3);
    unsigned long long e = 0, slice = (unsigned long long)sf << 60;
    unsigned pen = 0;
    int t[20];
#pragma unroll
    for (int j = 0; j < 20; j++) t[j] = smp[j];
    r.aerr0 = 0;
#pragma unroll 1
    for (int i = 0; i < len; i++) {
        int err;
        unsigned code = 0;
        cand_step(t[0], k, l, s0, s1, s2, s3, e, pen, code, err);
#pragma unroll
        for (int j = 0; j < 19; j++) t[j] = t[j + 1];
        if (i == 0) r.aerr0 = uabs(err);
        slice |= (unsigned long long)code << (57 - 3 * i);
    }
    r.lms = l;
    r.e = e;
    r.pen = pen;
    r.s_hi = (unsigned)(slice >> 32);
    r.s_lo = (unsigned)slice;
}

// min over the 16 lanes of a DPP row, result in every lane: 4 fused v_min_u32_dpp (row_ror 8,4,2,1).
// The s_nop 1 in front of each is the VALU-write -> DPP-read hazard (2 wait states), which hipcc does not
// insert inside an asm statement.
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// Winner of a slice = argmin (total rank, first-sample rank, sf): the closed form of the reference's
// rank-sorted candidate order with its two early exits (src/lib.rs:516-593).  total < 2^37; first-sample
// ranks order like |err0| (16 bits) -- two 32-bit keys: tot>>5, then tot[4:0] : |err0| : sf.
__device__ __forceinline__ unsigned pick_winner(unsigned long long e, unsigned pen, unsigned aerr0, int sf) {
    const unsigned long long tot = e + pen;
    const unsigned t_lo = (unsigned)tot, t_hi = (unsigned)(tot >> 32);
    const unsigned k1 = __builtin_amdgcn_alignbit(t_hi, t_lo, 5);
    unsigned k2 = ((t_lo & 31u) << 20) | (aerr0 << 4) | (unsigned)sf;
    const unsigned m1 = row_min_u32(k1);
    k2 = (k1 == m1) ? k2 : 0xffffffffu;
    return row_min_u32(k2) & 15u;
}


// ------------------------------------------------------------------------------------------------
// K1: encode
//
// Memory schedule.  All global-memory traffic of a chain happens once per BLOCK of 16 slices
// (320 sample-times), in straight-line code at the block top (run by the helper wavefront of a HELPER workgroup, by the
// compute wavefront itself otherwise): lane j of the chain's 16-lane row loads the 20 samples of slice 16*(blk+1)+j for
// the NEXT block (20 x 16-bit loads off one address with immediate offsets), parks the 10 pairs it loaded one block
// earlier -- behind the slice's marked-candidate bound, a 48-byte record -- in a wave-private LDS tile [chain][slice j]
// (3 x ds_write_b128), and stores the 8-byte slice it kept for row 16*(blk-1)+j of the PREVIOUS block.
// Inside a block there is no vector-memory instruction at all, so no s_waitcnt vmcnt can land on the
// serial slice step; the 16 candidate lanes of a chain fetch a slice's record with 3 broadcast
// ds_read_b128 (the speculative forms work on the pairs as they are, the others widen them in place).
// A frame is exactly 16 blocks, so frame
// headers / LMS dumps are written at block tops too.
// ------------------------------------------------------------------------------------------------
constexpr int kBlockSlices = 16;
constexpr unsigned kHelperMaxWaves = 512; // launches up to this many compute wavefronts get helper wavefronts (1024 SIMDs)
constexpr unsigned kSpecCooldownBlocks = 4; // blocks on the full step form after a speculation miss (misses cluster)
#ifndef QOA_PK_COOLDOWN_BLOCKS // (tools/k1lab: the packed forms' back-off after a candidate left the i16 window)
#define QOA_PK_COOLDOWN_BLOCKS 4
#endif
#ifndef QOA_PK_STREAK_MAX
#define QOA_PK_STREAK_MAX 6
#endif
constexpr unsigned kPkCooldownBlocks = QOA_PK_COOLDOWN_BLOCKS, kPkStreakMax = QOA_PK_STREAK_MAX;
#ifndef QOA_SPEC_COOLDOWN_BLOCKS // (tools/k1lab: the speculative forms' back-off after a miss -- a winner that carries a penalty)
#define QOA_SPEC_COOLDOWN_BLOCKS 4
#endif
#ifndef QOA_SPEC_STREAK_MAX
#define QOA_SPEC_STREAK_MAX 6
#endif
constexpr unsigned kSpecMissCooldownBlocks = QOA_SPEC_COOLDOWN_BLOCKS, kSpecStreakMax = QOA_SPEC_STREAK_MAX;
constexpr unsigned kCheapCooldownBlocks = 4; // blocks on the exact-quotient form once the cheap-quotient form declined slices
                                             // in two consecutive blocks (or twice in one)
// A slice's tile record is 12 dwords (three ds_read_b128): [0] marked_bound() of the slice, [1] unused, [2..11] its 20
// samples as 10 i16 pairs.  One record of slack: the block loops prefetch one slice past their last.  (17 * 12 = 204 dwords:
// the 4 chain tiles of a wave start 12 banks apart.)
constexpr int kSliceDwords = 12;
constexpr int kTileInts = (kBlockSlices + 1) * kSliceDwords;

// Where lanes whose slice is not wholly inside the clip point their prefetch (20 samples x 8 channels of zeros).
static __device__ unsigned short g_prefetch_pad[20 * 8]; // (zero-initialised; not const: a constant-address-space pointer would turn the loads into flat ones)

// Prefetch of one slice per lane, a block ahead: 20 loads off ONE address (sample t of this channel is p[t * NCH]: the
// offsets are immediates), nothing per sample but the load.  Round 2 clamped every sample's index to the clip's last
// sample -- ~10 VALU per sample, 200 per lane and block, 2 % of the kernel -- for the sake of the one partial slice a clip
// can end with.  Now a lane whose slice is not wholly inside the clip prefetches a pad of zeros instead and
// load_partial_slice() fetches that slice when its block begins (once per chain).
// raw[i] = sample i of the slice (zero-extended); pair_block_lane() makes the i16 pairs of them where they are consumed.
template <int NCH>
__device__ __forceinline__ void load_block_lane(const int16_t *__restrict__ p, unsigned slice_idx, unsigned samples,
                                                unsigned (&raw)[20]) {
    const unsigned long long base = (unsigned long long)slice_idx * 20u;
    const unsigned short *q = base + 20u <= samples ? (const unsigned short *)p + base * NCH : g_prefetch_pad;
#pragma unroll
    for (int i = 0; i < 20; i++) raw[i] = q[i * NCH];
}

// v[k] = samples (2k, 2k+1) of the slice as an i16 pair, the earlier one in the low half.  The empty asm pins the pairing
// -- and with it the wait for the loads -- to this place: left to itself hipcc hoists it to the first join behind the
// loads, which for a block that takes the ladder is right behind them (their whole latency exposed, +4 % on noise).
__device__ __forceinline__ void pair_block_lane(unsigned (&raw)[20], unsigned (&v)[10]) {
#pragma unroll
    for (int k = 0; k < 10; k++) {
        asm volatile("" : "+v"(raw[2 * k]), "+v"(raw[2 * k + 1]));
        v[k] = raw[2 * k] | (raw[2 * k + 1] << 16);
    }
}

// The partial slice a clip may end with (1..19 samples): indices past the clip end are clamped to its last sample (the
// values are never used: the tail path stops at the slice length).
template <int NCH>
__device__ __forceinline__ void load_partial_slice(const int16_t *__restrict__ p, unsigned slice_idx, unsigned samples,
                                                unsigned (&v)[10]) {
    const unsigned long long base = (unsigned long long)slice_idx * 20u;
    const unsigned long long last = samples - 1u;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const unsigned lo = (unsigned short)p[min(base + (unsigned)(2 * k), last) * NCH];
        const unsigned hi = (unsigned short)p[min(base + (unsigned)(2 * k + 1), last) * NCH];
        v[k] = lo | (hi << 16);
    }
}

// What a candidate of the slice costs at least if its quotient wraps to the opposite sign (cheap-quotient form, see
// run_block_spec): at that step |residual| >= 32768 and |dequantised| <= 49, so |error| >= min(32719, 32767 - max|sample|).
// Returns the square: the rank a marked candidate takes into the election (where real ranks count from 1), so that it wins --
// and the slice is abandoned -- exactly when the best unmarked rank is not below the bound (pk_block_spec in the generator).
__device__ __forceinline__ unsigned marked_bound(const unsigned (&pairs)[10]) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 mx = __builtin_bit_cast(s16x2, pairs[0]), mn = mx;
#pragma unroll
    for (int k = 1; k < 10; k++) {
        const s16x2 v = __builtin_bit_cast(s16x2, pairs[k]);
        mx = __builtin_elementwise_max(mx, v);
        mn = __builtin_elementwise_min(mn, v);
    }
    const int hi = max((int)mx.x, (int)mx.y), lo = min((int)mn.x, (int)mn.y);
    const int smax = max(hi, -lo);
    const int lb = max(min(32719, 32767 - smax), 0);
    return (unsigned)(lb * lb);
}

// the slice at `src` (tile layout: bound, unused, 10 pairs) widened to 20 ints
__device__ __forceinline__ void unpack_tile_slice(const int *src, int (&smp)[20]) {
    const int4 a = ((const int4 *)src)[0], b = ((const int4 *)src)[1], c = ((const int4 *)src)[2];
    const int pk[10] = {a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int k = 0; k < 10; k++) {
        smp[2 * k] = (int)(short)pk[k];
        smp[2 * k + 1] = pk[k] >> 16;
    }
}

// The two wavefronts of a HELPER workgroup synchronise once per block: LDS traffic done, then the barrier.
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// HELPER = true (launches that leave SIMDs idle: config 3 has wavefronts for half of them): a workgroup is ONE compute
// wavefront plus a helper wavefront on another SIMD of the CU that does everything the block top used to do -- prefetch the
// next block's samples, compute the marked bounds, fill the (now double-buffered) LDS tile, flush the previous block's slice
// words, write frame headers from the LMS state the compute wavefront parks in LDS at frame starts.  The compute wavefront
// touches no vector memory in its loop and pays one s_barrier per block instead of ~1,200 cycles of block top (3.2 % of the
// kernel: tools/k1lab, QOA_LAB_NO_BLOCKTOP on the periodic class).  Barrier b separates the compute wavefront's blocks
// b - 1 and b: before it the helper has filled tile[b & 1] and flushed out_tile[b & 1] (block b - 2's words); after it the
// helper flushes out_tile[(b - 1) & 1], fills tile[(b + 1) & 1] and writes frame b / 16's header.
//
// REC = true (plans created with QOA_PLAN_WAVE_STATS): the same kernel with eight wave-uniform counters -- wavefront-slices
// per step form, counted where the ladder hands blocks out -- and a cycle count; lane 0 of every compute wavefront writes
// them to wave_rec[wave * kWaveRecWords ..] once, at exit.  No atomics, nothing on the slice path; a separate instantiation,
// so the kernel every other plan runs is the same code as before (qoa_kernels.h: kWaveRec*).
template <int CPW, int NCH, bool HELPER = false, bool REC = false>
__global__ void __launch_bounds__(256) qoa_encode_kernel(const int16_t *__restrict__ pcm, uint8_t *__restrict__ out,
                                                         const EncChain *__restrict__ chains, unsigned n_chains,
                                                         int32_t *__restrict__ lms_state, unsigned flags,
                                                         unsigned long long *__restrict__ stats, unsigned *__restrict__ wave_rec) {
    // sample tiles [wave][chain][kTileInts]
    __shared__ __attribute__((aligned(16))) int s_smp[4 * CPW * kTileInts];
    // candidate end states of the slice in flight; two buffers: the speculative loop alternates between them so that
    // a slice's entry state (the previous winner's slot) survives until the slice is known to be final.  It switches
    // buffers with one v_xor of the slot address, so the pair must be aligned to its own size (and a row's 16 slots to
    // 512 bytes for the winner-address trick).
    // A slot is 32 bytes: the packed state (weights, the history pairs (h0,h1) and (h2,h3)) and, in the speculative forms,
    // what the next slice's first steps need besides ((h1,h2), the unclamped last reconstruction, two sign pairs).
    constexpr unsigned kCandBufBytes = 4u * CPW * 16u * 2u * sizeof(uint4);
    static_assert((kCandBufBytes & (kCandBufBytes - 1)) == 0, "buffer size must be a power of two");
    __shared__ __attribute__((aligned(2 * kCandBufBytes))) uint4 s_cand[2][4][CPW][16][2];
    // [0,16): the block's winning slices, [16,32): per-lane dummies; 48 entries = 384 B per chain so that the two
    // chains a 64-lane ds_write_b64 serves per pass fall on disjoint banks
    __shared__ __attribute__((aligned(16))) uint2 s_out[4][CPW][48];
    // quantiser + dequantiser table of the LDS-table step form (flags bit 4): [sf][scaled + 8] = dequant & 0xffff | code << 29
    // with code = QOA_QUANT_TAB[scaled + 8] (src/lib.rs:22), dequant = QOA_DEQUANT_TAB[sf][code] (:847-864)
    __shared__ unsigned s_qtab[16][17];
    __shared__ int s_hdr[CPW][8]; // HELPER: a chain's LMS state at frame entry, parked for the helper's frame header
    if (flags & 16u) {
        for (unsigned t = threadIdx.x; t < 16 * 17; t += blockDim.x) {
            const unsigned tsf = t / 17, idx = t % 17;
            const int c = (int)idx - 8, ac = c < 0 ? -c : c;
            const unsigned level = (unsigned)(ac > 7 ? 7 : ac) >> 1, code = 2 * level + (c < 0 ? 1u : 0u);
            const int deq = c < 0 ? -c_mags[tsf][level] : c_mags[tsf][level];
            s_qtab[tsf][idx] = ((unsigned)deq & 0xffffu) | (code << 29);
        }
        __syncthreads();
    }
    const unsigned lane = threadIdx.x & 63u;
    const unsigned grp = lane >> 4;
    if (grp >= (unsigned)CPW) return; // lanes beyond the packed chains idle
    const unsigned wave_in_block = threadIdx.x >> 6;
    const bool is_helper = HELPER && wave_in_block == 1;    // (both wavefronts of a HELPER workgroup serve the same chains)
    // should a compute wavefront ever share its SIMD with somebody's helper, it issues first
    if (HELPER && !is_helper) __builtin_amdgcn_s_setprio(3);
    const unsigned slot = HELPER ? 0u : wave_in_block;      // the LDS slot set of this wavefront's chains
    const unsigned wave = HELPER ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const unsigned chain_id = wave * CPW + grp;
    if (chain_id >= n_chains) return;
#ifdef QOA_LAB_HWID // placement experiment (tools/k1lab/small_launch.py): where did this wavefront land?
    if (stats && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        const unsigned w = wave * 2u + (is_helper ? 1u : 0u);
        if (w < 4096u) stats[10 + w] = ((unsigned long long)(xcc & 15u) << 32) | hwid | (1ull << 63);
    }
    const unsigned long long hwid_t0 = __builtin_readcyclecounter();
    // per compute wavefront: wavefront-slices ATTEMPTED per form (a declined slice counts on the form that declined it too)
    // 0 cheap, 1 exact-quotient speculative, 2 penalty, 3 wide, 4 full packed, 5 32-bit block, 6 slice loop, 7 blocks
#define QOA_LAB_FORM(k, n) do { if (stats && lane == 0 && wave < 2048u) stats[10 + 4096 + 2048 + wave * 8u + (k)] += (unsigned long long)(n); } while (0)
#else
    // REC: wavefront-slices ATTEMPTED per form (a declined slice counts on the form that declined it too):
    // 0 cheap-quotient, 1 exact-quotient speculative (or the LDS-table form), 2 penalty, 3 wide, 4 full packed, 5 32-bit block
    // statement, 6 one-slice statements (tail slices, ragged blocks), 7 blocks
    unsigned rec_cnt[kWaveRecForms] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)rec_cnt;
    const unsigned long long rec_t0 = REC ? __builtin_readcyclecounter() : 0ull;
#define QOA_LAB_FORM(k, n) do { if constexpr (REC) rec_cnt[k] += (unsigned)__builtin_amdgcn_readfirstlane((int)(n)); } while (0)
#endif
    const int sf = (int)(lane & 15u);
    const unsigned row_base = lane & ~15u;
    // HELPER: tiles are double-buffered (block b uses buffer b & 1; the second buffer is the slot set of "wavefront 1")
    int *const tile_buf[2] = {s_smp + (slot * CPW + grp) * kTileInts, s_smp + ((HELPER ? 1u : slot) * CPW + grp) * kTileInts};
    uint2 *const out_buf[2] = {s_out[slot][grp], s_out[HELPER ? 1u : slot][grp]};
    int *tile = tile_buf[0];
    uint2 *out_tile = out_buf[0];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const unsigned cand_addr = (unsigned)(size_t)(lds_ptr)&s_cand[0][slot][grp][sf][0];
    unsigned outb_addr = (unsigned)(size_t)(lds_ptr)out_tile;

    const EncChain ch = chains[chain_id];
    if (ch.samples == 0) return; // a clip that has no frames in this time slice (host pipeline): state untouched
    const int16_t *p = pcm + ch.pcm_off;
    const unsigned samples = ch.samples;
    const unsigned nslices = (samples + 19u) / 20u;
    const unsigned nblocks = (nslices + kBlockSlices - 1) / kBlockSlices;
    StepConst kc;
    kc.recip = c_recip[sf];
    kc.dq_lo = (unsigned)c_mags[sf][0] | ((unsigned)c_mags[sf][1] << 16);
    kc.dq_hi = (unsigned)c_mags[sf][2] | ((unsigned)c_mags[sf][3] << 16);
    kc.selbase = 0x0c0c0100u;
    kc.hi = 32767;
    const bool force_tail_path = (flags & 1u) != 0;
    const bool allow_packed = (flags & 4u) == 0; // bit 2: 32-bit step only (tests compare the two)
    const bool allow_spec = (flags & 8u) == 0;   // bit 3: no speculative (penalty-free) steps
    const bool lds_tables = (flags & 16u) != 0;  // bit 4: speculative steps look the quantiser up in LDS (full-chip launches)
    const bool allow_cheap = (flags & 32u) == 0; // bit 5: no cheap-quotient steps
    const bool allow_pen = (flags & 64u) == 0 && allow_cheap && allow_spec && !lds_tables; // bit 6: no penalty form
    const bool allow_b32 = (flags & 128u) == 0;  // bit 7: no 32-bit block statement (slices the packed forms decline go one by one)
    const bool allow_wide = (flags & 1024u) == 0 && allow_spec && !lds_tables; // bit 10: no wide form (64-bit rank, energy at every step)
    const unsigned qtab_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)&s_qtab[sf][8];
    int limq, winq;
    spec_limits(sf, limq, winq);
    unsigned spec_cooldown = 0;  // blocks left on the full form after a speculation miss (wave-uniform)
    unsigned cheap_cooldown = 0; // blocks left on the exact-quotient speculative form after the cheap one declined slices
    unsigned last_cheap_decline = 0xfffffff0u; // block of the cheap form's last decline (wave-uniform)
    // Consecutive attempt blocks in which the speculative forms / the penalty form had to hand a slice on (wave-uniform).
    // From the second one in a row the form sits out, 4 blocks at first and twice as long each further time up to 64:
    // material that defeats a form does so for long stretches (tonal passages keep the weights at the penalty threshold for
    // seconds; full-scale noise and clipped material never leave it), an isolated miss costs the rest of its block only.
    unsigned spec_streak = 0, pen_cooldown = 0, pen_streak = 0, pk_cooldown = 0, pk_streak = 0, wide_cooldown = 0, wide_streak = 0;
    // The steady state: no form sitting out, no streak running, every chain of the wave packed -- then a whole block goes to the
    // cheap-quotient statement behind ONE test (a lone wavefront pays ~25 cycles per branch and the ladder below has a dozen:
    // ~700 cycles per block, 1.6 % of the kernel, measured with tools/k1lab/stamps.py).  ok_blocks = the lane's number of whole
    // blocks while steady, 0 otherwise; recomputed whenever the ladder has run.
    const bool steady_flags = allow_packed && allow_spec && allow_cheap && !lds_tables && !force_tail_path;
    const unsigned full_blocks = samples / (20u * kBlockSlices);

#ifdef QOA_LAB_STAMPS // stats[10..13]: wave cycles in the steps / rest of the slice loop / the whole statement / the kernel
    unsigned long long lab_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lab_t = 0;
#define QOA_LAB_MARK(i) { const unsigned long long t_ = __builtin_readcyclecounter(); lab_acc[i] += t_ - lab_t; lab_t = t_; }
    const unsigned long long lab_k0 = __builtin_readcyclecounter();
#endif
    Lms st;
    if (flags & 2u) { // fresh encoder: history 0, weights [0, 0, -(1<<13), 1<<14] (src/lib.rs:346-352)
        st.h0 = st.h1 = st.h2 = st.h3 = 0;
        st.w0 = 0; st.w1 = 0; st.w2 = -(1 << 13); st.w3 = 1 << 14;
    } else { // state carried from the previous call on this stream (src/lib.rs:89, :414)
        const int32_t *s = lms_state + (size_t)chain_id * 8;
        st.h0 = s[0]; st.h1 = s[1]; st.h2 = s[2]; st.h3 = s[3];
        st.w0 = s[4]; st.w1 = s[5]; st.w2 = s[6]; st.w3 = s[7];
    }

    // The chain's state lives in `pk` while `packed` (st is then stale), in `st` otherwise.
    PkLms pk;
    bool packed = pk_try_pack(st, pk) && allow_packed;
    unsigned ok_blocks = steady_flags && packed ? full_blocks : 0u;

    uint8_t *o = out + ch.out_off;
    if (ch.flags & 1u) { // file header "qoaf" + samples, big-endian (src/lib.rs:382-385)
        if (sf == 0 && ch.channel == 0 && (!HELPER || is_helper)) *(uint2 *)o = make_uint2(bswap32(kMagic), bswap32(ch.total_samples));
        o += 8;
    }
    constexpr unsigned kFrameBytes = 8 + 16 * NCH + 8 * kSlicesPerFrame * NCH; // a full frame

    // ---- the pieces of the block top (the classic loop runs them itself, a HELPER workgroup's second wavefront for it) ----
    auto fill_tile = [&](int *t, unsigned b, unsigned (&rw)[20]) { // rw = load_block_lane() of slice 16 b + sf
        const unsigned long long s0 = ((unsigned long long)b * kBlockSlices + (unsigned)sf) * 20u;
        unsigned pre[10];
        pair_block_lane(rw, pre);
        if (s0 < samples && s0 + 20u > samples) load_partial_slice<NCH>(p, b * kBlockSlices + (unsigned)sf, samples, pre);
        int *dst = t + sf * kSliceDwords; // needs the prefetched samples: the only vmcnt wait of the loop
        ((uint4 *)dst)[0] = make_uint4(marked_bound(pre), 0u, pre[0], pre[1]);
        ((uint4 *)dst)[1] = make_uint4(pre[2], pre[3], pre[4], pre[5]);
        ((uint4 *)dst)[2] = make_uint4(pre[6], pre[7], pre[8], pre[9]);
    };
    auto flush_block = [&](const uint2 *ot, unsigned b) { // the winning slice words of block b (lane j: row 16 b + j)
        const unsigned k = b * kBlockSlices + (unsigned)sf;
        if (k < nslices) {
            uint8_t *so = o + (size_t)(k >> 8) * kFrameBytes + 8 + 16 * NCH + ((size_t)(k & 255u) * NCH + ch.channel) * 8;
            const uint2 kept = ot[sf];
            *(uint2 *)so = make_uint2(bswap32(kept.x), bswap32(kept.y));
        }
    };
    auto frame_header = [&](unsigned b, int h0, int h1, int h2, int h3, int w0, int w1, int w2, int w3) {
        // frame header + this channel's LMS state at frame entry (src/lib.rs:448-466); block b = 16 * frame, lane sf == 0
        const unsigned frame = b >> 4;
        uint8_t *fo = o + (size_t)frame * kFrameBytes;
        if (ch.channel == 0) {
            unsigned frame_len = min(samples - frame * (unsigned)kFrameLen, (unsigned)kFrameLen);
            unsigned fslices = (frame_len + 19u) / 20u;
            unsigned fsize = (8u + 16u * NCH + 8u * fslices * NCH) & 0xffffu;
            *(uint2 *)fo = make_uint2(bswap32(ch.hdr_hi), bswap32((frame_len << 16) | fsize));
        }
        uint2 *d = (uint2 *)(fo + 8 + 16 * ch.channel);
        d[0] = make_uint2(bswap32(((unsigned)h0 << 16) | ((unsigned)h1 & 0xffffu)), bswap32(((unsigned)h2 << 16) | ((unsigned)h3 & 0xffffu)));
        d[1] = make_uint2(bswap32(((unsigned)w0 << 16) | ((unsigned)w1 & 0xffffu)), bswap32(((unsigned)w2 << 16) | ((unsigned)w3 & 0xffffu)));
    };

    unsigned raw[20];                  // lane j: samples of slice 16*(blk+1)+j, in flight during block blk
    if constexpr (HELPER) {
        if (is_helper) {
            load_block_lane<NCH>(p, (unsigned)sf, samples, raw);
            fill_tile(tile_buf[0], 0, raw);
            for (unsigned b = 0; __builtin_amdgcn_ballot_w64(b <= nblocks) != 0ull; b++) { // one barrier more than the longest chain has blocks
                pair_barrier();
                if (b < nblocks && (b & 15u) == 0 && sf == 0) {
                    const int *h = s_hdr[grp];
                    frame_header(b, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
                }
                if (b >= 1 && b <= nblocks) flush_block(out_buf[(b - 1) & 1u], b - 1);
                if (b + 1 < nblocks) {
                    load_block_lane<NCH>(p, (b + 1) * kBlockSlices + (unsigned)sf, samples, raw);
                    fill_tile(tile_buf[(b + 1) & 1u], b + 1, raw);
                }
            }
            return;
        }
    } else {
        // the winning slice of row 16*blk+j waits in out_tile[j] (LDS) for the flush at the next block top
        load_block_lane<NCH>(p, (unsigned)sf, samples, raw);
    }

#ifndef QOA_LAB_STAMPS
#define QOA_LAB_MARK(i)
#else
    lab_t = __builtin_readcyclecounter();
#endif
    for (unsigned blk = 0; blk < nblocks; blk++) {
        // ---------------- block top: every vector-memory instruction of the loop lives here ----------------
        if constexpr (HELPER) {
            if ((blk & 15u) == 0 && sf == 0) { // park the state at frame entry for the helper's frame header
                if (packed) pk_unpack(pk, st);
                int *h = s_hdr[grp];
                h[0] = st.h0; h[1] = st.h1; h[2] = st.h2; h[3] = st.h3;
                h[4] = st.w0; h[5] = st.w1; h[6] = st.w2; h[7] = st.w3;
            }
            pair_barrier();
            QOA_LAB_MARK(3) // (lab) [3] loop back edge + the wait at the pair barrier
            tile = tile_buf[blk & 1u];
            out_tile = out_buf[blk & 1u];
            outb_addr = (unsigned)(size_t)(lds_ptr)out_tile;
        } else {
#ifdef QOA_LAB_NO_BLOCKTOP // timing experiment: what would the kernel cost if another wavefront did the block top?  (wrong bytes)
            if (blk == 0)
#endif
            fill_tile(tile, blk, raw);
            QOA_LAB_MARK(3) // (lab) [3] loop back edge + tile store
#ifdef QOA_LAB_NO_BLOCKTOP
            if (blk == 1)
#else
            if (blk > 0)
#endif
                flush_block(out_tile, blk - 1); // the previous block's 16 slices (always a full block)
#ifdef QOA_LAB_NO_BLOCKTOP
            if (blk == 0 && sf == 0) {
#else
            if ((blk & 15u) == 0 && sf == 0) {
#endif
                if (packed) pk_unpack(pk, st);
                frame_header(blk, st.h0, st.h1, st.h2, st.h3, st.w0, st.w1, st.w2, st.w3);
            }
            QOA_LAB_MARK(4) // [4] flush + frame header
#ifdef QOA_LAB_NO_BLOCKTOP
            if (blk == 0)
#endif
            load_block_lane<NCH>(p, (blk + 1) * kBlockSlices + (unsigned)sf, samples, raw); // prefetch next block
            QOA_LAB_MARK(5) // [5] prefetch issue
        }

        // ---------------- 16 slices, no vector memory ----------------
        const unsigned k0 = blk * kBlockSlices;
        const unsigned nj = min(nslices - k0, (unsigned)kBlockSlices);
        unsigned j = 0;
        typedef __attribute__((address_space(3))) int *lds_int;
        const unsigned tile0 = (unsigned)(size_t)(lds_int)tile, dummy_addr = outb_addr + 8u * (16u + (unsigned)sf);
        bool cheap_declined = false; // the steady path's statement has already declined slice j
        if (__builtin_amdgcn_ballot_w64(blk >= ok_blocks) == 0ull) { // steady: the whole block on the cheap-quotient statement
            unsigned marked = 0;
#ifdef QOA_LAB_STAMPS
            const unsigned long long lab_c0 = __builtin_readcyclecounter();
            j = run_block_spec<true>(tile0, cand_addr, kCandBufBytes, outb_addr, dummy_addr, sf, kc, limq, winq, kBlockSlices, pk, &marked,
                                     lab_acc);
            lab_acc[2] += __builtin_readcyclecounter() - lab_c0;
#else
            if (stats) // (wave-uniform: a kernel argument)
                j = run_block_spec<true, true>(tile0, cand_addr, kCandBufBytes, outb_addr, dummy_addr, sf, kc, limq, winq, kBlockSlices, pk, &marked);
            else
                j = run_block_spec<true>(tile0, cand_addr, kCandBufBytes, outb_addr, dummy_addr, sf, kc, limq, winq, kBlockSlices, pk);
#endif
            j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
            if (stats && marked) atomicAdd(&stats[4], (unsigned long long)marked);
            cheap_declined = j < kBlockSlices;
            QOA_LAB_FORM(0, j + (cheap_declined ? 1u : 0u));
        }
        if (j < kBlockSlices) { // the ladder: a block whose 16 slices are full in every chain of the wave goes to block statements
            const bool whole = !force_tail_path && (unsigned long long)(k0 + kBlockSlices) * 20u <= samples;
            const bool block_ok = __builtin_amdgcn_ballot_w64(!whole) == 0ull;
            // the packed forms, while every chain of the wave is packed and they are not sitting out
            if (block_ok && pk_cooldown == 0 && __builtin_amdgcn_ballot_w64(!packed) == 0ull) {
                if (allow_spec && spec_cooldown == 0) {
                    if (lds_tables) {
                        j = run_block_spec_ldsq(tile0, cand_addr, kCandBufBytes, outb_addr, dummy_addr, sf, kc, limq, winq, qtab_addr,
                                                kBlockSlices, pk);
                        QOA_LAB_FORM(1, j + (j < kBlockSlices ? 1u : 0u));
                    } else {
                        // cheap quotient first; a slice it declines is repeated with the exact quotient and the cheap form takes
                        // the block up again behind it -- unless declines come in quick succession (a loud passage)
                        bool proven = true;
                        while (j < kBlockSlices && proven) {
                            const bool cheap = allow_cheap && cheap_cooldown == 0;
                            if (cheap) {
                                if (!cheap_declined) {
                                    unsigned marked = 0;
                                    const unsigned j_was = j; // (the per-wavefront counts)
                                    (void)j_was;
                                    if (stats)
                                        j += run_block_spec<true, true>(tile0 + 4u * kSliceDwords * j, cand_addr, kCandBufBytes, outb_addr + 8u * j,
                                                                        dummy_addr, sf, kc, limq, winq, kBlockSlices - j, pk, &marked);
                                    else
                                        j += run_block_spec<true>(tile0 + 4u * kSliceDwords * j, cand_addr, kCandBufBytes, outb_addr + 8u * j,
                                                                  dummy_addr, sf, kc, limq, winq, kBlockSlices - j, pk);
                                    j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
                                    if (stats && marked) atomicAdd(&stats[4], (unsigned long long)marked);
                                    QOA_LAB_FORM(0, j - j_was + (j < kBlockSlices ? 1u : 0u));
                                }
                                cheap_declined = false;
                                if (j == kBlockSlices) break;
                                if (stats && sf == 0) atomicAdd(&stats[3], 1ull);
                                if (blk - last_cheap_decline <= 1u) cheap_cooldown = kCheapCooldownBlocks + 1;
                                last_cheap_decline = blk;
                            }
                            const unsigned n_exact = cheap && cheap_cooldown == 0 ? 1u : kBlockSlices - j;
                            unsigned r = run_block_spec<false>(tile0 + 4u * kSliceDwords * j, cand_addr, kCandBufBytes,
                                                               outb_addr + 8u * j, dummy_addr, sf, kc, limq, winq, n_exact, pk);
                            r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
                            j += r;
                            proven = r == n_exact;
                            QOA_LAB_FORM(1, r + (proven ? 0u : 1u));
                        }
                    }
                    j = (unsigned)__builtin_amdgcn_readfirstlane((int)j); // (wave-uniform: all forms return an SGPR)
                    if (j < kBlockSlices) { // slice j could not be proven: the penalty / full form takes over for the rest of the block
                        spec_streak = min(spec_streak + 1u, kSpecStreakMax);
                        if (spec_streak >= 2u) spec_cooldown = (kSpecMissCooldownBlocks << (spec_streak - 2u)) + 1;
                        if (stats && sf == 0) atomicAdd(&stats[2], 1ull);
                    } else {
                        spec_streak = 0;
                    }
                }
                if (j < kBlockSlices && allow_pen && pen_cooldown == 0) {
                    // where the speculation missed (and while it sits out): cheap-quotient steps with the penalty arithmetic
                    const unsigned r = run_block_pen(tile0 + 4u * kSliceDwords * j, cand_addr, kCandBufBytes, outb_addr + 8u * j, dummy_addr,
                                                     sf, kc, kBlockSlices - j, pk);
                    j += (unsigned)__builtin_amdgcn_readfirstlane((int)r);
                    QOA_LAB_FORM(2, (unsigned)__builtin_amdgcn_readfirstlane((int)r) + (j < kBlockSlices ? 1u : 0u));
                    if (j < kBlockSlices) { // declined (loud material, weights beyond the packed window): the full form goes on
                        pen_streak = min(pen_streak + 1u, 6u);
                        if (pen_streak >= 2u) pen_cooldown = (kSpecCooldownBlocks << (pen_streak - 2u)) + 1;
                        if (stats && sf == 0) atomicAdd(&stats[7], 1ull);
                    } else {
                        pen_streak = 0;
                    }
                }
                if (j < kBlockSlices && allow_wide && wide_cooldown == 0) {
                    // the penalty form declined (or sits out): ranks beyond the saturating rank's range, marked candidates
                    // without a bound -- noise.  Penalty-free winners assumed, proven per slice (run_block_wide).
                    const unsigned r = run_block_wide(tile0 + 4u * kSliceDwords * j, cand_addr, kCandBufBytes, outb_addr + 8u * j, dummy_addr,
                                                      sf, kc, limq, winq, kBlockSlices - j, pk);
                    const unsigned rw = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
                    j += rw;
                    QOA_LAB_FORM(3, rw + (j < kBlockSlices ? 1u : 0u));
                    if (stats && lane == 0 && rw) atomicAdd(&stats[9], (unsigned long long)rw); // wavefront-slices the wide form settled
                    if (j < kBlockSlices) { // a winner with a penalty, or a candidate beyond the packed window: the full form goes on
                        if (stats && sf == 0) atomicAdd(&stats[8], 1ull);
                        wide_streak = min(wide_streak + 1u, 6u);
                        if (wide_streak >= 2u) wide_cooldown = (kSpecCooldownBlocks << (wide_streak - 2u)) + 1;
                    } else {
                        wide_streak = 0;
                    }
                }
                if (j < kBlockSlices) {
                    const unsigned j0 = j;
                    j += run_block_pk(tile0 + 4u * kSliceDwords * j0, cand_addr, outb_addr + 8u * j0, dummy_addr, sf, kc, kBlockSlices - j0, pk);
                    j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
                    QOA_LAB_FORM(4, j - j0 + (j < kBlockSlices ? 1u : 0u));
                    if (j < kBlockSlices) { // a candidate left the i16 window: the 32-bit statement (or the slice loop) goes on
                        pk_streak = min(pk_streak + 1u, kPkStreakMax);
                        if (pk_streak >= 2u && allow_b32) pk_cooldown = (kPkCooldownBlocks << (pk_streak - 2u)) + 1;
                        if (allow_b32 && stats && sf == 0) atomicAdd(&stats[0], 1ull); // (else the slice loop below counts it)
                    } else {
                        pk_streak = 0;
                    }
                }
            }
            // whatever is left of a whole block -- the packed forms declined a slice, are sitting out, or a chain's state
            // does not fit them -- runs on the 32-bit block statement, which cannot decline
            if (block_ok && allow_b32 && j < kBlockSlices) {
                if (packed) pk_unpack(pk, st);
                else if (allow_packed && stats && sf == 0) atomicAdd(&stats[1], (unsigned long long)(kBlockSlices - j));
                if (stats && lane == 0) atomicAdd(&stats[5], (unsigned long long)(kBlockSlices - j));
                QOA_LAB_FORM(5, kBlockSlices - j);
#ifdef QOA_LAB_HWID // (the lab's per-wavefront counts make the compiler lose sight of j's uniformity)
                j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
#endif
                run_block_32(tile0 + 4u * kSliceDwords * j, cand_addr, outb_addr + 8u * j, dummy_addr, sf, kc, kBlockSlices - j, st);
                j = kBlockSlices;
                packed = pk_try_pack(st, pk) && allow_packed;
            }
            if (spec_cooldown) spec_cooldown--;
            if (cheap_cooldown) cheap_cooldown--;
            if (pen_cooldown) pen_cooldown--;
            if (pk_cooldown) pk_cooldown--;
            if (wide_cooldown) wide_cooldown--;
            const unsigned unsteady = spec_cooldown | cheap_cooldown | pk_cooldown | spec_streak | (blk - last_cheap_decline <= 1u ? 1u : 0u);
            ok_blocks = steady_flags && packed && __builtin_amdgcn_readfirstlane((int)unsteady) == 0 ? full_blocks : 0u;
        }
        QOA_LAB_MARK(6) // [6] the block statements and their dispatch
        QOA_LAB_FORM(6, nj - j);
        QOA_LAB_FORM(7, 1u);
        for (; j < nj; j++) {
            const int len = (int)min(samples - (k0 + j) * 20u, 20u);
            const bool full = (len == 20) && !force_tail_path;

            typedef __attribute__((address_space(3))) int *lds_ptr;
            const unsigned tile_addr = (unsigned)(size_t)(lds_ptr)(tile + j * kSliceDwords);
            if (__builtin_amdgcn_ballot_w64(!(full && packed)) == 0ull) { // every chain of the wave can go packed
                PkResult pr;
                if (stats && lane == 0) atomicAdd(&stats[6], 1ull); // wavefront-slices on the one-slice packed statement
                run_slice_pk(tile_addr, sf, kc, pk, pr);
                if (__builtin_amdgcn_ballot_w64(pr.pen >= kPkPenLimit) == 0ull) { // all 64 candidates provably exact
                    const unsigned win = pick_winner(pr.e, pr.pen, pr.aerr0, sf);
                    const int src_lane = (int)((row_base | win) << 2);
                    const unsigned w_hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)pr.s_hi);
                    const unsigned w_lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)pr.s_lo);
                    pk.w01 = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)pr.lms.w01);
                    pk.w23 = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)pr.lms.w23);
                    pk.pa = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)pr.lms.pa);
                    pk.pb = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)pr.lms.pb);
                    if ((unsigned)sf == j) out_tile[j] = make_uint2(w_hi, w_lo);
                    continue;
                }
                // else: fall through and evaluate the slice again with 32-bit weights from the same entry state
                if (stats && sf == 0) atomicAdd(&stats[0], 1ull); // (chain, slice) pairs the packed path declined
            } else if (full && allow_packed && stats && sf == 0) {
                atomicAdd(&stats[1], 1ull); // (chain, slice) pairs on the 32-bit path because a state was not admissible
            }
            if (packed) pk_unpack(pk, st);
            if (stats && lane == 0) atomicAdd(&stats[5], 1ull); // wavefront-slices on the 32-bit / tail path

            CandResult r;
            if (__builtin_amdgcn_ballot_w64(!full) == 0ull) {
                run_slice_asm(tile_addr, sf, kc, st, r);
            } else {
                int smp[20];
                unpack_tile_slice(tile + j * kSliceDwords, smp); // same address in all 16 lanes: LDS broadcast
                run_slice_tail(smp, len, sf, kc, st, r);
            }
            const unsigned win = pick_winner(r.e, r.pen, r.aerr0, sf);
            const int src_lane = (int)((row_base | win) << 2);

            // LMS carry: every lane of the row takes the winner's end state (src/lib.rs:482);
            // lane j keeps the winner's slice for the block flush (src/lib.rs:488).
            unsigned w_hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)r.s_hi);
            unsigned w_lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane, (int)r.s_lo);
            st.h0 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.h0);
            st.h1 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.h1);
            st.h2 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.h2);
            st.h3 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.h3);
            st.w0 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.w0);
            st.w1 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.w1);
            st.w2 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.w2);
            st.w3 = __builtin_amdgcn_ds_bpermute(src_lane, r.lms.w3);
            if ((unsigned)sf == j) out_tile[j] = make_uint2(w_hi, w_lo);
            packed = pk_try_pack(st, pk) && allow_packed;
        }
    }
    if constexpr (HELPER) {
        pair_barrier(); // the helper flushes the last block behind it
    } else {
        flush_block(out_tile, nblocks - 1); // the last (possibly partial) block
    }
#ifdef QOA_LAB_HWID
    if (stats && lane == 0 && wave < 2048u) stats[10 + 4096 + wave] = __builtin_readcyclecounter() - hwid_t0;
#else
    if constexpr (REC) {
        if (lane == 0 && wave_rec) { // (lane 0 = the wave's first chain: the longest one, clips are handed out longest first)
            const unsigned long long dt = __builtin_readcyclecounter() - rec_t0;
            unsigned *r = wave_rec + (size_t)wave * kWaveRecWords;
#pragma unroll
            for (int k = 0; k < kWaveRecForms; k++) r[k] = rec_cnt[k];
            r[kWaveRecForms] = (unsigned)dt;
            r[kWaveRecForms + 1] = (unsigned)(dt >> 32);
            r[kWaveRecForms + 2] = nslices;
            r[kWaveRecForms + 3] = 1u; // record valid
        }
    }
#endif
    if (packed) pk_unpack(pk, st);
    if (sf == 0) {
        int32_t *s = lms_state + (size_t)chain_id * 8;
        s[0] = st.h0; s[1] = st.h1; s[2] = st.h2; s[3] = st.h3;
        s[4] = st.w0; s[5] = st.w1; s[6] = st.w2; s[7] = st.w3;
    }
#ifdef QOA_LAB_STAMPS
    if (stats && lane == 0) {
        atomicAdd(&stats[10], lab_acc[0]); atomicAdd(&stats[11], lab_acc[1]); atomicAdd(&stats[12], lab_acc[2]);
        atomicAdd(&stats[13], __builtin_readcyclecounter() - lab_k0);
        for (int i = 3; i < 8; i++) atomicAdd(&stats[11 + i], lab_acc[i]);
    }
#endif
}

#ifndef QOA_REC_TU // (qoa_kernels_rec.hip compiles this file again for the recording instantiations alone: see the launchers)
// ------------------------------------------------------------------------------------------------
// Test hook: one slice, all 16 candidates, through the very same device functions as K1.  Returns every
// candidate's total rank, first-sample rank, packed slice and end state so that tests can compare the
// search lane by lane with the oracle (tests/test_gpu_parity.py, KAT-2).
// ------------------------------------------------------------------------------------------------
__global__ void qoa_debug_slice_kernel(const int32_t *__restrict__ state, const int32_t *__restrict__ samples, int len,
                                       int generic, unsigned long long *__restrict__ tot,
                                       unsigned long long *__restrict__ fst, unsigned long long *__restrict__ slice,
                                       int32_t *__restrict__ end_state) {
    const int sf = (int)(threadIdx.x & 15u);
    if (threadIdx.x >= 16) return;
    StepConst kc;
    kc.recip = c_recip[sf];
    kc.dq_lo = (unsigned)c_mags[sf][0] | ((unsigned)c_mags[sf][1] << 16);
    kc.dq_hi = (unsigned)c_mags[sf][2] | ((unsigned)c_mags[sf][3] << 16);
    kc.selbase = 0x0c0c0100u;
    kc.hi = 32767;
    Lms st;
    st.h0 = state[0]; st.h1 = state[1]; st.h2 = state[2]; st.h3 = state[3];
    st.w0 = state[4]; st.w1 = state[5]; st.w2 = state[6]; st.w3 = state[7];
    int smp[20];
#pragma unroll
    for (int i = 0; i < 20; i++) smp[i] = samples[i];
    CandResult r;
    // generic: 0 = one-asm 32-bit slice (samples staged in LDS), 1 = rolled tail path, 2 = 20 cand_step() statements,
    // 3 = packed fast path
    __shared__ __attribute__((aligned(16))) int s_dbg[kSliceDwords]; // one slice in the tile layout
    bool declined = false;
    if ((generic == 0 || generic == 3 || generic == 4) && len == 20) {
        if (sf == 0) {
#pragma unroll
            for (int k = 0; k < 10; k++) s_dbg[2 + k] = (int)(((unsigned)smp[2 * k] & 0xffffu) | ((unsigned)smp[2 * k + 1] << 16));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the 16 lanes of the wave see lane 0's LDS writes (in-order LDS)
        typedef __attribute__((address_space(3))) int *lds_ptr;
        if (generic == 0) {
            run_slice_asm((unsigned)(size_t)(lds_ptr)s_dbg, sf, kc, st, r);
        } else { // the packed fast path, exactly as K1 uses it
            PkLms pk;
            PkResult pr;
            declined = __builtin_amdgcn_ballot_w64(!pk_try_pack(st, pk)) != 0ull;
            if (!declined) {
                run_slice_pk((unsigned)(size_t)(lds_ptr)s_dbg, sf, kc, pk, pr);
                declined = __builtin_amdgcn_ballot_w64(pr.pen >= kPkPenLimit) != 0ull;
                pk_unpack(pr.lms, r.lms);
                r.e = pr.e; r.pen = pr.pen; r.aerr0 = pr.aerr0; r.s_hi = pr.s_hi; r.s_lo = pr.s_lo;
            }
            if (declined && generic == 3) run_slice_asm((unsigned)(size_t)(lds_ptr)s_dbg, sf, kc, st, r); // what K1 does then
        }
    } else if (generic == 1 || len != 20) {
        run_slice_tail(smp, len, sf, kc, st, r);
    } else {
        run_slice_full(smp, sf, kc, st, r);
    }
    tot[sf] = r.e + r.pen;
    // [35:32] winner of the row, [31:0] |err0|, bit 63: the packed path declined this slice (generic == 3)
    fst[sf] = ((unsigned long long)declined << 63) | ((unsigned long long)pick_winner(r.e, r.pen, r.aerr0, sf) << 32) | r.aerr0;
    slice[sf] = ((unsigned long long)r.s_hi << 32) | r.s_lo;
    int32_t *o = end_state + sf * 8;
    o[0] = r.lms.h0; o[1] = r.lms.h1; o[2] = r.lms.h2; o[3] = r.lms.h3;
    o[4] = r.lms.w0; o[5] = r.lms.w1; o[6] = r.lms.w2; o[7] = r.lms.w3;
}
#endif // QOA_REC_TU
// ------------------------------------------------------------------------------------------------
// host-side launchers
//
// The recording instantiations (REC = true: 8 channel counts x with / without helper wavefronts, default packing only) are
// compiled in a translation unit of their own, qoa_kernels_rec.hip = this file with QOA_REC_TU defined: they take hipcc as
// long again as everything else in here, and make builds the two objects side by side.
// ------------------------------------------------------------------------------------------------
#ifndef QOA_REC_TU
bool encode_launch_has_helper(int cpw, unsigned n, unsigned flags) {
    // A launch that leaves at least half of the chip's 1024 SIMDs idle gives every compute wavefront a helper wavefront
    // (HELPER workgroups: see the kernel).  flags bit 8: never, bit 9: always (tests, experiments).
    const unsigned waves = (n + (unsigned)cpw - 1) / (unsigned)cpw;
    return (flags & 256u) == 0 && (waves <= kHelperMaxWaves || (flags & 512u) != 0);
}

#endif // (qoa_kernels_rec.hip calls the main object's copy)

template <int CPW, bool REC = false>
static hipError_t launch_encode_nch(int nch, const int16_t *pcm, uint8_t *out, const EncChain *chains, unsigned n,
                                    int32_t *lms, unsigned flags, unsigned long long *stats, unsigned *wave_rec, hipStream_t stream) {
    // Fewer wavefronts than one workgroup holds (a single mono / stereo / 3-channel clip): one wavefront per workgroup, so that
    // each lands on a compute unit of its own.  Two wavefronts of one partly filled 4-wave workgroup sometimes end up on the
    // same SIMD, where they share the issue port: the same single-clip encode took 160 or 340 ms (round-1 verdict, weak #9;
    // now 150-200).  From 4 wavefronts up the 4-wave workgroups stay: 8 single-wave workgroups on 8 CUs made the 8-channel
    // streaming call slower (0.48 vs 0.38 ms).
    unsigned waves = (n + CPW - 1) / CPW;
    {
        if (encode_launch_has_helper(CPW, n, flags)) {
            dim3 g(waves), b(128);
#define QOA_LAUNCH(N)                                                                                                    \
    case N:                                                                                                              \
        hipLaunchKernelGGL((qoa_encode_kernel<CPW, N, true, REC>), g, b, 0, stream, pcm, out, chains, n, lms, flags, stats, wave_rec); \
        break;
            switch (nch) {
                QOA_LAUNCH(1) QOA_LAUNCH(2) QOA_LAUNCH(3) QOA_LAUNCH(4) QOA_LAUNCH(5) QOA_LAUNCH(6) QOA_LAUNCH(7) QOA_LAUNCH(8)
            default: return hipErrorInvalidValue;
            }
#undef QOA_LAUNCH
            return hipGetLastError();
        }
    }
    const unsigned kWavesPerBlock = waves < 4 ? 1 : 4;
    unsigned blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    dim3 g(blocks), b(64 * kWavesPerBlock);
#define QOA_LAUNCH(N)                                                                                              \
    case N:                                                                                                        \
        hipLaunchKernelGGL((qoa_encode_kernel<CPW, N, false, REC>), g, b, 0, stream, pcm, out, chains, n, lms, flags, stats, wave_rec); \
        break;
    switch (nch) {
        QOA_LAUNCH(1) QOA_LAUNCH(2) QOA_LAUNCH(3) QOA_LAUNCH(4) QOA_LAUNCH(5) QOA_LAUNCH(6) QOA_LAUNCH(7) QOA_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef QOA_LAUNCH
    return hipGetLastError();
}

#ifdef QOA_REC_TU
hipError_t launch_encode_rec(int nch, const int16_t *pcm, uint8_t *out, const EncChain *chains, unsigned n, int32_t *lms,
                             unsigned flags, unsigned long long *stats, unsigned *wave_rec, hipStream_t stream) {
    return launch_encode_nch<4, true>(nch, pcm, out, chains, n, lms, flags, stats, wave_rec, stream);
}
#else
hipError_t launch_encode(int cpw, int nch, const int16_t *pcm, uint8_t *out, const EncChain *chains, unsigned n,
                         int32_t *lms, unsigned flags, unsigned long long *stats, hipStream_t stream, unsigned *wave_rec) {
    if (n == 0) return hipSuccess;
    if (wave_rec) { // the recording instantiations exist for the default packing only (qoa_kernels_rec.hip)
        if (cpw != 4) return hipErrorInvalidValue;
        return launch_encode_rec(nch, pcm, out, chains, n, lms, flags, stats, wave_rec, stream);
    }
    switch (cpw) {
    case 1: return launch_encode_nch<1>(nch, pcm, out, chains, n, lms, flags, stats, nullptr, stream);
    case 2: return launch_encode_nch<2>(nch, pcm, out, chains, n, lms, flags, stats, nullptr, stream);
    case 4: return launch_encode_nch<4>(nch, pcm, out, chains, n, lms, flags, stats, nullptr, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_debug_slice(const int32_t *state, const int32_t *samples, int len, int generic,
                              unsigned long long *tot, unsigned long long *fst, unsigned long long *slice,
                              int32_t *end_state, hipStream_t stream) {
    hipLaunchKernelGGL(qoa_debug_slice_kernel, dim3(1), dim3(64), 0, stream, state, samples, len, generic, tot, fst,
                       slice, end_state);
    return hipGetLastError();
}
#endif // QOA_REC_TU

} // namespace qoa
