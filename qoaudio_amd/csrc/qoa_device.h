// qoa_device.h -- device helpers shared by the encode (qoa_kernels.hip) and decode (qoa_decode.hip) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "qoa_format.h"

namespace qoa {

// one copy per translation unit (no relocatable device code needed)
static __constant__ int c_recip[16] = QOA_RECIPROCALS;
static __constant__ int c_mags[16][4] = QOA_DEQUANT_MAGS;

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// 24-bit multiply-add as a single instruction with EXACT wrapping semantics: the low 32 bits of
// sext24(a) * sext24(b) + c.  Not HIP's __mul24: that is `mul nsw` in the device library, i.e. signed
// overflow is undefined, and hipcc then narrows a u64 rank sum to 32 bits (caught by the parity tests:
// err^2 reaches 2^32).  Inline asm keeps the instruction and drops the assumption.
__device__ __forceinline__ int mad24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int mul32(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
// clamp(v, lo, hi); the backend folds min(max()) with constant bounds into one v_med3_i32
__device__ __forceinline__ int med3(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ unsigned uabs(int v) { unsigned s = (unsigned)(v >> 31); return ((unsigned)v ^ s) - s; }

struct Lms {
    int h0, h1, h2, h3; // history, oldest first (src/lib.rs:72)
    int w0, w1, w2, w3; // weights (src/lib.rs:73)
};

// full 32-bit wrapping multiply-add in ONE instruction: the low dword of v_mad_u64_u32.  Measured on gfx950
// (tools/ubench): same issue cost as v_mad_i32_i24 and as v_mul_lo_u32, so the dot products need no 24-bit
// operand guard and no second code path.  The high dword is garbage and ignored.
__device__ __forceinline__ unsigned long long mad32(int a, int b, unsigned long long acc) {
    unsigned long long r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
    return r;
}
__device__ __forceinline__ void sign_mults(const Lms &l, int &s0, int &s1, int &s2, int &s3) {
    s0 = (l.h0 >> 31) | 1; s1 = (l.h1 >> 31) | 1; s2 = (l.h2 >> 31) | 1; s3 = (l.h3 >> 31) | 1;
}

__device__ __forceinline__ unsigned bswap32(unsigned v) { return __builtin_bswap32(v); }

// compile-time counted loop: the body gets std::integral_constant<int, I> (asm "n" operands need true constants)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


} // namespace qoa
