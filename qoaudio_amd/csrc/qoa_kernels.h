// qoa_kernels.h -- launchers of the gfx950 kernels (qoa_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "qoa_format.h"

namespace qoa {

// flags bit0: run every slice through the rolled tail path (debug / parity tests)
//       bit1: start every chain from a fresh encoder's LMS state instead of loading d_lms
//       bit2: never take the i16-packed fast path      bit3: no speculative (penalty-free) steps
//       bit4: speculative steps with the quantiser / dequantiser in an LDS table (fewer issue slots, one LDS round trip per
//             step: for launches with several wavefronts per SIMD)
//       bit5: no cheap-quotient speculative steps (run_block_spec<true>)
//       bit6: no penalty form   bit7: no 32-bit block statement   bit8: no helper wavefronts   bit9: helper wavefronts always
//       bit10: no wide form
// d_stats (may be null: then the kernel executes no atomic; else 10 x u64, the counters include/qoaudio_amd.h lists under
//          qoa_encode_plan_counters): [0] += (chain, slice) pairs the packed path declined and the 32-bit path re-evaluated,
//          [1] += full slices evaluated by the 32-bit path because a chain's state was not admissible,
//          [2] += (chain, slice) pairs whose speculative evaluation was repeated by the full form,
//          [3] += slices the cheap-quotient form declined, [4] += (candidate, slice) pairs it marked,
//          [5] / [6] += wavefront-slices on the 32-bit step / the one-slice packed statement, [7] += penalty-form declines,
//          [8] += wide-form declines, [9] += wavefront-slices the wide form settled
// d_wave_rec (may be null; needs cpw == 4): kWaveRecWords x u32 per compute wavefront (wave w at [w * kWaveRecWords]), written
//          once at the wavefront's exit by the recording instantiation of the kernel: [0 .. kWaveRecForms) wavefront-slices
//          attempted per step form (0 cheap-quotient, 1 exact-quotient speculative / LDS-table, 2 penalty, 3 wide, 4 full packed,
//          5 32-bit block statement, 6 one-slice statements, 7 blocks), then the wavefront's cycles (lo, hi), the slice count of
//          its first chain, and 1 (record valid).  The caller zeroes the buffer before the launch.
constexpr int kWaveRecForms = 8;
constexpr int kWaveRecWords = kWaveRecForms + 4;
hipError_t launch_encode(int cpw, int nch, const int16_t *d_pcm, uint8_t *d_out, const EncChain *d_chains,
                         unsigned n_chains, int32_t *d_lms, unsigned flags, unsigned long long *d_stats,
                         hipStream_t stream, unsigned *d_wave_rec = nullptr);
// (the recording instantiations, cpw = 4: a translation unit of their own, qoa_kernels_rec.hip; called by launch_encode)
hipError_t launch_encode_rec(int nch, const int16_t *d_pcm, uint8_t *d_out, const EncChain *d_chains, unsigned n_chains,
                             int32_t *d_lms, unsigned flags, unsigned long long *d_stats, unsigned *d_wave_rec,
                             hipStream_t stream);

// Does launch_encode give this launch helper wavefronts (two-wavefront HELPER workgroups)?  flags as above (bit8 never,
// bit9 always); the one place that decides it -- qoa_encode_plan_launch_info reports it from here.
bool encode_launch_has_helper(int cpw, unsigned n_chains, unsigned flags);

hipError_t launch_decode(const uint8_t *d_in, int16_t *d_out, const DecChain *d_chains, unsigned n_chains,
                         unsigned *d_mismatches, hipStream_t stream);

// K2 for 1..8 channels: one DecChain entry per FRAME (its `channel` field is ignored), LDS-transposed output
// d_scratch: kDecodeScratchBytes of device memory (sink for the stores of pieces beyond a frame's end)
constexpr unsigned kDecodeScratchBytes = 1024;
// whole_pieces: every frame's PCM byte count (n_out * nch * 2) is a multiple of 16 -> the variant without the
// 2-byte tail path (no store under a branch, counted vmcnt waits)
hipError_t launch_decode_frames(int nch, bool whole_pieces, const uint8_t *d_in, int16_t *d_out,
                                const DecChain *d_frames, unsigned n_frames, unsigned *d_mismatches,
                                uint8_t *d_scratch, hipStream_t stream);

// K2u, the fast path for uniform frames (rows == 256, n_out == 5120) of 1 .. 8 channels: any number of frames (the last
// wavefront's spare frame slots decode its last frame once more: same bytes to the same addresses), every frame's PCM on
// a dword grid (streamed stores when all frames of a wavefront start on a 128-byte line, write-back stores otherwise).
// decode_uniform_group(nch) = frames per wavefront (0: no K2u for this channel count).
constexpr unsigned decode_uniform_group(int nch) { return nch >= 1 && nch <= 8 ? 64u / (unsigned)nch : 0u; }
hipError_t launch_decode_uniform(int nch, const uint8_t *d_in, int16_t *d_out, const DecChain *d_frames, unsigned n_frames,
                                 unsigned *d_mismatches, hipStream_t stream);
// K2u's absolute-line variant (round 5): uniform frames whose PCM address is off the 128-byte grid but a multiple of 8 -- the
// flush works on the frame's absolute cache lines (whole-line streamed stores; the frame's first and last line partial).
// Also correct, just not needed, for frames on the grid.  decode_uniform_abs_supported(nch): the channel counts it is built for.
constexpr bool decode_uniform_abs_supported(int nch) { return nch == 2 || nch == 4 || nch == 8; }
hipError_t launch_decode_uniform_abs(int nch, const uint8_t *d_in, int16_t *d_out, const DecChain *d_frames, unsigned n_frames,
                                     unsigned *d_mismatches, hipStream_t stream);

// test hook: one slice, 16 candidates (device pointers; tot/fst/slice: 16 x u64, end_state: 16 x 8 x i32)
hipError_t launch_debug_slice(const int32_t *d_state, const int32_t *d_samples, int len, int generic,
                              unsigned long long *d_tot, unsigned long long *d_fst, unsigned long long *d_slice,
                              int32_t *d_end_state, hipStream_t stream);

} // namespace qoa
