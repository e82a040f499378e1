// Instruction-issue microbenchmarks for gfx950 integer VALU ops (development aid).
// Each test times REP x 32 instructions in one wave with s_memtime; run with 1 wave/SIMD (64*4 threads)
// and 2 waves/SIMD (64*8 threads) per block, one block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 64

#define BODY8(X) X X X X X X X X
#define BODY32(X) BODY8(X) BODY8(X) BODY8(X) BODY8(X)

#define KERNEL(name, ASMSTR)                                                                   \
    __global__ void name(unsigned long long *out, int *sink, int seed) {                       \
        int a = threadIdx.x + seed, b = threadIdx.x * 3 + 1, c = seed, d = 7, e = 11, f = 13, g = 17, h = 19; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int i = 0; i < REP; i++) {                                                        \
            asm volatile(BODY32(ASMSTR) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)); \
        }                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                          \
        sink[threadIdx.x] = a + b + c + d + e + f + g + h;                                     \
    }

// dependent chains (each instr depends on the previous through %0)
KERNEL(k_add_dep, "v_add_u32 %0, %0, %1\n")
KERNEL(k_add_ind, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n")
KERNEL(k_mad24_dep, "v_mad_i32_i24 %0, %0, %1, %2\n")
KERNEL(k_mad24_ind, "v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %3, %3, %1, %2\n v_mad_i32_i24 %4, %4, %1, %2\n v_mad_i32_i24 %5, %5, %1, %2\n")
KERNEL(k_mul24_dep, "v_mul_i32_i24 %0, %0, %1\n")
KERNEL(k_mullo_dep, "v_mul_lo_u32 %0, %0, %1\n")
KERNEL(k_mullo_ind, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1\n")
KERNEL(k_med3_dep, "v_med3_i32 %0, %0, %1, %2\n")
KERNEL(k_perm_dep, "v_perm_b32 %0, %0, %1, %2\n")
KERNEL(k_add3_dep, "v_add3_u32 %0, %0, %1, %2\n")
KERNEL(k_ashr_dep, "v_ashrrev_i32 %0, 1, %0\n")
KERNEL(k_xor_dep, "v_xor_b32 %0, %0, %1\n")
KERNEL(k_dpp_min_dep, "s_nop 1\n v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_add_ind, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_madu64, "v_mad_u64_u32 v[10:11], vcc, %0, %1, v[10:11]\n")
KERNEL(k_bfe_dep, "v_bfe_i32 %0, %0, 4, 24\n")
KERNEL(k_lshlor_dep, "v_lshl_or_b32 %0, %0, 3, %1\n")
KERNEL(k_cndmask_dep, "v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL(k_cmp_cnd, "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n")
KERNEL(k_pk_add_dep, "v_pk_add_u16 %0, %0, %1\n")
KERNEL(k_pk_mad_dep, "v_pk_mad_i16 %0, %0, %1, %2\n")
KERNEL(k_pk_mul_dep, "v_pk_mul_lo_u16 %0, %0, %1\n")
KERNEL(k_dot2_dep, "v_dot2_i32_i16 %0, %1, %2, %0\n")
KERNEL(k_dot2_ind, "v_dot2_i32_i16 %0, %1, %2, %0\n v_dot2_i32_i16 %3, %1, %2, %3\n v_dot2_i32_i16 %4, %1, %2, %4\n v_dot2_i32_i16 %5, %1, %2, %5\n")
KERNEL(k_dot4_dep, "v_dot4_i32_i8 %0, %1, %2, %0\n")
KERNEL(k_bperm, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_bperm_pipe, "ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %2, %1, %2\n ds_bpermute_b32 %3, %1, %3\n ds_bpermute_b32 %4, %1, %4\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n s_nop 3\n v_add_u32 %0, s20, %0\n")
KERNEL(k_fma_dep, "v_fma_f32 %0, %0, %1, %2\n")
KERNEL(k_fma_ind, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n")
KERNEL(k_snop, "s_nop 0\n")
KERNEL(k_salu, "s_add_u32 s20, s20, 1\n")
KERNEL(k_mix_valu_salu, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n")
KERNEL(k_sad, "v_sad_u32 %0, %0, %1, %2\n")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2\n")
KERNEL(k_mul_u24_hi, "v_mul_hi_i32_i24 %0, %0, %1\n")

KERNEL(k_add_e64_ind, "v_add_u32_e64 %0, %0, %1\n v_add_u32_e64 %2, %2, %1\n v_add_u32_e64 %3, %3, %1\n v_add_u32_e64 %4, %4, %1\n")
KERNEL(k_or_lit_ind, "v_or_b32 %0, 0x10001, %0\n v_or_b32 %2, 0x10001, %2\n v_or_b32 %3, 0x10001, %3\n v_or_b32 %4, 0x10001, %4\n")
KERNEL(k_mix_23, "v_add_u32 %0, %0, %1\n v_mad_i32_i24 %2, %2, %1, %5\n v_add_u32 %3, %3, %1\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_dot2c_ind, "v_dot2c_i32_i16 %0, %1, %5\n v_dot2c_i32_i16 %2, %1, %5\n v_dot2c_i32_i16 %3, %1, %5\n v_dot2c_i32_i16 %4, %1, %5\n")
KERNEL(k_dpp_filled, "v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_med3_ind, "v_med3_i32 %0, %0, %1, %5\n v_med3_i32 %2, %2, %1, %5\n v_med3_i32 %3, %3, %1, %5\n v_med3_i32 %4, %4, %1, %5\n")
KERNEL(k_snop1, "s_nop 1\n")
KERNEL(k_mov_ind, "v_mov_b32 %0, %1\n v_mov_b32 %2, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1\n")


// run lengths of the two issue classes: does a second wavefront's simple ALU op dual-issue only inside long runs?
KERNEL(k_run_2_2, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n" "v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_run_4_4, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_run_8_8, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_run_16_16, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_run_12_4, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" "v_mad_i32_i24 %0, %0, %1, %5\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_run_3_1, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_run_1_3, "v_add_u32 %0, %0, %1\n v_mad_i32_i24 %2, %2, %1, %5\n v_mad_i32_i24 %3, %3, %1, %5\n v_mad_i32_i24 %4, %4, %1, %5\n")
KERNEL(k_maxmin, "v_max_i32 %0, %0, %1\n v_min_i32 %2, %2, %1\n v_max_i32 %3, %3, %1\n v_min_i32 %4, %4, %1\n")
KERNEL(k_sub_and, "v_sub_u32 %0, %0, %1\n v_and_b32 %2, %2, %1\n v_lshlrev_b32 %3, 1, %3\n v_or_b32 %4, %4, %1\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16\n v_alignbit_b32 %2, %2, %1, 16\n v_alignbit_b32 %3, %3, %1, 16\n v_alignbit_b32 %4, %4, %1, 16\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %2, %2, 3, %1\n v_lshl_add_u32 %3, %3, 3, %1\n v_lshl_add_u32 %4, %4, 3, %1\n")
KERNEL(k_sdwa_add, "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %2, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %3, %3, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %4, %4, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n")
KERNEL(k_pk_ashr, "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %2, 15, %2 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %3, 15, %3 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %4, 15, %4 op_sel_hi:[0,1]\n")
KERNEL(k_bfe_u, "v_bfe_u32 %0, %0, 3, 6\n v_bfe_u32 %2, %2, 3, 6\n v_bfe_u32 %3, %3, 3, 6\n v_bfe_u32 %4, %4, 3, 6\n")

// r02: does an SDWA / packing consumer of a just-written VGPR cost more than a plain one?  (pairs are dependent both ways via %0)
KERNEL(k_pair_plain, "v_ashrrev_i32 %0, 1, %0\n v_sub_u32 %0, %1, %0\n")
KERNEL(k_pair_sdwa, "v_ashrrev_i32 %0, 1, %0\n v_sub_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_pair_sdwa_src, "v_ashrrev_i32 %0, 1, %0\n v_sub_u32_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_pair_sdwa_d2, "v_ashrrev_i32 %0, 1, %0\n v_add_u32 %2, %2, %1\n v_sub_u32_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_pair_sdwa_d3, "v_ashrrev_i32 %0, 1, %0\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_sub_u32_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_mad_mul, "v_mad_i32_i24 %0, %0, %1, %2\n v_mul_i32_i24 %0, %0, %1\n")
KERNEL(k_mad_mul_sdwa, "v_mad_i32_i24 %0, %0, %1, %2\n v_mul_i32_i24_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_add_cvtpk, "v_add_u32 %0, %0, %1\n v_cvt_pk_i16_i32 %0, %1, %0\n")
KERNEL(k_add_med3, "v_add_u32 %0, %0, %1\n v_med3_i32 %0, %0, %1, %2\n")
KERNEL(k_or3_dep, "v_or3_b32 %0, %0, %1, %2\n")
KERNEL(k_mul24_after_perm, "v_perm_b32 %0, %0, %1, %2\n v_mul_i32_i24 %0, %0, %1\n")
KERNEL(k_xor_after_perm, "v_perm_b32 %0, %0, %1, %2\n v_xor_b32 %0, %0, %1\n")

struct T { const char *name; void (*k)(unsigned long long *, int *, int); int per_body; };

int main() {
    unsigned long long *out; int *sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096 * 4);
    T tests[] = {
        {"v_add_u32 dep", k_add_dep, 1}, {"v_add_u32 4-indep", k_add_ind, 4},
        {"v_mad_i32_i24 dep", k_mad24_dep, 1}, {"v_mad_i32_i24 4-indep", k_mad24_ind, 4}, {"v_mul_i32_i24 dep", k_mul24_dep, 1},
        {"v_mul_lo_u32 dep", k_mullo_dep, 1}, {"v_mul_lo_u32 4-indep", k_mullo_ind, 4},
        {"v_med3_i32 dep", k_med3_dep, 1}, {"v_perm_b32 dep", k_perm_dep, 1}, {"v_add3_u32 dep", k_add3_dep, 1},
        {"v_ashrrev dep", k_ashr_dep, 1}, {"v_xor dep", k_xor_dep, 1}, {"s_nop1+v_min_u32_dpp dep", k_dpp_min_dep, 2},
        {"v_add_u32_dpp 2-indep", k_dpp_add_ind, 2}, {"v_mad_u64_u32", k_madu64, 1},
        {"v_bfe_i32 dep", k_bfe_dep, 1}, {"v_lshl_or dep", k_lshlor_dep, 1}, {"v_cndmask dep", k_cndmask_dep, 1}, {"v_cmp+v_cndmask", k_cmp_cnd, 2},
        {"v_pk_add_u16 dep", k_pk_add_dep, 1}, {"v_pk_mad_i16 dep", k_pk_mad_dep, 1}, {"v_pk_mul_lo_u16 dep", k_pk_mul_dep, 1},
        {"v_dot2_i32_i16 dep", k_dot2_dep, 1}, {"v_dot2_i32_i16 4-indep", k_dot2_ind, 4}, {"v_dot4_i32_i8 dep", k_dot4_dep, 1},
        {"ds_bpermute+wait", k_bperm, 1}, {"ds_bpermute 4-pipe", k_bperm_pipe, 4}, {"v_readlane+nop+add", k_readlane, 3},
        {"v_fma_f32 dep", k_fma_dep, 1}, {"v_fma_f32 4-indep", k_fma_ind, 4}, {"s_nop 0", k_snop, 1}, {"s_add_u32", k_salu, 1}, {"v_add+s_add", k_mix_valu_salu, 2},
        {"v_add_u32_e64 4-indep", k_add_e64_ind, 4}, {"v_or_b32 literal 4-indep", k_or_lit_ind, 4}, {"add/mad24 alternating", k_mix_23, 4},
        {"v_dot2c (VOP2) 4-indep", k_dot2c_ind, 4}, {"2 add + dpp (hazard filled)", k_dpp_filled, 3}, {"v_med3 4-indep", k_med3_ind, 4},
        {"s_nop 1", k_snop1, 1}, {"v_mov_b32 4-indep", k_mov_ind, 4},
        {"run 2 add + 2 mad24", k_run_2_2, 4}, {"run 4 add + 4 mad24", k_run_4_4, 8}, {"run 8 add + 8 mad24", k_run_8_8, 16},
        {"run 16 add + 16 mad24", k_run_16_16, 32}, {"run 12 add + 4 mad24", k_run_12_4, 16}, {"3 add + 1 mad24", k_run_3_1, 4}, {"1 add + 3 mad24", k_run_1_3, 4},
        {"v_max/min_i32 4-indep", k_maxmin, 4}, {"sub/and/lshl/or 4-indep", k_sub_and, 4}, {"v_alignbit 4-indep", k_alignbit, 4},
        {"v_lshl_add_u32 4-indep", k_lshl_add, 4}, {"v_add_u32_sdwa 4-indep", k_sdwa_add, 4}, {"v_pk_ashrrev_i16 4-indep", k_pk_ashr, 4},
        {"v_bfe_u32 4-indep", k_bfe_u, 4},
        {"ashr -> sub (dep pair)", k_pair_plain, 2}, {"ashr -> sub_sdwa (src1 fresh)", k_pair_sdwa, 2},
        {"ashr -> sub_sdwa (sel src fresh)", k_pair_sdwa_src, 2}, {"ashr, add, sub_sdwa (dist 2)", k_pair_sdwa_d2, 3},
        {"ashr, add, add, sub_sdwa (d 3)", k_pair_sdwa_d3, 4},
        {"mad24 -> mul24", k_mad_mul, 2}, {"mad24 -> mul24_sdwa", k_mad_mul_sdwa, 2}, {"add -> cvt_pk_i16_i32", k_add_cvtpk, 2},
        {"add -> med3", k_add_med3, 2}, {"v_or3 dep", k_or3_dep, 1}, {"perm -> mul24", k_mul24_after_perm, 2}, {"perm -> xor", k_xor_after_perm, 2},
        {"v_sad_u32 dep", k_sad, 1}, {"v_xad_u32 dep", k_xad, 1}, {"v_mul_hi_i32_i24 dep", k_mul_u24_hi, 1},
    };
    printf("%-28s %12s %12s %12s   (cycles per instruction, s_memtime ticks)\n", "test", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (auto &t : tests) {
        double res[3];
        int wv[3] = {4, 8, 16};
        for (int m = 0; m < 3; m++) {
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(t.k, dim3(1), dim3(64 * wv[m]), 0, 0, out, sink, r);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(wv[m]);
            hipMemcpy(h.data(), out, 8 * wv[m], hipMemcpyDeviceToHost);
            unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
            res[m] = (double)mx / (REP * 32.0 * t.per_body);
        }
        printf("%-28s %12.2f %12.2f %12.2f\n", t.name, res[0], res[1], res[2]);
    }
    return 0;
}
