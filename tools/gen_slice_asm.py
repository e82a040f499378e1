#!/usr/bin/env python3
"""Generates qoaudio_amd/csrc/qoa_slice_asm.inc (run: python tools/gen_slice_asm.py): K1's slice evaluation as inline-asm
texts.  Writing whole slices -- and whole 16-slice blocks -- as ONE statement removes the s_nop / s_waitcnt pads hipcc
puts at every asm-statement boundary, lets the history / sign rings rotate by register NAME (20 % 4 == 0: after a slice
every ring register again holds the element it started with) and puts every wait state under the author's control.

  QOA_SLICE_ASM_TEXT       20 x step(): 41 instructions, full 32-bit weights (the same sequence as cand_step() in
                           qoa_kernels.hip).  Scratch: v[96:115] samples, v[116:119] dot-product pairs, v120-v123
                           temporaries, v124-v127 the +-1 sign ring.
  QOA_SLICE_PK_ASM_TEXT    20 x pk_step(): 35 instructions on i16-packed weights / history, one slice.
  QOA_BLOCK_PK_ASM_TEXT    pk_block(): a loop over up to 16 slices of pk_step() incl. winner election and LDS exchange;
                           the straight-line body is ordered by schedule().
  QOA_BLOCK_SPEC_ASM_TEXT  pk_block(spec=True): the same loop on speculative steps (26 / 29 instructions; i16 sample pairs,
                           saturating 32-bit rank formed two steps at a time, v_cvt_pk_i16_i32 for the history pair).
  QOA_BLOCK_SPEC_LDSQ_ASM_TEXT  ... with the quantiser / dequantiser in an LDS table (pk_step_ldsq).
  QOA_BLOCK_CHEAP_ASM_TEXT pk_block(spec=True, cheap=True): speculative steps with the cheap quotient (22 / 25 instructions,
                           cheap_quotient_note()).
All forms read a slice's samples from its 48-byte LDS tile record (sample_reads()).
"""
import os

SMP0, P, Q, T, U, SNEG, D, S0 = 96, 116, 118, 120, 121, 122, 123, 124
# The LDS tile holds a slice's 20 samples as 10 i16 pairs (48-byte stride); they are read into v[SPK0:SPK0+9].  The
# speculative forms use the pairs as they are (SDWA word selects, packed error terms); the others widen them into
# v[SMP0:SMP0+19] first (unpack_samples: the pair registers are the upper half of that range, widened in place).
SPK0 = SMP0 + 10
RANK_EXACT_BELOW = 32767 * 32767   # saturating 32-bit rank of the speculative forms: below this value it is the true rank


STEP_OPERANDS = dict(h=[f"%[h{j}]" for j in range(4)], w=[f"%[w{j}]" for j in range(4)], acca="%[acca]", accb="%[accb]",
                     e="%[e]", pen="%[pen]", aerr0="%[aerr0]")
# the 32-bit block loop (b32_block) keeps the state in fixed registers: v[84:87] history, v[88:91] weights = the two halves of
# a candidate's 32-byte LDS slot
B32_FIXED = dict(h=[f"v{84 + j}" for j in range(4)], w=[f"v{88 + j}" for j in range(4)], acca="v77", accb="v78",
                 e="v[82:83]", pen="v76", aerr0="v79")


def step(i, N=STEP_OPERANDS):
    b = i % 4
    h = [N["h"][(b + j) % 4] for j in range(4)]            # oldest .. newest
    s = [f"v{S0 + (b + j) % 4}" for j in range(4)]
    w = N["w"]
    smp = f"v{SMP0 + i}"
    acc = N["acca"] if i < 10 else N["accb"]
    p, q = f"v[{P}:{P + 1}]", f"v[{Q}:{Q + 1}]"
    if N.get("pair_pen"):
        return step_pair_pen(i, N, h, s, w, smp, acc)
    L = [
        f"v_mul_lo_u32 v{P}, {w[0]}, {h[0]}",
        f"v_mul_lo_u32 v{Q}, {w[0]}, {w[0]}",
        f"v_mad_u64_u32 {p}, vcc, {w[1]}, {h[1]}, {p}",
        f"v_mad_u64_u32 {q}, vcc, {w[1]}, {w[1]}, {q}",
        f"v_mad_u64_u32 {p}, vcc, {w[2]}, {h[2]}, {p}",
        f"v_mad_u64_u32 {q}, vcc, {w[2]}, {w[2]}, {q}",
        f"v_mad_u64_u32 {p}, vcc, {w[3]}, {h[3]}, {p}",
        f"v_mad_u64_u32 {q}, vcc, {w[3]}, {w[3]}, {q}",
        f"v_ashrrev_i32 v{P}, 13, v{P}",
        f"v_ashrrev_i32 v{Q}, 18, v{Q}",
        f"v_sub_u32 v{T}, {smp}, v{P}",
        f"v_max_i32 v{Q}, 0x8ff, v{Q}",
        f"v_mad_i32_i24 v{U}, v{T}, %[recip], %[rnd]",
        f"v_add_u32 v{Q}, 0xfffff701, v{Q}",
        f"v_med3_i32 v{T}, v{T}, -1, 1",
        f"v_ashrrev_i32 v{U}, 16, v{U}",
        f"v_mad_i32_i24 {N['pen']}, v{Q}, v{Q}, {N['pen']}",
        f"v_lshl_add_u32 v{SNEG}, v{T}, 1, v{U}",           # n + 2 sgn(residual)
        f"v_med3_i32 v{U}, v{U}, v{SNEG}, v{T}",            # scaled = n + sgn(residual) - sgn(n), see cand_step()
        f"v_sub_u32 v{T}, 0, v{U}",
        f"v_ashrrev_i32 v{SNEG}, 31, v{U}",
        f"v_max_i32 v{T}, v{U}, v{T}",
        f"v_min_i32 v{T}, 7, v{T}",
        f"v_and_b32 v{T}, 6, v{T}",
        f"v_mad_u32_u24 v{D}, v{T}, %[c101], %[selb]",
        f"v_lshl_add_u32 {acc}, {acc}, 3, v{T}",
        f"v_perm_b32 v{D}, %[dqhi], %[dqlo], v{D}",
        f"v_sub_u32 {acc}, {acc}, v{SNEG}",
        f"v_xor_b32 v{D}, v{D}, v{SNEG}",
        f"v_sub_u32 v{D}, v{D}, v{SNEG}",
        f"v_add_u32 v{U}, v{D}, v{P}",
        f"v_ashrrev_i32 v{D}, 4, v{D}",
        f"v_med3_i32 {h[0]}, v{U}, %[lo], %[hi]",          # reconstructed -> the oldest history slot = new newest
    ]
    # NOTE: h[0] (oldest) was last read by the first v_mul_lo_u32 above; s[0] is last read by the update below.
    L += [
        f"v_mad_i32_i24 {w[0]}, {s[0]}, v{D}, {w[0]}",
        f"v_sub_u32 v{T}, {smp}, {h[0]}",                   # err
        f"v_mad_i32_i24 {w[1]}, {s[1]}, v{D}, {w[1]}",
        f"v_ashrrev_i32 {s[0]}, 31, {h[0]}",                # sign of the new sample -> the oldest sign slot
        f"v_mad_i32_i24 {w[2]}, {s[2]}, v{D}, {w[2]}",
        f"v_mad_i64_i32 {N['e']}, vcc, v{T}, v{T}, {N['e']}",
        f"v_mad_i32_i24 {w[3]}, {s[3]}, v{D}, {w[3]}",
        f"v_or_b32 {s[0]}, 1, {s[0]}",
    ]
    assert len(L) == 41
    if i == 0:  # |err| of the first sample orders the first-sample ranks
        L += [f"v_sub_u32 v{U}, 0, v{T}", f"v_max_i32 {N['aerr0']}, v{T}, v{U}"]
    return L


def step_pair_pen(i, N, h, s, w, smp, acc):
    """The 32-bit step of the block loop with the weights penalty squared two steps at a time (round 4): the energy of an even
    step stays in v118, the odd step's goes to v92; behind it v_perm_b32 puts their high words side by side ((E >> 16) as
    i16), v_pk_ashrrev_i16 .., 2 makes E >> 18 of each, v_pk_max_i16 / v_pk_sub_i16 with 0x08ff08ff the reference's
    max(E >> 18, 0x8ff) - 0x8ff (the energy is NOT biased beforehand as in the packed forms: this form must reproduce the
    reference for every state, including energies that wrap the i32), and one v_dot2_i32_i16 squares and adds both: 5
    instructions per step pair where shift, max, add and v_mad per step were 8; with the sign multiplications below 38.5
    instead of 41 per step.  The five are
    spread over the odd step so that no packed result is read by the instruction right behind it."""
    QE, QO, PP, A1 = 118, 92, "v94", "v95"
    first = i in (0, 10)
    Qr = QO if i % 2 else QE
    p, q = f"v[{P}:{P + 1}]", f"v[{Qr}:{Qr + 1}]"
    odd = i % 2 == 1
    pen = N["pen"]
    L = [
        f"v_mul_lo_u32 v{P}, {w[0]}, {h[0]}",
        f"v_mul_lo_u32 v{Qr}, {w[0]}, {w[0]}",
        f"v_mad_u64_u32 {p}, vcc, {w[1]}, {h[1]}, {p}",
        f"v_mad_u64_u32 {q}, vcc, {w[1]}, {w[1]}, {q}",
        f"v_mad_u64_u32 {p}, vcc, {w[2]}, {h[2]}, {p}",
        f"v_mad_u64_u32 {q}, vcc, {w[2]}, {w[2]}, {q}",
        f"v_mad_u64_u32 {p}, vcc, {w[3]}, {h[3]}, {p}",
        f"v_mad_u64_u32 {q}, vcc, {w[3]}, {w[3]}, {q}",
        f"v_ashrrev_i32 v{P}, 13, v{P}",
        f"v_perm_b32 {PP}, v{QO}, v{QE}, %[hiw]" if odd else None,
        f"v_sub_u32 v{T}, {smp}, v{P}",
        f"v_pk_ashrrev_i16 {PP}, 2, {PP} op_sel_hi:[0,1]" if odd else None,
        f"v_mad_i32_i24 v{U}, v{T}, %[recip], %[rnd]",
        f"v_pk_max_i16 {PP}, {PP}, %[c8ff2]" if odd else None,
        f"v_med3_i32 v{T}, v{T}, -1, 1",
        f"v_pk_sub_i16 {PP}, {PP}, %[c8ff2]" if odd else None,
        f"v_ashrrev_i32 v{U}, 16, v{U}",
        f"v_lshl_add_u32 v{SNEG}, v{T}, 1, v{U}",           # n + 2 sgn(residual)
        f"v_dot2_i32_i16 {pen}, {PP}, {PP}, {0 if i == 1 else pen}" if odd else None,
        f"v_med3_i32 v{U}, v{U}, v{SNEG}, v{T}",            # scaled = n + sgn(residual) - sgn(n), see cand_step()
        # |scaled| and, further down, the dequantised value's sign as multiplications by a1 = +-1 (as the speculative forms do:
        # one instruction less than negate + max and xor + subtract)
        f"v_ashrrev_i32 v{SNEG}, 31, v{U}",
        f"v_or_b32 {A1}, 1, v{SNEG}",
        f"v_mul_i32_i24 v{T}, v{U}, {A1}",                   # |scaled| (< 2^16)
        f"v_min_i32 v{T}, 7, v{T}",
        f"v_and_b32 v{T}, 6, v{T}",
        f"v_mad_u32_u24 v{D}, v{T}, %[c101], %[selb]",
        None if first else f"v_lshl_add_u32 {acc}, {acc}, 3, v{T}",   # (the first code of a slice-word half needs no shift)
        f"v_perm_b32 v{D}, %[dqhi], %[dqlo], v{D}",
        f"v_sub_u32 {acc}, {('v%d' % T) if first else acc}, v{SNEG}",
        f"v_mul_i32_i24 v{D}, v{D}, {A1}",                   # dequantised
        f"v_add_u32 v{U}, v{D}, v{P}",
        f"v_ashrrev_i32 v{D}, 4, v{D}",
        f"v_med3_i32 {h[0]}, v{U}, %[lo], %[hi]",          # reconstructed -> the oldest history slot = new newest
        f"v_mad_i32_i24 {w[0]}, {s[0]}, v{D}, {w[0]}",
        f"v_sub_u32 v{T}, {smp}, {h[0]}",                   # err
        f"v_mad_i32_i24 {w[1]}, {s[1]}, v{D}, {w[1]}",
        f"v_ashrrev_i32 {s[0]}, 31, {h[0]}",                # sign of the new sample -> the oldest sign slot
        f"v_mad_i32_i24 {w[2]}, {s[2]}, v{D}, {w[2]}",
        f"v_mad_i64_i32 {N['e']}, vcc, v{T}, v{T}, {0 if i == 0 else N['e']}",   # (the first term starts the rank)
        f"v_mad_i32_i24 {w[3]}, {s[3]}, v{D}, {w[3]}",
        f"v_or_b32 {s[0]}, 1, {s[0]}",
    ]
    L = [l for l in L if l is not None]
    assert len(L) == (41 if odd else 36) - (1 if first else 0)
    if i == 0:  # |err| of the first sample orders the first-sample ranks
        L += [f"v_sub_u32 v{U}, 0, v{T}", f"v_max_i32 {N['aerr0']}, v{T}, v{U}"]
    return L


def b32_block():
    """Up to %[n] consecutive full slices on the 32-bit step (step(): 41 instructions, full i32 weights -- it cannot decline)
    in one statement: the block loop's last resort for chains whose state, or whose LOSING candidates, leave the i16 window
    slice after slice (clipped material).  Before round 3 such slices went one by one through run_slice_asm + ten
    ds_bpermute + a re-admission test, after a wasted packed attempt: 2,100 issue slots per slice against ~900 here.
    State: v[84:87] history, v[88:91] weights; every candidate parks both halves in its 32-byte LDS slot, the election is the
    full form's (keys rank >> 5, then rank[6:0] : |err0| : sf << 5), all 16 lanes read the winner's slot back."""
    N = dict(B32_FIXED, pair_pen=True)
    K1, M, K2, ADDR, OADDR = "v116", "v117", "v118", "v72", "v120"   # K1/M/K2 share the dot-product registers (dead by then)
    TCUR, OCUR = "v74", "v75"
    SHI, SLO, ELO, EHI = "v80", "v81", "v82", "v83"
    reads = sample_reads(TCUR)
    dpp = lambda r, n: f"v_min_u32_dpp {r}, {r}, {r} row_ror:{n} row_mask:0xf bank_mask:0xf"
    L = [
        f"v_mov_b32 {TCUR}, %[tile]",
        f"v_subrev_u32 {OCUR}, 8, %[outb]",
    ] + sample_reads("%[tile]") + [
        "s_mov_b32 %[done], 0",
        "Lqoa_top_%=:",
        "s_waitcnt lgkmcnt(0)",                              # the slice's samples, the winner's state (a slice-word store may trail: harmless)
    ]
    for j in range(4):
        L += [f"v_ashrrev_i32 v{S0 + j}, 31, {N['h'][j]}", f"v_or_b32 v{S0 + j}, 1, v{S0 + j}"]
    # (no per-slice clears: the rank, the penalty sum and the slice word's halves all start with their first term)
    L += unpack_samples()
    for i in range(20):
        L += step(i, N)
    # (round 4: keys, slice-word halves, the candidates' state writes, sample reads and cursors all sit in the wait-state gaps of
    # the two DPP reductions -- two instructions between a DPP read and the VALU write of its source; one s_nop is left)
    L += [
        f"v_mad_u64_u32 {N['e']}, vcc, {N['pen']}, 1, {N['e']}",   # total rank = sum err^2 + sum penalty^2
        f"v_alignbit_b32 {K1}, {EHI}, {ELO}, 5",
        f"v_lshlrev_b32 {K2}, 25, {ELO}",
        f"v_lshl_or_b32 {K2}, {N['aerr0']}, 9, {K2}",
        f"v_min_u32_dpp {M}, {K1}, {K1} row_ror:8 row_mask:0xf bank_mask:0xf",
        f"v_or_b32 {K2}, %[sf4], {K2}",
        f"v_lshrrev_b32 {SHI}, 2, {N['acca']}",
        dpp(M, 4),
        f"v_or_b32 {SHI}, %[sfsh], {SHI}",
        f"v_add_u32 {TCUR}, 48, {TCUR}",
        dpp(M, 2),
        reads[0],
        f"v_lshl_or_b32 {SLO}, {N['acca']}, 30, {N['accb']}",
        dpp(M, 1),
        f"v_cmp_eq_u32 vcc, {K1}, {M}",
        f"v_cndmask_b32 {K2}, -1, {K2}, vcc",
        "ds_write_b128 %[cand], v[84:87]",
        "ds_write_b128 %[cand], v[88:91] offset:16",
        dpp(K2, 8),
        reads[1],
        reads[2],
        dpp(K2, 4),
        f"v_add_u32 {OCUR}, 8, {OCUR}",
        "s_add_u32 %[done], %[done], 1",
        dpp(K2, 2),
        "s_cmp_lt_u32 %[done], %[n]",
        "s_nop 0",
        dpp(K2, 1),
        f"v_bfi_b32 {ADDR}, %[mf0], {K2}, %[cand]",          # the winner's slot
        f"ds_read_b128 v[84:87], {ADDR}",
        f"ds_read_b128 v[88:91], {ADDR} offset:16",
        f"v_cmp_eq_u32 vcc, {ADDR}, %[cand]",
        f"v_cndmask_b32 {OADDR}, %[dummy], {OCUR}, vcc",      # the winner stages its slice, the others hit a dummy
        f"ds_write_b64 {OADDR}, v[80:81]",
        "s_cbranch_scc1 Lqoa_top_%=",
        "s_waitcnt lgkmcnt(0)",
    ]
    return L


# ---------------------------------------------------------------------------------------------------------------
# Packed fast path: weights and history travel as i16 pairs, so the two 4-term dot products are 2 + 2 v_dot2_i32_i16
# and the 4 weight updates are 2 v_pk_mad_i16 -- 35 instructions per step.  Valid only while every weight fits i16
# and the weight energy stays below the bound the kernel checks after the slice (qoa_kernels.hip: kPkPenLimit).
#
# Pair ring: B_n = (h[n-2], h[n-1]) lives in R[n % 4]; step n reads A_n = B_(n-2) and B_n and writes
# B_(n+1) = alignbit(recon_n, B_n, 16).  Sign ring S[] alike.  R0 = %[pb], R2 = %[pa] are the asm's in/out operands:
# in (h2,h3) / (h0,h1), out (h18,h19) / (h16,h17) of the slice.
#
# Hazards the assembler does not pad inside an asm statement (gfx940/950): a v_dot2 result needs 3 other instructions
# before a different VALU reads it; a packed (v_pk_*) result needs 1.  The slots are filled with work deferred from
# the previous step (its error term) and with the sign pair of B_n.
PK_P, PK_Q, PK_PR, PK_T, PK_U, PK_SN, PK_D, PK_X, PK_E = 116, 117, 118, 119, 120, 121, 122, 123, 91
PK_R = ["%[pb]", "v124", "%[pa]", "v125"]
PK_S = ["v126", "v127", "v94", "v95"]
PK_RC = ["v92", "v93"]


PK_OPERANDS = dict(w01="%[w01]", w23="%[w23]", pen="%[pen]", e="%[e]", acca="%[acca]", accb="%[accb]", aerr0="%[aerr0]",
                   R=PK_R)
# The block loop keeps everything that lives across slices in fixed registers (the statement's in/out operands are
# register variables bound to v[84:87]): v[80:81] slice words, v[82:83] rank, v[84:87] packed state.
PK_FIXED = dict(w01="v84", w23="v85", pen="v88", e="v[82:83]", acca="v89", accb="v90", aerr0="v79",
                R=["v87", "v124", "v86", "v125"], alt=[f"v{r}" for r in range(67, 76)], qm="v66",
                one2="s95")  # 0x00010001 in an SGPR: a 4-byte v_or instead of 8 (a lone wave runs close to its fetch rate)


def pk_step(i, N=PK_OPERANDS, spec=False):
    R, S, RC = N["R"], PK_S, N.get("RC", PK_RC)
    A, B, Bn = R[(i + 2) % 4], R[i % 4], R[(i + 1) % 4]
    SA, SB = S[(i + 2) % 4], S[i % 4]
    smp = f"v{SMP0 + i}"
    acc = N["acca"] if i < 10 else N["accb"]
    w01, w23, pen, e = N["w01"], N["w23"], N["pen"], N["e"]
    rc = RC[i % 2]
    P, Q, PR, T, U, SN, D, X, E = (f"v{r}" for r in (PK_P, PK_Q, PK_PR, PK_T, PK_U, PK_SN, PK_D, PK_X, PK_E))
    if N.get("alt") and i % 2:  # second temporary set: lets the scheduler overlap neighbouring steps
        P, Q, PR, T, U, SN, D, X, E = N["alt"]
    sub_t = f"v_sub_u32 {T}, {smp}, {PR}"                  # residual
    if i == 0 and N.get("acc_cleared"):                    # (the block loop clears the slice-word halves in its election's gaps)
        slot_a, slot_b = None, None
    elif i == 0:
        slot_a, slot_b = f"v_mov_b32 {N['acca']}, 0" + N.get("fill", ""), f"v_mov_b32 {N['accb']}, 0" + N.get("fill", "")
    elif spec and N.get("wide"):
        # wide form: the true 64-bit rank, one error per step -- sample i - 1 (a half of its pair register) minus its clamped
        # reconstruction, the high half of the newest history pair B
        slot_a = (f"v_sub_u32_sdwa {E}, sext(v{SPK0 + (i - 1) // 2}), sext({B}) dst_sel:DWORD dst_unused:UNUSED_PAD "
                  f"src0_sel:WORD_{(i - 1) % 2} src1_sel:WORD_1")
        slot_b = f"v_mad_i64_i32 {e}, vcc, {E}, {E}, {0 if i == 1 else e}" + N.get("fill", "")   # (the first term starts the sum)
    elif spec:
        if N.get("alt") and i % 4 == 0:
            E = N["alt"][8]                                  # error pairs alternate between two registers
        slot_a, slot_b = spec_rank_pair(i, N, B, E) if i % 2 == 0 else (None, None)
    else:
        slot_a = f"v_sub_u32 {E}, v{SMP0 + i - 1}, {RC[(i - 1) % 2]}"      # err of the previous step
        slot_b = f"v_mad_i64_i32 {e}, vcc, {E}, {E}, {e}"
    if spec:
        sub_t = spec_residual(i, T, PR)
    # Speculative form (spec): no penalty arithmetic at all; the weight energy is only sampled, at every 4th step, into a
    # running maximum that the block loop tests after the slice (see pk_block).  Its samples stay i16 pairs and its rank is
    # a saturating 32-bit sum formed two steps at a time (spec_rank_pair).
    # Penalty form (N["pen_steps"], round 3): the cheap-quotient speculative step WITH the penalty arithmetic of the full form
    # -- for the stretches where the speculation's premise (a penalty-free winner) does not hold: see pk_block.
    pen_steps = bool(N.get("pen_steps"))
    pen_full = not spec or pen_steps
    pair_pen = pen_steps or (not spec and bool(N.get("alt")))   # penalties squared two steps at a time (two temporary sets needed)
    # (until late in round 4 the wide form sampled the energy at every step; it was not the margins that made noise miss on the
    # other forms -- with samples at every 4th step and their margins the wide form declines 0.025 % of white noise's slices)
    check = spec and not pen_steps and i % 4 == 0
    cheap = bool(N.get("cheap"))
    acc_first = bool(N.get("acc_cleared")) and spec and i in (0, 10)
    A1 = CH_A1[i % 2]
    PT = X if not pen_steps else PEN_TMP[i % 2]            # penalty temporary (the cheap steps keep X for their flag OR)
    L = [
        f"v_dot2_i32_i16 {P}, {w01}, {A}, 0",
        f"v_dot2_i32_i16 {Q}, {w01}, {w01}, %[negc]" if pen_full or check else None,   # sum w^2 - (0x8ff << 18)
        f"v_dot2_i32_i16 {P}, {w23}, {B}, {P}",
        # (a sampling step's energy goes straight into the running maximum at step 0: no copy)
        (f"v_dot2_i32_i16 {N['qm'] if check and i == 0 else Q}, {w23}, {w23}, {Q}") if pen_full or check else None,
        f"v_pk_ashrrev_i16 {SB}, 15, {B} op_sel_hi:[0,1]",   # the inline 15 has no high half to select
        slot_a,
        f"v_or_b32 {SB}, {N.get('one2', '0x10001')}, {SB}",  # (+-1, +-1) for B_n
        f"v_ashrrev_i32 {PR}, 13, {P}",                     # predicted
        f"v_ashrrev_i32 {PT}, 18, {Q}" if pen_full and not pair_pen else None,
        sub_t,
        slot_b,
        f"v_max_i32 {PT}, 0, {PT}" if pen_full and not pair_pen else None,   # penalty
        f"v_mad_i32_i24 {U}, {T}, %[recip], %[rnd]",
        f"v_med3_i32 {T}, {T}, -1, 1" if not cheap else f"v_ashrrev_i32 {SN}, 31, {T}",
        f"v_mad_i32_i24 {pen}, {PT}, {PT}, {pen}" if pen_full and not pair_pen else None,
        (None if i == 0 else f"v_max_i32 {N['qm']}, {N['qm']}, {Q}") if check else None,
        f"v_ashrrev_i32 {U}, 16, {U}" if not cheap else None,
    ] + ([
        f"v_lshl_add_u32 {SN}, {T}, 1, {U}",
        f"v_med3_i32 {U}, {U}, {SN}, {T}",                  # scaled
        f"v_sub_u32 {T}, 0, {U}",
        f"v_ashrrev_i32 {SN}, 31, {U}",
        f"v_max_i32 {T}, {U}, {T}",
        f"v_min_i32 {T}, 7, {T}",
        f"v_and_b32 {T}, 6, {T}",
    ] if not spec else [
        # speculative form, exact quotient: |scaled| and the dequantised value's sign as multiplications by a1 = +-1
        f"v_lshl_add_u32 {SN}, {T}, 1, {U}",
        f"v_med3_i32 {U}, {U}, {SN}, {T}",                  # scaled
        f"v_ashrrev_i32 {SN}, 31, {U}",
        f"v_or_b32 {A1}, 1, {SN}",
        f"v_mul_i32_i24 {T}, {U}, {A1}",                    # |scaled| (< 2^17)
        f"v_min_i32 {T}, 7, {T}",
        f"v_and_b32 {T}, 6, {T}",
    ] if not cheap else [
        # Cheap quotient (cheap_quotient_note): level from |n|, sign from the residual.  q = (n ^ s) - s is |n| whenever
        # n and the residual do not have opposite signs, and negative when they do -- the flag pk_block looks for.
        # Both sign changes are multiplications by a1 = +-1 (= s | 1): q = n * a1, with n = product >> 16 taken by the SDWA
        # word select, and further down dequantised = magnitude * a1.
        f"v_or_b32 {A1}, 1, {SN}",
        f"v_mul_i32_i24_sdwa {U}, sext({U}), {A1} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
        f"v_med3_i32 {T}, {U}, -1, 7",                      # clamp(q, -1, 7): a negative q -- the lane is marked -- becomes all ones
        f"v_and_b32 {T}, -10, {T}",                         # 2 * level; marked: 0xfffffff6
    ]) + [
        f"v_mad_u32_u24 {D}, {T}, %[c101], %[selb]",
        # cheap forms: OR instead of add -- the same for a level, and a marked step sets bits 31..4 of the slice-word half: bit
        # 31 then stays set through the <= 9 shifts that follow (the run of ones always reaches it), which is the MARK the
        # election reads (round 3 OR-ed the steps' clamped quotients together: one v_or3 per step pair, 10 per slice)
        # (acc_first: the first code of a slice-word half needs no shift -- and the half no clearing: code = 2 * level - sign in
        # ONE instruction at steps 0 and 10, round 4)
        None if acc_first else (f"v_lshl_or_b32 {acc}, {acc}, 3, {T}" if cheap else f"v_lshl_add_u32 {acc}, {acc}, 3, {T}"),
        f"v_perm_b32 {D}, %[dqhi], %[dqlo], {D}",
        # (step 0's is tagged as a filler: it takes the wait state behind step 1's v_dot2 pair, which has no rank term to hold yet)
        f"v_sub_u32 {acc}, {T if acc_first else acc}, {SN}" + (N.get("fill", "") if acc_first and i == 0 else ""),
    ] + ([
        f"v_xor_b32 {D}, {D}, {SN}",
        f"v_sub_u32 {D}, {D}, {SN}",                        # dequantised
    ] if not spec else [
        f"v_mul_i32_i24 {D}, {D}, {A1}",                    # dequantised
    ]) + ([
        f"v_add_u32 {U}, {D}, {PR}",
        f"v_ashrrev_i32 {D}, 4, {D}",                       # delta
        f"v_med3_i32 {rc}, {U}, %[lo], %[hi]",              # reconstructed
        f"v_pk_mad_i16 {w01}, {SA}, {D}, {w01} op_sel_hi:[1,0,1]",
        f"v_pk_mad_i16 {w23}, {SB}, {D}, {w23} op_sel_hi:[1,0,1]",
        f"v_alignbit_b32 {Bn}, {rc}, {B}, 16",              # B_(n+1) = (h[n-1], recon)
    ] if not spec else [
        # speculative forms: the unclamped reconstruction goes to RC[i % 2]; B_(n+1) = (sat16(previous), sat16(this)) is one
        # v_cvt_pk_i16_i32 (clamping the previous value again gives the same i16) -- no v_med3 + v_alignbit
        f"v_add_u32 {rc}, {D}, {PR}",
        f"v_ashrrev_i32 {D}, 4, {D}",                       # delta
        f"v_pk_mad_i16 {w01}, {SA}, {D}, {w01} op_sel_hi:[1,0,1]",
        f"v_pk_mad_i16 {w23}, {SB}, {D}, {w23} op_sel_hi:[1,0,1]",
        f"v_cvt_pk_i16_i32 {Bn}, {RC[(i + 1) % 2]}, {rc}",
    ])
    L = [l for l in L if l is not None]
    if pair_pen and i % 2:
        # Penalty and full form of the block loop, two steps at a time (round 4): the steps' energies minus the limit sit in Q of the even step (main
        # temporaries) and of this one; their high words are (E - limit) >> 16 as i16, two more arithmetic shifts make the
        # reference's (E >> 18) - 0x8ff, and max(0, .)^2 of both goes into the sum with one v_dot2: 4 instructions per step pair
        # where v_ashrrev + v_max + v_mad per step were 6.  Each penalty is < 2^13, the sum of 20 squares < 2^31 as before.
        q_even = f"v{PK_Q}"
        k = max(j for j, l in enumerate(L) if l.startswith("v_dot2_i32_i16 " + Q + ",")) + 1
        L[k:k] = [
            f"v_perm_b32 {PT}, {Q}, {q_even}, %[hiw]",
            f"v_pk_ashrrev_i16 {PT}, 2, {PT} op_sel_hi:[0,1]",
            f"v_pk_max_i16 {PT}, {PT}, 0 op_sel_hi:[1,0]",
            f"v_dot2_i32_i16 {pen}, {PT}, {PT}, {0 if i == 1 and pen_steps else pen}",   # (the full form clears its sum in the election)
        ]
    if not spec:
        assert len(L) == (35 if not pair_pen else 36 if i % 2 else 32)
        if i == 1:  # |err| of the first sample, right after slot_a produced it
            k = L.index(slot_a) + 1
            L[k:k] = [f"v_sub_u32 {D}, 0, {E}", f"v_max_i32 {N['aerr0']}, {E}, {D}"]
    elif N.get("wide"):
        assert len(L) == 26 + ((3 if i else 2) if check else 0) + (2 if i else 0) - (1 if acc_first else 0)
        if i == 1:  # |err| of the first sample, right after slot_a produced it
            k = L.index(slot_a) + 1
            L[k:k] = [f"v_sub_u32 {D}, 0, {E}", f"v_max_i32 {N['aerr0']}, {E}, {D}"]
    else:
        assert len(L) == (22 if cheap else 26) + ((2 if i == 0 else 3) if check else 0) + ((6 if i % 2 else 2) if pen_steps else 0) + (2 if i % 2 == 0 else 0) - (
            2 if i == 0 and N.get("acc_cleared") else 0) - (1 if acc_first else 0)
        if i == 2:
            k = L.index(slot_b) + 1
            L[k:k] = spec_aerr0(N, E, CH_TMP if cheap else X)
    return L


def cheap_quotient_note():
    """Why the cheap form is exact.  The reference's quantiser input is r = n + sgn(v) - sgn(n) (qoa_div, src/lib.rs:613-617)
    with n the rounded quotient in wrapping i32 arithmetic -- the steps compute the very same n.  Only min(|r|, 7) & 6 (the
    level) and the sign of r are used (src/lib.rs:569-571, QOA_QUANT_TAB).  Unless n and v have strictly opposite signs,
    |r| = max(|n|, [v != 0]) -- same level as |n|, because 0 and 1 share level 0 -- and r is negative exactly when v is.
    Opposite signs need the product to wrap, |v| * recip >= 2^31 - 2^15: only scalefactors 0 and 1 can (|v| <= 295 k with
    i16 weights and history), and then q < 0 marks the lane.  A marked candidate is taken out of the election; what is left
    to show is that it would have lost: up to the marked step its state is exact, and at that step |v| >= 32768 with a
    dequantised value of at most 49, so its true error there is at least min(32719, 32767 - max|sample of the slice|)
    (either the reconstruction is within 49 of the prediction, or it sits on the rail opposite to v).  The square of that
    bound comes with the slice's samples (BOUND_IN); the winner's rank must be below it, else the slice is declined."""


def spec_residual(i, T, PR):
    """sample i of the slice is half (i & 1) of pair register i >> 1"""
    return (f"v_sub_u32_sdwa {T}, sext(v{SPK0 + i // 2}), {PR} dst_sel:DWORD dst_unused:UNUSED_PAD "
            f"src0_sel:WORD_{i % 2} src1_sel:DWORD")


def spec_rank_pair(i, N, B, E):
    """Error terms of samples i-2, i-1 in two instructions: B (the newest history pair at step i) holds exactly their
    reconstructions, so (err, err) = sample pair - B, saturated to i16, and rank += err.err, saturated to i32.  Both
    saturations only ever make the sum SMALLER than the true rank but never below 32767^2 once they act, so a rank below
    that value is exact and any other is at least that large (pk_block tests the winner's)."""
    elo = N["e"].split(":")[0].replace("v[", "v")
    return (f"v_pk_sub_i16 {E}, v{SPK0 + i // 2 - 1}, {B} clamp" + N.get("fill", ""),
            # the cheap forms count the rank from 1: a marked candidate's key (its bound, see pk_block_spec) then lies strictly
            # below its true rank + 1, and a bound of 0 beats every unmarked candidate
            f"v_dot2_i32_i16 {elo}, {E}, {E}, {('1' if N.get('cheap') else '0') if i == 2 else elo} clamp" + N.get("fill", ""))


def spec_aerr0(N, E, X):
    """|err| of the first sample (orders the first-sample ranks) from the low half of the first error pair"""
    sd = "dst_sel:DWORD dst_unused:UNUSED_PAD"
    return [l + N.get("fill", "") for l in (f"v_sub_u32_sdwa {X}, 0, sext({E}) {sd} src0_sel:DWORD src1_sel:WORD_0",
                                            f"v_max_i32_sdwa {N['aerr0']}, sext({E}), {X} {sd} src0_sel:WORD_0 src1_sel:DWORD")]


# Speculative block loops: the part of a candidate's end state that the next slice's first steps would otherwise re-derive
# from the history pairs -- B_19 = (h17, h18), the unclamped last reconstruction (-> h19), the sign pairs of B_18 and B_19 --
# sits in v[92:95] and travels through the second half of the candidate's 32-byte LDS slot (pk_block).
SPEC_REGS = dict(R=["v87", "v124", "v86", "v92"], RC=["v102", "v93"])
SLOT2 = "v[92:95]"
ELO_REG = "v82"   # the block loops' rank register (PK_FIXED e = v[82:83]; the speculative forms use the low dword only)


def pk_step_ldsq(i, N, deferred=None):
    """Speculative step with the quantiser in an LDS table (BASELINE.json's "tables staged in LDS"): for batches that give
    every SIMD several wavefronts, where the ~100 cycles of the lookup are covered by the other wavefronts and what counts
    is the issue-slot total.  scaled -> clamp(-8, 8) -> entry [sf][c + 8] = dequant & 0xffff | code << 29 replaces the
    closed-form quantiser + v_perm dequant + sign fix-ups of pk_step(): 22 (25 at a sampling step) VALU + 1 LDS read
    instead of 30 (33).  Same values: QOA_QUANT_TAB / QOA_DEQUANT_TAB (src/lib.rs:22, :847-864) are what the closed form
    was derived from.  The order below keeps the v_dot2 / v_pk wait states by construction (not list-scheduled: the
    read's wait is a fence, and with other wavefronts filling the slots dependent issue costs nothing).
    Returns (lines, acc_line): the slice-word accumulate of this step is handed to the NEXT step (`deferred`), which issues
    it in the wait states behind its v_dot2 pair -- at odd steps nothing else is left to fill them."""
    R, S, RC = N["R"], PK_S, N.get("RC", PK_RC)
    A, B, Bn = R[(i + 2) % 4], R[i % 4], R[(i + 1) % 4]
    SA, SB = S[(i + 2) % 4], S[i % 4]
    smp = f"v{SMP0 + i}"
    acc = N["acca"] if i < 10 else N["accb"]
    w01, w23, e = N["w01"], N["w23"], N["e"]
    rc = RC[i % 2]
    P, Q, PR, T, U, SN, D, X, E = (f"v{r}" for r in (PK_P, PK_Q, PK_PR, PK_T, PK_U, PK_SN, PK_D, PK_X, PK_E))
    if i == 0:
        slot_a, slot_b = f"v_mov_b32 {N['acca']}, 0", f"v_mov_b32 {N['accb']}, 0"
    else:
        slot_a, slot_b = spec_rank_pair(i, N, B, E) if i % 2 == 0 else (None, None)
    check = i % 4 == 0
    L = [
        f"v_dot2_i32_i16 {P}, {w01}, {A}, 0",
        f"v_dot2_i32_i16 {Q}, {w01}, {w01}, %[negc]" if check else None,     # sum w^2 - (0x8ff << 18), sampled
        f"v_dot2_i32_i16 {P}, {w23}, {B}, {P}",
        f"v_dot2_i32_i16 {Q}, {w23}, {w23}, {Q}" if check else None,
        f"v_pk_ashrrev_i16 {SB}, 15, {B} op_sel_hi:[0,1]",
        slot_a,
        deferred,                                           # (also keeps the packed shift's result one slot away from the v_or)
        f"v_or_b32 {SB}, {N.get('one2', '0x10001')}, {SB}",
        f"v_ashrrev_i32 {PR}, 13, {P}",                     # predicted
        spec_residual(i, T, PR),
        slot_b,
        f"v_mad_i32_i24 {U}, {T}, %[recip], %[rnd]",
        f"v_med3_i32 {T}, {T}, -1, 1",
        (f"v_mov_b32 {N['qm']}, {Q}" if i == 0 else f"v_max_i32 {N['qm']}, {N['qm']}, {Q}") if check else None,
        f"v_ashrrev_i32 {U}, 16, {U}",
        f"v_lshl_add_u32 {SN}, {T}, 1, {U}",
        f"v_med3_i32 {U}, {U}, {SN}, {T}",                  # scaled (qoa_div, src/lib.rs:613-617)
        f"v_med3_i32 {U}, {U}, -8, 8",                      # :569
        f"v_lshl_add_u32 {X}, {U}, 2, %[qtab]",
        f"ds_read_b32 {X}, {X}",
        "s_waitcnt lgkmcnt(0)",
        f"v_add_u32_sdwa {rc}, {PR}, sext({X}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0",
        f"v_bfe_i32 {D}, {X}, 4, 12",                       # delta = dequantised >> 4 (:822)
        f"v_cvt_pk_i16_i32 {Bn}, {RC[(i + 1) % 2]}, {rc}",  # B_(n+1) = (h[n-1], recon), both clamped (see pk_step)
        f"v_pk_mad_i16 {w01}, {SA}, {D}, {w01} op_sel_hi:[1,0,1]",
    ] + ([f"v_pk_sub_i16 {E}, v{SPK0 + 9}, {Bn} clamp"] if i == 19 else []) + [   # the last error pair (spec_rank_pair)
        f"v_pk_mad_i16 {w23}, {SB}, {D}, {w23} op_sel_hi:[1,0,1]",
    ] + ([f"v_dot2_i32_i16 {ELO_REG}, {E}, {E}, {ELO_REG} clamp"] if i == 19 else [])
    L = [l for l in L if l is not None]
    if i == 2:
        k = L.index(slot_b) + 1
        L[k:k] = spec_aerr0(N, E, f"v{PK_D}")               # (temporary: D; X still holds step 1's table word)
    return L, f"v_alignbit_b32 {acc}, {acc}, {X}, 29"       # slice = slice << 3 | code (:584)


def pk_zero(N):
    return [f"v_mov_b64 {N['e']}, 0", f"v_mov_b32 {N['pen']}, 0"]


def pk_prologue(N, zero=True, spec=False):
    R, S = N["R"], PK_S
    return ([f"v_ashrrev_i32 {N.get('RC', PK_RC)[1]}, 16, {R[0]}"] if spec else []) + [   # h3: step 0's "previous reconstruction"
        f"v_alignbit_b32 {R[3]}, {R[0]}, {R[2]}, 16",        # B_(-1) = (h1, h2)
        f"v_pk_ashrrev_i16 {S[2]}, 15, {R[2]} op_sel_hi:[0,1]",
        f"v_pk_ashrrev_i16 {S[3]}, 15, {R[3]} op_sel_hi:[0,1]",
        f"v_or_b32 {S[2]}, {N.get('one2', '0x10001')}, {S[2]}",
        f"v_or_b32 {S[3]}, {N.get('one2', '0x10001')}, {S[3]}",
    ] + (pk_zero(N) if zero else [])


def sample_reads(addr):
    """a slice's tile record: dword 0 = BOUND_IN (see pk_block, cheap form), dword 1 unused, dwords 2..11 the sample pairs"""
    return [f"ds_read_b128 v[{SPK0 - 2}:{SPK0 + 1}], {addr} offset:0",
            f"ds_read_b128 v[{SPK0 + 2}:{SPK0 + 5}], {addr} offset:16",
            f"ds_read_b128 v[{SPK0 + 6}:{SPK0 + 9}], {addr} offset:32"]


BOUND_IN = f"v{SPK0 - 2}"
# extra fixed registers of the cheap-quotient form (v96..v103 are free in the speculative forms: no widened samples)
CH_W, CH_TMP, CH_BIG, CH_BND = "v96", "v97", "v98", "v99"
CH_A1 = ["v100", "v101"]   # +-1 of the residual's sign, per temporary set
PEN_TMP = ["v102", "v103"]  # penalty form: the penalty of a step, per temporary set


def unpack_samples():
    """i16 pairs v[SPK0 + k] -> sign-extended v[SMP0 + 2k], v[SMP0 + 2k + 1]; in place: pair k is read before the two
    registers written for it can reach an unread pair (SMP0 + 2k + 1 <= SPK0 + k for k <= 9)."""
    L = []
    for k in range(10):
        L += [f"v_bfe_i32 v{SMP0 + 2 * k}, v{SPK0 + k}, 0, 16", f"v_ashrrev_i32 v{SMP0 + 2 * k + 1}, 16, v{SPK0 + k}"]
    return L


def pk_slice():
    N = PK_OPERANDS
    lines = sample_reads("%[tile]") + pk_prologue(N) + ["s_waitcnt lgkmcnt(0)"] + unpack_samples()
    for i in range(20):
        lines += pk_step(i, N)
    lines += [f"v_sub_u32 v{PK_E}, v{SMP0 + 19}, {PK_RC[1]}", f"v_mad_i64_i32 {N['e']}, vcc, v{PK_E}, v{PK_E}, {N['e']}"]
    return lines


# ---------------------------------------------------------------------------------------------------------------
# List scheduler for a straight-line VALU region.  A lone wavefront issues an instruction that reads the result of
# the instruction right before it ~0.33 cycle late (tools/ubench: dependent chain 4.46 vs independent 4.14 cycles
# per instruction), so the region is re-ordered to put an independent instruction between producer and consumer
# wherever the dependence graph allows, under the hard wait-state rules of gfx940/950:
#   v_dot2 result -> read by anything but the same opcode's accumulator operand: 3 instructions between
#   v_dot2 result -> overwritten by another VALU: 4 between;  packed (v_pk_*) result -> any VALU read: 1 between.
import re

_MOD = re.compile(r"\b(op_sel_hi|op_sel|row_ror|row_mask|bank_mask|offset|dst_sel|dst_unused|src0_sel|src1_sel):(\[[^\]]*\]|\S+)"
                  r"|\bclamp\b")


def _regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"sext\((.*)\)", tok)
    if m:
        tok = m.group(1)
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return [f"v{k}" for k in range(int(m.group(1)), int(m.group(2)) + 1)]
    if re.fullmatch(r"v\d+", tok) or tok == "vcc" or tok.startswith("%["):
        return [tok]
    return []  # immediates


def _parse(line):
    op, rest = line.split(None, 1)
    ops = [t for t in _MOD.sub("", rest).split(",")]
    regs = [_regs(t) for t in ops]
    if op in ("v_mad_i64_i32", "v_mad_u64_u32"):
        dst, src = regs[0] + regs[1], sum(regs[2:], [])
    else:
        dst, src = regs[0], sum(regs[1:], [])
    acc = regs[-1] if op.startswith("v_dot2") else []      # the accumulator operand
    return op, dst, src, acc


FILL = "  #fill"   # tag: off-chain work the scheduler keeps back for slots where nothing else may issue (see schedule)


def schedule(lines):
    """Lines tagged FILL (the speculative form's rank terms: nothing on the serial chain waits for them) are only issued
    when no untagged instruction may issue -- they fill the wait states behind the prediction's v_dot2 pair instead of
    being spent on soft (back-to-back) gaps, which would leave an s_nop there."""
    n = len(lines)
    fill = [l.endswith(FILL) for l in lines]
    lines = [l[:-len(FILL)] if f else l for l, f in zip(lines, fill)]
    info = [_parse(l) for l in lines]
    hard = [dict() for _ in range(n)]   # hard[j][i] = minimum distance j - i
    soft = [dict() for _ in range(n)]
    last_write, readers = {}, {}
    for j, (op, dst, src, acc) in enumerate(info):
        def need(i, h, sft):
            hard[j][i] = max(hard[j].get(i, 0), h)
            soft[j][i] = max(soft[j].get(i, 0), sft, h)
        for r in src:
            if r in last_write:
                i = last_write[r]
                pop = info[i][0]
                sft = 2
                if pop.startswith("v_dot2"):
                    h = 1 if (op == pop and r in acc) else 4
                    if h == 1:
                        sft = 1     # accumulator chaining is forwarded: keep the fillers for the 3-slot gap behind the pair
                elif pop.startswith("v_pk_") or pop.startswith("v_cvt_pk"):
                    h = 2
                else:
                    h = 1
                need(i, h, sft)
        for r in dst:
            if r in last_write:
                i = last_write[r]
                need(i, 5 if info[i][0].startswith("v_dot2") and not op.startswith("v_dot2") else 1, 1)
            for i in readers.get(r, []):
                if i != j:
                    need(i, 1, 1)
        for r in src:
            readers.setdefault(r, []).append(j)
        for r in dst:
            last_write[r] = j
            readers[r] = []
    succ = [[] for _ in range(n)]
    for j in range(n):
        for i in hard[j]:
            succ[i].append(j)
    prio = [0] * n
    for i in range(n - 1, -1, -1):
        prio[i] = 1 + max([prio[j] + soft[j][i] - 1 for j in succ[i]], default=0)
    pos, order, t = {}, [], 0
    npred = [len(hard[j]) for j in range(n)]
    ready = {j for j in range(n) if npred[j] == 0}
    while len(pos) < n:
        ok = [j for j in ready if all(t - pos[i] >= d for i, d in hard[j].items())]
        if not ok:
            order.append("s_nop 0")
            t += 1
            continue
        main = [j for j in ok if not fill[j]]
        spare = sorted(j for j in ok if fill[j])
        good = [j for j in main if all(t - pos[i] >= d for i, d in soft[j].items())]
        if good:
            j = max(good, key=lambda k: (prio[k], -k))
        elif main and len(spare) < 2:
            j = max(main, key=lambda k: (prio[k], -k))
        else:
            j = spare[0]  # oldest filler first (its source registers are the next to be recycled); with two or more in
                          # hand one may also go into a soft gap -- one stays in reserve for the next hard gap
        pos[j] = t
        order.append(lines[j])
        t += 1
        ready.discard(j)
        for k in succ[j]:
            npred[k] -= 1
            if npred[k] == 0:
                ready.add(k)
    back_to_back = sum(1 for j in range(n) for i, d in hard[j].items() if pos[j] - pos[i] == 1 and info[i][1] and
                       set(info[i][1]) & set(info[j][2]))
    return order, back_to_back


LAB = set(filter(None, os.environ.get("QOA_GEN_LAB", "").split(",")))   # timing experiments (tools/k1lab/stamps.sh); never in the product


def pk_block(verbose=False, spec=False, ldsq=False, cheap=False, pen=False, count=False, wide=False):
    """Up to %[n] consecutive full slices of a block in one statement: candidate steps, winner election, exchange of
    the winner's state and staging of its slice, all through LDS; exits early (done < n) on a slice the packed path
    must decline, with the slice's entry state restored.

    Winner election = two 16-lane DPP min reductions (keys: rank >> 5, then rank[7:0] : |err0| : sf << 4).  The two
    wait states a DPP read needs after a VALU write of its source are filled with the slice-word packing, the loop
    counter and the NEXT slice's five sample reads, so the reads' latency is off the serial chain too.
    Exchange: every candidate parks its end state in its own LDS slot before the election; afterwards all 16 lanes read
    the winner's slot ((key2 & 0x1e0) selects it: slots are 32 bytes apart, rows 512-byte aligned).  The full form uses
    the first 16 bytes (weights + the two history pairs of the state); the speculative forms also SLOT2.

    spec=True is the speculative form: the steps carry no penalty arithmetic (pk_step); instead every candidate's
    weight energy minus 0x8ff << 18, sampled at steps 0, 4, .. 16, is kept as a running maximum QM and tested twice:
      all 64 candidates  QM < %[limq]   (arithmetic provably exact: the energy cannot have left the i16 window in the
                                         three unsampled steps either) -- before the election, as the full form does;
      the winner         QM < %[winq]   (its penalty was zero at all 20 steps, so its rank is its true rank, and every
                                         other candidate's true rank is at least its computed one: same winner).
    The second test comes after the election, when this slice's candidates already sit in the LDS slots; the slots are
    therefore double-buffered (%[pp] = byte distance) so that the entry state is still there for a late exit.

    pen=True (with spec and cheap) is the PENALTY form: the cheap-quotient speculative step plus the full form's penalty
    arithmetic on every step (26 instructions + the rank pair every second step).  Nothing is speculated about the
    penalty: the rank the election sees is the saturating err^2 sum plus the penalty sum, exact whenever it is below
    32767^2 (then so is its err^2 part); the exactness of the i16 arithmetic is the full form's test, penalty sum <
    %[limit] for all 64 candidates.  No energy sampling, no winner test after the election.  It is what the block loop
    falls back to where the speculative forms miss -- tonal passages, whose winners carry a penalty (3 % of real music's
    slices, clustered) -- instead of the 35-instruction full form."""
    assert not pen or (spec and cheap and not ldsq)
    assert not wide or (spec and not cheap and not pen and not ldsq)
    stamps = "stamps" in LAB and cheap and not pen       # s90 = cycles in the 20 steps, s91 = cycles in the rest of the loop
    N = dict(PK_FIXED, **SPEC_REGS) if spec else PK_FIXED
    assert not spec or (N["R"][3], N["RC"][1], PK_S[2], PK_S[3]) == ("v92", "v93", "v94", "v95")   # = SLOT2
    K1, M, K2, ADDR, OADDR = "v116", "v117", "v118", "v76", "v120"   # K1/M/K2 share the dot registers (long dead by now)
    TCUR, OCUR = "v78", "v77"
    SHI, SLO, ELO, EHI = "v80", "v81", "v82", "v83"
    CANDC, PADDR, THR, QM = "v65", "v64", "v119", N["qm"]
    cand = CANDC if spec else "%[cand]"
    reads = sample_reads(TCUR)
    if spec and not wide:
        K1 = ELO
    dpp = lambda r, n: f"v_min_u32_dpp {r}, {r}, {r} row_ror:{n} row_mask:0xf bank_mask:0xf"
    assert ELO == ELO_REG
    if spec:
        return pk_block_spec(N, verbose, ldsq, cheap, pen, stamps, count, wide, dict(
            K1=K1, M=M, K2=K2, ADDR=ADDR, OADDR=OADDR, TCUR=TCUR, OCUR=OCUR, SHI=SHI, SLO=SLO, ELO=ELO, CANDC=CANDC,
            PADDR=PADDR, THR=THR, QM=QM, reads=reads, dpp=dpp))
    # ---- the full form ----
    zero = [f"v_mov_b64 {N['e']}, 0", f"v_mov_b32 {N['pen']}, 0"]   # rank and penalty sums are cleared for the next slice
    L = [
        "s_mov_b32 s95, 0x10001",
        "ds_write_b128 %[cand], v[84:87]",                   # slot = entry state: what an early exit restores
        f"v_mov_b32 {TCUR}, %[tile]",
        f"v_subrev_u32 {OCUR}, 8, %[outb]",
        f"v_mov_b32 {ADDR}, %[cand]",
    ] + sample_reads("%[tile]") + zero + [
        "s_mov_b32 %[done], 0",
        "s_waitcnt lgkmcnt(0)",
        "Lqoa_top_%=:",
        "s_waitcnt lgkmcnt(1)",                              # the winner's state; a slice-word store may trail
    ]
    body = pk_prologue(N, zero=False)                        # rank and penalty sums are cleared in the election's gaps
    for i in range(20):
        body += pk_step(i, N, False)
    E = N["alt"][8]                                          # step 19 used the second temporary set
    body = unpack_samples() + body + [
        f"v_sub_u32 {E}, v{SMP0 + 19}, {PK_RC[1]}",
        f"v_mad_i64_i32 {N['e']}, vcc, {E}, {E}, {N['e']}",
        f"v_mad_u64_u32 {N['e']}, vcc, {N['pen']}, 1, {N['e']}",   # total rank = sum err^2 + sum penalty^2
    ]
    body, b2b = schedule(body)
    if verbose:
        print(f"block body: {len(body)} instructions after scheduling, {b2b} consumers directly behind their producer")
    L += body
    L += [
        f"v_cmp_ge_u32 vcc, {N['pen']}, %[limit]",
        "s_cbranch_vccnz Lqoa_declined_%=",
        # keys of the full form: rank < 2^37 -> rank >> 5, then rank[7:0] : |err0| : sf << 4
        # (slots are 32 bytes apart: %[sf4] = sf << 5, %[mf0] = 0x1e0; the second key keeps rank[6:0] : |err0| : sf << 5)
        # (round 4: the last key instruction, the slice word's halves and the candidates' state write sit in DPP gaps too)
        f"v_alignbit_b32 {K1}, {EHI}, {ELO}, 5",
        f"v_lshlrev_b32 {K2}, 25, {ELO}",
        f"v_lshl_or_b32 {K2}, {N['aerr0']}, 9, {K2}",
        f"v_min_u32_dpp {M}, {K1}, {K1} row_ror:8 row_mask:0xf bank_mask:0xf",
        f"v_or_b32 {K2}, %[sf4], {K2}",
        f"v_lshrrev_b32 {SHI}, 2, {N['acca']}",
        dpp(M, 4),
        f"v_or_b32 {SHI}, %[sfsh], {SHI}",
        f"v_add_u32 {TCUR}, 48, {TCUR}",
        dpp(M, 2),
        reads[0],
        f"v_lshl_or_b32 {SLO}, {N['acca']}, 30, {N['accb']}",
        dpp(M, 1),
        f"v_cmp_eq_u32 vcc, {K1}, {M}",
        f"v_cndmask_b32 {K2}, -1, {K2}, vcc",
        f"ds_write_b128 {cand}, v[84:87]",
        reads[1],
        dpp(K2, 8),
        reads[2],
        f"v_add_u32 {OCUR}, 8, {OCUR}",
        dpp(K2, 4),
        "s_add_u32 %[done], %[done], 1",
        "s_cmp_lt_u32 %[done], %[n]",
        dpp(K2, 2),
    ] + zero + [                                             # clear e and pen for the next slice (dead by now)
        dpp(K2, 1),
        f"v_bfi_b32 {ADDR}, %[mf0], {K2}, {cand}",           # the winner's slot
        f"ds_read_b128 v[84:87], {ADDR}",
        f"v_cmp_eq_u32 vcc, {ADDR}, {cand}",
        f"v_cndmask_b32 {OADDR}, %[dummy], {OCUR}, vcc",      # the winner stages its slice, the others hit a dummy
        f"ds_write_b64 {OADDR}, v[80:81]",
        "s_cbranch_scc1 Lqoa_top_%=",
        "s_branch Lqoa_end_%=",
        "Lqoa_declined_%=:",
        f"ds_read_b128 v[84:87], {ADDR}",                    # entry state of the declined slice
        "Lqoa_end_%=:",
        "s_waitcnt lgkmcnt(0)",
    ]
    return L


def pk_block_spec(N, verbose, ldsq, cheap, pen, stamps, count, wide, G):
    """The speculative block loops (exact quotient, cheap quotient, LDS tables, penalty form).  ONE branch per slice:
    a not-taken s_cbranch_vccnz behind its v_cmp costs a lone wavefront ~24 cycles and a taken branch ~21
    (tools/ubench/branchbench.hip) -- the four exit tests of round 2's loop plus the loop branch were 4.6 % of the slice.
    Every exit condition is therefore folded into one per-lane comparison at the end of the slice (the candidate slots
    are double-buffered, so any slice can still be abandoned there: its entry state is in the other buffer):

      QMv >= THR   with  THR = winner lane ? WQ : LQ
        QMv  the candidate's running energy maximum (penalty form: its penalty sum)
        LQ   %[limq] (penalty form: %[limit]): some candidate's i16 arithmetic may have wrapped -- any lane;
             replaced by "always" when the row's best rank M is not provably exact (M >= 32767^2, spec_rank_pair)
        WQ   %[winq]: the winner's penalty was not provably zero (not in the penalty form);
             replaced by "always" in a MARKED lane (cheap quotient, cheap_quotient_note): a marked candidate stays in the
             election with its BOUND as rank -- what it provably costs at least (tile dword 0) -- while every real rank of
             the cheap forms is counted from 1 (spec_rank_pair): the key is strictly below the candidate's true rank + 1, so an
             unmarked winner has truly beaten it (rank + 1 <= bound <= its rank, no tie possible), and a bound of 0 -- no
             proof at all -- wins every election, which abandons the slice.
    "always" is INT_MIN for the signed energy comparison and 0 for the unsigned penalty comparison (CH_BIG).
    The loop branch is taken when no lane's comparison fired and slices are left: tmp = fired ? -1 : done; tmp < n."""
    K1, M, K2, ADDR, OADDR, TCUR, OCUR = G["K1"], G["M"], G["K2"], G["ADDR"], G["OADDR"], G["TCUR"], G["OCUR"]
    SHI, SLO, ELO, CANDC, PADDR, THR, QM, reads, dpp = (G[k] for k in ("SHI", "SLO", "ELO", "CANDC", "PADDR", "THR", "QM",
                                                                       "reads", "dpp"))
    WQ, MK = "v83", CH_BND                                   # v83: the rank's high dword in the full form, unused here
    if cheap:
        K1 = "v116"                                          # the election's copy of the rank (a v_dot2's result may not be overwritten so soon)
    LQ = QM if pen else ("%[limq]" if wide else N["pen"])    # a register the form does not use otherwise
    QMV = N["pen"] if pen else QM
    LIMV = "%[limitv]"                                       # penalty form: kPkPenLimit in a VGPR (v_cndmask reads vcc: no SGPR source besides)
    cmp_ge = "v_cmp_ge_u32" if pen else "v_cmp_ge_i32"
    L = [
        "s_mov_b32 s95, 0x10001",
        f"s_mov_b32 s94, 0x{RANK_EXACT_BELOW + (1 if cheap else 0):x}",   # (the cheap forms' ranks are offset by 1)
        f"v_mov_b32 {CH_BIG}, {'0' if pen else '0x80000000'}",
    ] + ([f"v_mov_b32 {MK}, 0"] if cheap and count else []) + [
        "ds_write_b128 %[cand], v[84:87]",                   # slot = entry state: what an exit restores
        f"v_mov_b32 {TCUR}, %[tile]",
        f"v_subrev_u32 {OCUR}, 8, %[outb]",
        f"v_mov_b32 {ADDR}, %[cand]",
        f"v_mov_b32 {CANDC}, %[cand]",
    ] + sample_reads("%[tile]") + pk_prologue(N, zero=False, spec=True) + (   # once per call; then SLOT2 carries it
        [] if ldsq else [f"v_mov_b32 {N['acca']}, 0", f"v_mov_b32 {N['accb']}, 0"]) + [
        "s_mov_b32 %[done], 0",
    ] + (["s_mov_b32 s90, 0", "s_mov_b32 s91, 0", "s_memtime s[88:89]"] if stamps else []) + [
        "s_waitcnt lgkmcnt(0)",
        "Lqoa_top_%=:",
        # the winner's state; a slice-word store may trail, and SLOT2's read (waited for below)
        "s_waitcnt lgkmcnt(2)",
    ] + (["s_memtime s[86:87]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s92, s86, s88", "s_add_u32 s91, s91, s92"] if stamps else [])
    # (no per-slice clearing: the penalty sum, the wide form's rank and the slice-word halves all start with their first term)
    body = []
    if ldsq:
        N = dict(N, alt=None)                                # one temporary set: the steps are not interleaved
    else:
        N = dict(N, fill=FILL, cheap=cheap, pen_steps=pen, acc_cleared=True, wide=wide)
    acc_line = None
    for i in range(20):
        if ldsq:
            lines, acc_line = pk_step_ldsq(i, N, acc_line)
            body += lines
        else:
            body += pk_step(i, N, True)
    if ldsq:
        body.append(acc_line)
        b2b = -1                                             # kept in program order (see pk_step_ldsq)
    else:
        E = N["alt"][8]                                      # pair 9: the odd pairs' register (pk_step)
        if wide:
            body += [f"v_sub_u32_sdwa {E}, sext(v{SPK0 + 9}), sext({N['R'][0]}) dst_sel:DWORD dst_unused:UNUSED_PAD "
                     f"src0_sel:WORD_1 src1_sel:WORD_1",                           # the last error (pk_step, wide)
                     f"v_mad_i64_i32 {N['e']}, vcc, {E}, {E}, {N['e']}" + FILL]
        else:
            body += [f"v_pk_sub_i16 {E}, v{SPK0 + 9}, {N['R'][0]} clamp" + FILL,   # the last error pair (spec_rank_pair)
                     f"v_dot2_i32_i16 {ELO}, {E}, {E}, {ELO} clamp" + FILL]
        # (the switch to the other slot buffer -- the entry state stays put -- sits in a wait-state gap of the election)
        if cheap:                                                                  # marked lanes (cheap_quotient_note): the sign
            body.append(f"v_or_b32 {CH_W}, {N['acca']}, {N['accb']}" + FILL)       # bit of a slice-word half (pk_step)
            body.append(f"v_cmp_gt_i32 vcc, 0, {CH_W}" + FILL)
        # (penalty form: total rank = err^2 sum + penalty sum, both < 2^31: no carry -- added in the election, behind the three
        # instructions that stand between the rank's last v_dot2 and its first reader anyway; inside the scheduled body it had
        # nothing to wait behind but three s_nop)
        body, b2b = schedule(body)
    if verbose:
        print(f"block body (speculative{', cheap' if cheap else ''}{', penalty' if pen else ''}{', lds' if ldsq else ''}): "
              f"{len(body)} instructions after scheduling, {b2b} consumers directly behind their producer")
    # SLOT2 is first needed near the end of step 0: wait for it there, not at the loop top
    first = min(k for k, l in enumerate(body) if l.startswith("v_") and set(_parse(l)[2]) & {"v92", "v93", "v94", "v95"})
    body.insert(first, "s_waitcnt lgkmcnt(1)")
    L += body
    if stamps:
        L += ["s_memtime s[88:89]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s92, s88, s86", "s_add_u32 s90, s90, s92"]
    # ---- election, exchange, the one exit test ----
    # ELO's last writer is a v_dot2 (or the penalty form's v_add): three instructions go in front of its first reader
    WQr = WQ if cheap else ("%[winq]" if not pen else LIMV)
    if wide:   # the full form's keys: rank < 2^37 -> rank >> 5, then rank[6:0] : |err0| : sf << 5
        key_lines = [
            f"v_alignbit_b32 {K1}, v83, {ELO}, 5",
            f"v_lshlrev_b32 {K2}, 25, {ELO}",
            f"v_lshl_or_b32 {K2}, {N['aerr0']}, 9, {K2}",
            f"v_or_b32 {K2}, %[sf4], {K2}",
        ]
    else:
        key_lines = [f"v_lshl_or_b32 {K2}, {N['aerr0']}, 9, %[sf4]"]   # second key |err0| : sf << 5 (slots are 32 bytes apart, %[mf0] = 0x1e0)
    # Both reductions are chains of four DPP reads, each of which needs two instructions between it and the VALU write of its
    # source; a v_dot2's result (the rank) needs three before it is read.  The gaps hold the election's own bookkeeping --
    # the keys, the slice word's halves, tile / out-tile cursors, the next slice's three sample reads, the switch to the other
    # slot buffer, the exactness test of the saturating rank -- so that (but for one s_nop in the forms with a 32-bit rank, and
    # the statement of counting plans, which keeps round 3's order) nothing of it takes an issue slot of its own.
    shi_line, slo_line = f"v_lshrrev_b32 {SHI}, 2, {N['acca']}", f"v_lshl_or_b32 {SLO}, {N['acca']}, 30, {N['accb']}"
    or_shi, add_t, add_o = f"v_or_b32 {SHI}, %[sfsh], {SHI}", f"v_add_u32 {TCUR}, 48, {TCUR}", f"v_add_u32 {OCUR}, 8, {OCUR}"
    xor_c = f"v_xor_b32 {CANDC}, %[pp], {CANDC}"             # the other slot buffer: the entry state stays put
    dpp8 = f"v_min_u32_dpp {M}, {K1}, {K1} row_ror:8 row_mask:0xf bank_mask:0xf"
    mov_p = f"v_mov_b32 {PADDR}, {ADDR}"                     # the slot that holds this slice's entry state (read if it is abandoned)
    s_add = "s_add_u32 %[done], %[done], 1"
    gap_a = [f"v_cmp_le_u32 vcc, s94, {M}",                  # is the best rank its true rank?  (see spec_rank_pair)
             f"v_cndmask_b32 {LQ}, {LIMV if pen else '%[limq]'}, {CH_BIG}, vcc"]
    gap_b, gap_c = [s_add, slo_line], ["s_nop 0", mov_p]
    if cheap:
        k1_line = f"v_cndmask_b32 {K1}, {ELO}, {BOUND_IN}, vcc"   # a marked candidate's rank: what it provably costs (real ranks count from 1)
        wq_line = f"v_cndmask_b32 {WQ}, {LIMV if pen else '%[winq]'}, {CH_BIG}, vcc"   # ... and if that wins, the slice is abandoned
        if pen:   # (the sum goes to the election's copy: a v_dot2's result may be READ three instructions later, not overwritten)
            k1_line = f"v_cndmask_b32 {K1}, {K1}, {BOUND_IN}, vcc"
        pen_sum = [f"v_add_u32 {K1}, {ELO}, {N['pen']}"] if pen else []   # total rank = err^2 sum + penalty sum (both < 2^31)
        if count:                                            # (counting plans: marked (candidate, slice) pairs of this call)
            L += key_lines + [shi_line, slo_line] + pen_sum + [k1_line, wq_line, f"v_addc_co_u32 {MK}, vcc, 0, {MK}, vcc", dpp8,
                              or_shi, add_t, dpp(M, 4), reads[0], reads[1], dpp(M, 2), reads[2], add_o, xor_c, dpp(M, 1)]
            gap_b = [s_add, "s_nop 0"]
        else:   # (the rank's last v_dot2, then: the marks' v_cmp, the second key, wq_line -- three -- and its first reader)
            L += key_lines + [wq_line] + pen_sum + [k1_line, shi_line, or_shi, dpp8, add_t, reads[0], dpp(M, 4), reads[1], reads[2],
                              dpp(M, 2), add_o, xor_c, dpp(M, 1)]
    elif wide:  # (the rank is a v_mad_i64's: no v_dot2 wait states; its true rank, so no exactness test -- the gap takes two reads)
        L += key_lines[:3] + [dpp8, key_lines[3], shi_line, dpp(M, 4), or_shi, add_t, dpp(M, 2), reads[0], xor_c, dpp(M, 1)]
        gap_a, gap_c = [reads[1], reads[2]], [add_o, mov_p]
    else:       # exact quotient, 32-bit saturating rank (registers or LDS tables): the three behind the v_dot2 are key, high half, head
        L += key_lines + [shi_line, or_shi, dpp8, add_o, add_t, dpp(M, 4), reads[0], reads[1], dpp(M, 2), reads[2], xor_c, dpp(M, 1)]
    L += [
        f"v_cmp_eq_u32 vcc, {K1}, {M}",
        f"v_cndmask_b32 {K2}, -1, {K2}, vcc",
        f"ds_write_b128 {CANDC}, v[84:87]",
        f"ds_write_b128 {CANDC}, {SLOT2} offset:16",
        dpp(K2, 8),
    ] + gap_a + [dpp(K2, 4)] + gap_b + [dpp(K2, 2)] + gap_c + [
        dpp(K2, 1),
        f"v_bfi_b32 {ADDR}, %[mf0], {K2}, {CANDC}",          # the winner's slot
        f"ds_read_b128 v[84:87], {ADDR}",
        f"ds_read_b128 {SLOT2}, {ADDR} offset:16",
    ] + ["s_nop 0"] * sum(int(t[4:]) for t in LAB if t.startswith("nops")) + [   # (lab: is the winner's state read exposed?)
        f"v_cmp_eq_u32 vcc, {ADDR}, {CANDC}",
        f"v_cndmask_b32 {OADDR}, %[dummy], {OCUR}, vcc",      # the winner stages its slice, the others hit a dummy
        f"v_cndmask_b32 {THR}, {LQ}, {WQr}, vcc",
        f"ds_write_b64 {OADDR}, v[80:81]",
        f"{cmp_ge} vcc, {QMV}, {THR}",                       # the one exit test (see above)
        "s_cmp_lg_u64 vcc, 0",
        "s_cselect_b32 %[tmp], -1, %[done]",
        "s_cmp_lt_u32 %[tmp], %[n]",
        "s_cbranch_scc1 Lqoa_top_%=",
        "s_cmp_lg_u32 %[tmp], -1",
        "s_cbranch_scc1 Lqoa_end_%=",                        # all n slices done
        "s_sub_u32 %[done], %[done], 1",                     # abandoned: the slice's entry state comes back
        f"ds_read_b128 v[84:87], {PADDR}",
        "Lqoa_end_%=:",
        "s_waitcnt lgkmcnt(0)",
    ] + (["s_mov_b32 %[tb], s90", "s_mov_b32 %[tt], s91"] if stamps else [])
    return L


def emit(name, lines):
    return f"#define {name} \\\n" + " \\\n".join(f'    "{l}\\n\\t"' for l in lines) + "\n"


def render(verbose=False):
    lines = sample_reads("%[tile]")
    for j in range(4):
        lines += [f"v_ashrrev_i32 v{S0 + j}, 31, %[h{j}]", f"v_or_b32 v{S0 + j}, 1, v{S0 + j}"]
    lines += ["v_mov_b64 %[e], 0", "v_mov_b32 %[pen], 0", "v_mov_b32 %[acca], 0", "v_mov_b32 %[accb], 0"]
    lines += ["s_waitcnt lgkmcnt(0)"] + unpack_samples()
    for i in range(20):
        lines += step(i)
    pk = pk_slice()
    blk = pk_block(verbose)
    blk_spec = pk_block(verbose, spec=True)
    blk_ldsq = pk_block(verbose, spec=True, ldsq=True)
    blk_cheap = pk_block(verbose, spec=True, cheap=True)
    blk_cheap_count = pk_block(False, spec=True, cheap=True, count=True)
    blk_pen = pk_block(verbose, spec=True, cheap=True, pen=True)
    blk_wide = pk_block(verbose, spec=True, wide=True)
    blk_32 = b32_block()
    out = ("// GENERATED by tools/gen_slice_asm.py -- do not edit.\n"
           "// QOA_SLICE_ASM_TEXT: 20 x 41-instruction candidate steps of one slice, full 32-bit weights.\n"
           "// QOA_SLICE_PK_ASM_TEXT: 20 x 35-instruction steps on i16-packed weights and history (fast path).\n"
           "// QOA_BLOCK_PK_ASM_TEXT: a loop over consecutive full slices on the packed path incl. winner election and\n"
           "//   state exchange; fixed registers v67-v127, state in v[84:87].\n"
           "// QOA_BLOCK_SPEC_ASM_TEXT: the same loop with speculative steps (26/29 instructions, no penalty arithmetic) and\n"
           "//   the two energy tests that prove a slice's result; fixed registers v64-v127.\n"
           + emit("QOA_SLICE_ASM_TEXT", lines) + "\n" + emit("QOA_SLICE_PK_ASM_TEXT", pk) + "\n"
           + emit("QOA_BLOCK_PK_ASM_TEXT", blk) + "\n" + emit("QOA_BLOCK_SPEC_ASM_TEXT", blk_spec) + "\n"
           + "// QOA_BLOCK_SPEC_LDSQ_ASM_TEXT: the speculative loop with the quantiser + dequantiser in an LDS table (21/24\n"
           + "//   VALU + 1 LDS read per step): for launches that give every SIMD several wavefronts.\n"
           + emit("QOA_BLOCK_SPEC_LDSQ_ASM_TEXT", blk_ldsq) + "\n"
           + "// QOA_BLOCK_CHEAP_ASM_TEXT: the speculative loop with the cheap quotient (22/25 VALU per step; level from |n|, sign\n"
           + "//   from the residual, candidates whose quotient wrapped to the opposite sign are marked and proven losers).\n"
           + emit("QOA_BLOCK_CHEAP_ASM_TEXT", blk_cheap) + "\n"
           + "// QOA_BLOCK_CHEAP_COUNT_ASM_TEXT: the same with a count of the marked (candidate, slice) pairs (plans with counters).\n"
           + emit("QOA_BLOCK_CHEAP_COUNT_ASM_TEXT", blk_cheap_count) + "\n"
           + "// QOA_BLOCK_PEN_ASM_TEXT: the cheap-quotient loop WITH the penalty arithmetic on every step (26/step): what the\n"
           + "//   block loop falls back to where the speculative forms miss (winners that carry a penalty), before the full form.\n"
           + emit("QOA_BLOCK_PEN_ASM_TEXT", blk_pen) + "\n"
           + "// QOA_BLOCK_WIDE_ASM_TEXT: exact-quotient speculative steps with the true 64-bit rank (28/step; the weight energy sampled at\n"
           + "//   every 4th step as in the other speculative forms): full-scale noise -- winners' ranks beyond 32767^2, marked candidates without a bound.\n"
           + emit("QOA_BLOCK_WIDE_ASM_TEXT", blk_wide) + "\n"
           + "// QOA_BLOCK_32_ASM_TEXT: a loop over consecutive full slices on the 32-bit step (it cannot decline): the block loop's\n"
           + "//   last resort; state in v[84:91], fixed registers v72-v127.\n"
           + emit("QOA_BLOCK_32_ASM_TEXT", blk_32))
    return out, (len(lines), len(pk), len(blk), len(blk_spec), len(blk_ldsq), len(blk_cheap), len(blk_pen), len(blk_wide), len(blk_32))


INC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "qoaudio_amd", "csrc", "qoa_slice_asm.inc")


def main():
    out, counts = render(verbose=True)
    path = os.environ.get("QOA_GEN_OUT", INC_PATH)           # (lab variants go elsewhere)
    assert not LAB or path != INC_PATH, "QOA_GEN_LAB variants must not overwrite the product's texts"
    with open(path, "w") as f:
        f.write(out)
    print(f"wrote {path}: " + " + ".join(str(c) for c in counts) + " lines")


if __name__ == "__main__":
    main()
