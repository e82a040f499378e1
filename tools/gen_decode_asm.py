#!/usr/bin/env python3
"""Generates qoaudio_amd/csrc/qoa_decode_asm.inc (run: python tools/gen_decode_asm.py): one slice ROW of K2u (the decoder
for uniform full frames, qoa_decode.hip) as ONE inline-asm statement per (channel count, row of the round).

Why asm: K2u is bound by VALU issue (profiles/r02_*: the VALU is busy ~80 % of the kernel at the ~1.9 GHz the chip runs
this memory-heavy kernel at), so its cost is the instruction count of the serial sample step.  hipcc's code spends
14.4 VALU per sample (an accumulator-zeroing move per sample for v_dot2c, state shuffling around every row, a
10-instruction range guard per row); this text spends 9 per sample + 1.5 for the table lookup + ~9 per row.

The row, for one lane = one (frame, channel) chain (reference: decode_one_slice_per_channel, src/lib.rs:291-330):
  bswap the slice words; scalefactor -> its row of the LDS PAIR table [sf][code a * 8 + code b] = deq(a) | deq(b) << 16
  (src/lib.rs:847-864); 10 x {6 bits -> entry, entry >> 4 = the two weight steps of :822};  range guard (below);
  20 x sample step:
      p      = dot2(w01, A) + dot2(w23, B)              the 4-tap prediction on i16 pairs, i32 wrapping sum (:799-805)
      u      = (p >> 13) + deq                          unclamped reconstruction, alternating between two registers
      w     += sign(h) * (deq >> 4)                     2 x v_pk_mad_u16 with the (+-1, +-1) pairs (:822-826)
      B'     = (clamp(u of the sample before), clamp(u))  ONE v_cvt_pk_i16_i32: the clamp of :310 and the next history
                                                        pair of :827 (clamping the previous value again gives the same i16)
      image  <- recon                                   ds_write_b16_d16_hi of B' (stereo) / ds_write_b32 of the pair (mono)
Pair ring as in K1: B_n = (h[n-2], h[n-1]) lives in R[n % 4]; step n reads A = R[(n+2) % 4], B = R[n % 4], writes
R[(n+1) % 4]; 20 % 4 == 0 brings every ring register back to its role, so %[pa], %[pm], %[pb] are plain in/out operands.

Range guard (exactness of the i16 arithmetic).  One sample moves a weight by at most ceil(mag3(sf) / 16); %[slack] is a
lower bound of 32767 - max|w|.  A row whose 20 * ceil(mag3 / 16) (LDS table s_grow[sf]) fits the slack cannot leave
i16: it runs and the slack shrinks by that amount.  Otherwise nothing is executed, %[flag] = 1, and the caller
recomputes the exact slack and either re-enters with %[force] != 0 or takes the 32-bit path.

Wait states the assembler does not pad inside an asm statement (gfx940/950; see tools/gen_slice_asm.py): a v_dot2 result
needs 3 other instructions before a different VALU reads it, a packed (v_pk_*) result 1.  The step order below keeps
them by construction (asserted by check_hazards()).
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# fixed scratch registers (clobbered): 20 table entries + temporaries
E0 = 64                      # v64..v73: pair-table entries (deq, deq), v74..v83: their weight steps (deq >> 4 pairs)
D0 = 74
HI, LO, TAB, T, G = 84, 85, 86, 87, 88
P, U, RC0, RC1, RT, ST = 89, 90, 91, 92, 93, 94   # RT / ST: the fourth ring slots
CLOBBERS = list(range(64, 95))

def ring(kind):
    if kind == "P":
        return ["%[pb]", f"v{RT}", "%[pa]", "%[pm]"]
    return ["%[sb]", f"v{ST}", "%[sa]", "%[sm]"]


def extraction():
    """10 lookups per row: entry j of the PAIR table [sf][code(2j) * 8 + code(2j+1)] = deq(2j) | deq(2j+1) << 16, then the
    two weight steps (deq >> 4, src/lib.rs:822) as one packed shift.  (One lookup per sample cost 2 VALU + 1 LDS read per
    sample; this is 1.5 VALU + 0.5 reads.  8-byte entries that carry the weight steps too -- 1 VALU + 0.5 ds_read_b64 --
    measured no faster for stereo and slower for mono: profiles/r02_k2lab_log.md, run 17.)"""
    L = []
    for j in range(10):
        p = 54 - 6 * j
        e, d = f"v{E0 + j}", f"v{D0 + j}"
        if p >= 32:
            L.append(f"v_bfe_u32 {e}, v{HI}, {p - 32}, 6")
        elif p + 5 < 32:
            L.append(f"v_bfe_u32 {e}, v{LO}, {p}, 6")
        else:  # the pair that straddles the two words
            L.append(f"v_alignbit_b32 {e}, v{HI}, v{LO}, {p}")
            L.append(f"v_and_b32 {e}, 63, {e}")
        L.append(f"v_lshl_add_u32 {e}, {e}, 2, v{TAB}")
        L.append(f"ds_read_b32 {e}, {e}")
    return L


def deltas(j0, j1):
    return [f"v_pk_ashrrev_i16 v{D0 + j}, 4, v{E0 + j} op_sel_hi:[0,1]" for j in range(j0, j1)]


def step(i, nch, row, ring_bytes, mirror=0):
    R, S = ring("P"), ring("S")
    A, B, Bn = R[(i + 2) % 4], R[i % 4], R[(i + 1) % 4]
    SA, SB = S[(i + 2) % 4], S[i % 4]
    e, d = f"v{E0 + i // 2}", f"v{D0 + i // 2}"
    half = i % 2          # which half of the pair registers this sample uses
    bcast = "op_sel:[1,0,0]" if half else "op_sel:[0,0,0] op_sel_hi:[0,1,1]"   # that half into both lanes of src0
    u, uprev = f"v{RC0 + i % 2}", f"v{RC0 + (i + 1) % 2}"
    L = [
        f"v_dot2_i32_i16 v{P}, %[w01], {A}, 0",
        f"v_dot2_i32_i16 v{P}, %[w23], {B}, v{P}",
        f"v_pk_ashrrev_i16 {SB}, 15, {B} op_sel_hi:[0,1]",       # the inline 15 has no high half to select
        f"v_pk_mad_u16 %[w01], {d}, {SA}, %[w01] {bcast}",
        f"v_or_b32 {SB}, %[one2], {SB}",                           # (+-1, +-1) for B
        f"v_ashrrev_i32 {u}, 13, v{P}",
        f"v_add_u32_sdwa {u}, {u}, sext({e}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_{half}",
        f"v_pk_mad_u16 %[w23], {d}, {SB}, %[w23] {bcast}",
        f"v_cvt_pk_i16_i32 {Bn}, {uprev}, {u}",                   # (h3, recon), both clamped
    ]
    if nch == 1:
        if i & 1:  # consecutive samples are adjacent in a mono image: the new history pair IS the dword to store
            off = ((row * 20 + i - 1) * 2) % ring_bytes
            L.append(f"ds_write_b32 %[img], {Bn} offset:{off}")
    else:
        off = ((row * 20 + i) * nch * 2) % ring_bytes
        L.append(f"ds_write_b16_d16_hi %[img], {Bn} offset:{off}")   # the new sample is the pair's high half
        if off < mirror:   # (absolute-line rows: the ring's first bytes again behind its end, see GEOMETRY_ABS)
            L.append(f"ds_write_b16_d16_hi %[img], {Bn} offset:{off + ring_bytes}")
    return L


def row_text(nch, row, ring_bytes, mirror=0):
    L = [
        f"v_perm_b32 v{HI}, %[rx], %[rx], %[bsw]",
        f"v_perm_b32 v{LO}, %[ry], %[ry], %[bsw]",
        f"v_lshrrev_b32 v{T}, 28, v{HI}",
        f"v_lshl_add_u32 v{TAB}, v{T}, 8, %[tab]",                 # &s_deq2[sf][0]: 64 entries of 4 bytes
        f"v_lshl_add_u32 v{T}, v{T}, 2, %[grow]",
        f"ds_read_b32 v{G}, v{T}",
    ]
    L += extraction()
    L += [
        "s_cmp_lg_u32 %[force], 0",
        "s_cbranch_scc1 Lqoa_go_%=",
        "s_waitcnt lgkmcnt(10)",                      # the 10 table reads are younger than G's
        f"v_cmp_lt_i32 vcc, %[slack], v{G}",
        "s_cbranch_vccnz Lqoa_decline_%=",
        f"v_sub_u32 %[slack], %[slack], v{G}",
        "Lqoa_go_%=:",
    ]
    # LDS ops retire in order: the table reads first, then one image write per sample (per second sample, mono)
    L += ["s_waitcnt lgkmcnt(5)"] + deltas(0, 5)      # entries 0..4 are in (the 5 youngest reads may still be out)
    L += [f"v_ashrrev_i32 v{RC1}, 16, {ring('P')[0]}"]  # h3: sample 0's "reconstruction before"
    for i in range(20):
        if i == 8:
            # before sample 10: everything but the youngest image writes; the table reads are all older than those
            L += ["s_waitcnt lgkmcnt(%d)" % (4 if nch == 1 else 8)] + deltas(5, 10)
        L += step(i, nch, row, ring_bytes, mirror)
    L += [
        "s_mov_b32 %[flag], 0",
        "s_branch Lqoa_end_%=",
        "Lqoa_decline_%=:",
        "s_waitcnt lgkmcnt(0)",                       # the scratch registers are the compiler's again after this statement
        "s_mov_b32 %[flag], 1",
        "Lqoa_end_%=:",
    ]
    return L


_MOD = re.compile(r"\b(op_sel_hi|op_sel|offset|dst_sel|dst_unused|src0_sel|src1_sel):(\[[^\]]*\]|\S+)")


def _regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"sext\((.*)\)", tok)
    if m:
        tok = m.group(1)
    if re.fullmatch(r"v\d+", tok) or tok.startswith("%["):
        return [tok]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return [f"v{r}" for r in range(int(m.group(1)), int(m.group(2)) + 1)]
    return []


def check_hazards(lines):
    """dot2 -> other VALU read: >= 3 instructions between; packed -> VALU read: >= 1 between."""
    last = {}
    n = 0
    for l in lines:
        if not l.startswith("v_"):
            continue  # (conservative: only VALU instructions are counted as wait states)
        op, rest = l.split(None, 1)
        ops = [_regs(t) for t in _MOD.sub("", rest).split(",")]
        dst, src = ops[0], sum(ops[1:], [])
        for r in src:
            if r in last:
                pop, pos = last[r]
                if pop.startswith("v_dot2") and not (op.startswith("v_dot2") and r in ops[-1]):
                    assert n - pos >= 4, (l, pop, n - pos)
                if pop.startswith("v_pk_") or pop.startswith("v_cvt_pk"):
                    assert n - pos >= 2, (l, pop, n - pos)
        for r in dst:
            last[r] = (op, n)
        n += 1


def emit(name, lines):
    return f"#define {name} \\\n" + " \\\n".join(f'    "{l}\\n\\t"' for l in lines) + "\n"


# (channels, rows per round, image ring bytes): must agree with K2uCfg in qoa_decode.hip (static_asserts there)
GEOMETRY = {1: (16, int(os.environ.get("QOA_DEC_RING1", "256"))), 2: (8, int(os.environ.get("QOA_DEC_RING2", "256"))), 4: (4, 640), 8: (2, 640),   # (env: tools/k2lab variants)
            # channel counts that do not divide a wavefront (round 4): rounds of 16 / gcd(channels, 16) rows, 2-row phases
            # (ring = whole lines AND whole sample-times: the channel offset sits in %[img], outside the wrapped offset)
            3: (16, 384), 5: (16, 640), 6: (8, 768), 7: (16, 896)}


# Rows of the ABSOLUTE-LINE variant of K2u (round 5: frames whose PCM is off the 128-byte grid): (rows per round, ring bytes,
# mirror bytes).  There the flush reads 16-byte pieces that are aligned to the frame's absolute cache lines, i.e. at ring
# offsets congruent to -(address mod 128): a piece may start 8 bytes before the ring's end, so the ring's first `mirror` bytes
# are written a second time behind its end (the sample's lane issues a second ds_write: 2 of 80 per ring revolution) and every
# piece is 16 contiguous bytes.  The ring divides the round's bytes (640), so that the bytes a round leaves behind for the next
# round's first line sit where that line's pieces look for them.  Must agree with K2uCfg<NCH, true> in qoa_decode.hip.
GEOMETRY_ABS = {2: (8, 320, 8), 4: (4, 640, 8), 8: (2, 640, 16)}   # (8 channels: the first whole sample-time, 16 bytes; 8 are needed)


def render():
    out = ["// GENERATED by tools/gen_decode_asm.py -- do not edit.",
           "// QOA_DEC_ROW_TEXT_<channels>_<row of the round>: one slice row of K2u (20 samples of every chain of the wave) on",
           "// i16-packed weights / history; scratch v64-v94.  See the generator for operands and wait-state rules.",
           f"#define QOA_DEC_ROW_CLOBBERS " + ", ".join(f'"v{r}"' for r in CLOBBERS),
           ] + ["#define QOA_DEC_ASM_RING_%d %d" % (n, GEOMETRY[n][1]) for n in sorted(GEOMETRY)] + [
               "#define QOA_DEC_ASM_ROWS_%d %d" % (n, GEOMETRY[n][0]) for n in sorted(GEOMETRY)] + [
               "#define QOA_DEC_ASM_ABS_RING_%d %d" % (n, GEOMETRY_ABS[n][1]) for n in sorted(GEOMETRY_ABS)] + [
               "#define QOA_DEC_ASM_ABS_MIRROR_%d %d" % (n, GEOMETRY_ABS[n][2]) for n in sorted(GEOMETRY_ABS)] + [
               "// ... and over the rows of the absolute-line variant: QOA_DEC_ROW_CASE_ABS(N, R) -> QOA_DEC_ROW_TEXT_A<N>_<R>",
               "#define QOA_DEC_ROW_CASES_ABS " + " ".join(f"QOA_DEC_ROW_CASE_ABS({n}, {r})" for n in sorted(GEOMETRY_ABS)
                                                            for r in range(GEOMETRY_ABS[n][0])),
               "// the kernel's dispatch over (channel count, row of the round): expands QOA_DEC_ROW_CASE(N, R) for every text below",
               "#define QOA_DEC_ROW_CASES " + " ".join(f"QOA_DEC_ROW_CASE({n}, {r})" for n in sorted(GEOMETRY) for r in range(GEOMETRY[n][0])),
               ""]
    n_lines = 0
    for nch, (rows, ring_bytes) in sorted(GEOMETRY.items()):
        assert ring_bytes % (32 if nch == 1 else 64) == 0 and (nch == 1 or (ring_bytes % 128 == 0 and ring_bytes % (2 * nch) == 0)), (nch, ring_bytes)
        for row in range(rows):
            t = row_text(nch, row, ring_bytes)
            check_hazards(t)
            n_lines = len(t)
            out.append(emit(f"QOA_DEC_ROW_TEXT_{nch}_{row}", t))
    for nch, (rows, ring_bytes, mirror) in sorted(GEOMETRY_ABS.items()):
        assert nch > 1 and (rows * 40 * nch) % ring_bytes == 0 and ring_bytes % (2 * nch) == 0 and mirror % (2 * nch) == 0 and mirror >= 8
        for row in range(rows):
            t = row_text(nch, row, ring_bytes, mirror)
            check_hazards(t)
            out.append(emit(f"QOA_DEC_ROW_TEXT_A{nch}_{row}", t))
    return "\n".join(out), n_lines


INC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "qoaudio_amd", "csrc", "qoa_decode_asm.inc")


def main():
    out, n = render()
    path = os.environ.get("QOA_GEN_OUT", INC_PATH)
    assert path != INC_PATH or not ({"QOA_DEC_RING1", "QOA_DEC_RING2"} & set(os.environ)), "lab geometry must not overwrite the product's rows"
    with open(path, "w") as f:
        f.write(out)
    print(f"wrote {path}: {n} lines per row")


if __name__ == "__main__":
    main()
