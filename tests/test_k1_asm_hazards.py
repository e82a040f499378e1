"""Wait states inside K1's inline-asm texts (qoaudio_amd/csrc/qoa_slice_asm.inc).  The assembler pads nothing inside an asm
statement, the list scheduler of tools/gen_slice_asm.py is supposed to keep the gfx940/950 rules -- this test re-checks the
EMITTED text with its own small parser, text by text, so a generator edit cannot silently break one:
  * a v_dot2 result may be read by another instruction only as the accumulator of the same opcode, or 3 instructions later;
    it may be overwritten by a different VALU opcode 4 instructions later;
  * a packed result (v_pk_*, v_cvt_pk_*) needs 1 instruction before a VALU reads it;
  * a DPP instruction reads a VALU-written register 2 instructions later at the earliest.
Distances are counted in issued instructions of the straight-line text; a label forgets everything before it (paths join
there; the registers read right behind the loop labels come from LDS)."""
import os
import re

from conftest import ROOT

INC = os.path.join(ROOT, "qoaudio_amd", "csrc", "qoa_slice_asm.inc")
MODS = re.compile(r"\b(op_sel_hi|op_sel|row_ror|row_mask|bank_mask|offset|dst_sel|dst_unused|src0_sel|src1_sel|quad_perm):"
                  r"(\[[^\]]*\]|\S+)|\bclamp\b")


def texts():
    out, name = {}, None
    for line in open(INC):
        m = re.match(r"#define (\w+) \\", line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        m = re.match(r'\s+"(.*?)\\n\\t"', line)
        if m and name:
            out[name].append(m.group(1))
        elif not line.strip():
            name = None
    return out


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"sext\((.*)\)", tok)
    if m:
        tok = m.group(1)
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return [f"v{k}" for k in range(int(m.group(1)), int(m.group(2)) + 1)]
    return [tok] if re.fullmatch(r"v\d+", tok) or tok.startswith("%[") else []


def check(name, lines):
    last = {}   # register -> (opcode, position, accumulator-only?) of its last VALU write
    pos = 0
    for l in lines:
        if l.endswith(":"):
            last.clear()
            continue
        op, _, rest = l.partition(" ")
        ops = [regs(t) for t in MODS.sub("", rest).split(",")] if rest else []
        if op.startswith("v_"):
            wide = op in ("v_mad_i64_i32", "v_mad_u64_u32")
            dst = ops[0] + (ops[1] if wide else [])
            src = sum(ops[2 if wide else 1:], [])
            acc = ops[-1] if op.startswith("v_dot2") else []
            dpp = "_dpp" in op
            for r in src:
                if r in last:
                    pop, ppos = last[r]
                    gap = pos - ppos - 1
                    if pop.startswith("v_dot2") and not (op == pop and r in acc and src.count(r) == 1):
                        assert gap >= 3, (name, l, pop, gap)
                    if pop.startswith("v_pk_") or pop.startswith("v_cvt_pk"):
                        assert gap >= 1, (name, l, pop, gap)
                    if dpp:
                        assert gap >= 2, (name, l, pop, gap)
            for r in dst:
                if r in last and last[r][0].startswith("v_dot2") and not op.startswith("v_dot2"):
                    assert pos - last[r][1] - 1 >= 4, (name, l, last[r])
            for r in dst:
                last[r] = (op, pos)
        elif op.startswith("ds_read"):
            for r in regs(rest.split(",")[0]):
                last.pop(r, None)
        pos += 1 + (int(rest) if op == "s_nop" else 0)   # s_nop N = N + 1 wait states


def test_every_text_keeps_the_wait_states():
    t = texts()
    assert set(t) >= {"QOA_SLICE_ASM_TEXT", "QOA_SLICE_PK_ASM_TEXT", "QOA_BLOCK_PK_ASM_TEXT", "QOA_BLOCK_SPEC_ASM_TEXT",
                      "QOA_BLOCK_SPEC_LDSQ_ASM_TEXT", "QOA_BLOCK_CHEAP_ASM_TEXT", "QOA_BLOCK_PEN_ASM_TEXT"}
    for name, lines in t.items():
        assert len(lines) > 400, name
        check(name, lines)


def test_the_checker_sees_a_broken_text():
    good = ["v_dot2_i32_i16 v1, v2, v3, 0", "v_dot2_i32_i16 v1, v4, v5, v1", "v_mov_b32 v9, v8", "v_mov_b32 v10, v8",
            "v_mov_b32 v11, v8", "v_ashrrev_i32 v6, 13, v1", "v_pk_ashrrev_i16 v7, 15, v6 op_sel_hi:[0,1]", "v_mov_b32 v9, v8",
            "v_or_b32 v7, s95, v7", "s_nop 0", "v_mov_b32 v12, v8", "v_min_u32_dpp v7, v7, v7 row_ror:8 row_mask:0xf bank_mask:0xf"]
    check("good", good)
    import pytest
    for drop in (2, 7, 9):     # one filler behind the v_dot2 pair / behind the packed shift / in front of the DPP read
        with pytest.raises(AssertionError):
            check("bad", good[:drop] + good[drop + 1:])
