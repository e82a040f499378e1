"""Static check of the decode kernels' code objects (CPU only: disassembles the gfx950 ELF inside libqoaudio_amd.so).

K2 and K2u fetch their slices with asynchronous global loads whose results are consumed rows later.  If the
destination were an ordinary asm output, hipcc would be free to copy or spill it between the load and its wait -- it
believes the value is there when the asm statement ends -- and the kernel would decode garbage without any error.

K2 (general kernel): the loads target AccVGPRs, which hipcc does not allocate in these kernels, and only the statement that
has waited for them moves them on.  Pinned down on the binary that ships:
  * the only instructions that touch an AccVGPR are the slice loads (destination) and `v_accvgpr_read_b32` (source);
  * every reader is preceded, in its basic block, by an `s_waitcnt vmcnt(..)` with no slice load in between.

K2u (uniform kernel, round 3): the loads are LDS-DMA (`global_load_lds_dwordx4`): no register holds a load in flight and
the kernel uses no AccVGPR at all.  What must hold instead:
  * M0 (the DMA's LDS destination) is written in the two instructions before every DMA load;
  * WAR: between the last ordinary LDS read and a DMA load there is an `s_waitcnt lgkmcnt(0)` (the stage slot is being
    overwritten: the reads of its old contents must have returned);
    (The RAW side -- the stage is read only behind the unit top's `s_waitcnt vmcnt` -- is by construction in `unit_top` and
    covered by the bit-exact GPU tests: in a disassembly the stage reads cannot be told from the cold path's table reads,
    and the loop's layout order is not its execution order.)

No scratch (spill) instructions exist in any decode kernel.
"""
import os
import re
import struct
import subprocess

import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "qoaudio_amd", "libqoaudio_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(tmp_path):
    """gfx950 ELF code objects of every translation unit bundled into the library's .hip_fatbin section."""
    fat = tmp_path / "fat.bin"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, str(tmp_path / "copy.so")])
    blob = fat.read_bytes()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                path = tmp_path / f"co_{len(out)}.elf"
                path.write_bytes(blob[pos + off:pos + off + size])
                out.append(path)
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def _decode_kernels(tmp_path):
    """{kernel symbol: [instruction text, ...]} for the qoa_decode_* kernels."""
    kernels = {}
    for co in _code_objects(tmp_path):
        text = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", str(co)], text=True)
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                # the asm statements' local labels (Lqoa_*) are ELF symbols too: they do not end the kernel they sit in
                if not m.group(1).startswith("Lqoa_"):
                    name = m.group(1)
                continue
            if name and "qoa_decode" in name and line.strip():
                ins = line.split("//")[0].strip()
                if ins:
                    kernels.setdefault(name, []).append(ins)
    return kernels


ACC = re.compile(r"\ba(\d+|\[\d+:\d+\])")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.skip("libqoaudio_amd.so not built")
    k = _decode_kernels(tmp_path_factory.mktemp("disasm"))
    assert any("qoa_decode_uniform_kernel" in n for n in k), sorted(k)
    assert any("qoa_decode_frames_kernel" in n for n in k), sorted(k)
    return k


def test_accvgprs_are_touched_only_by_the_prefetch_protocol(kernels):
    checked = 0
    for name, ins in kernels.items():
        if "qoa_decode_uniform_kernel" in name:
            assert not any(ACC.search(i.split(None, 1)[1]) for i in ins if " " in i), name  # K2u: no AccVGPR at all
            continue
        for i in ins:
            if not ACC.search(i.split(None, 1)[1] if " " in i else ""):
                continue
            op = i.split()[0]
            assert op in ("global_load_dwordx2", "v_accvgpr_read_b32"), (name, i)
            if op.startswith("global_load"):
                assert ACC.match(i.split(None, 1)[1]), (name, i)  # AccVGPR is the destination
            checked += 1
    assert checked > 100  # the protocol is in use in many template instances


def test_prefetched_slices_are_read_only_behind_a_wait(kernels):
    for name, ins in kernels.items():
        waited = False  # an s_waitcnt vmcnt with no AccVGPR load issued since
        for i in ins:
            op = i.split()[0]
            if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                waited = False  # (conservative: a reader must find its wait in its own straight-line run)
            if op == "s_waitcnt" and "vmcnt" in i:
                waited = True
            if op.startswith("global_load") and ACC.match(i.split(None, 1)[1]):
                waited = False
            if op == "v_accvgpr_read_b32" and ACC.search(i.split(None, 1)[1]):
                assert waited, (name, i)


def test_k2u_lds_dma_protocol(kernels):
    """The uniform kernels' slice loads: LDS-DMA with M0 set right before, lgkmcnt(0) between the last LDS read and the DMA that
    overwrites the stage."""
    seen = 0
    for name, ins in kernels.items():
        if "qoa_decode_uniform_kernel" not in name:
            continue
        lds_read_pending = False   # an LDS read issued, no lgkmcnt(0) since
        dma_unwaited = False       # a DMA load issued, no vmcnt wait since
        n_dma = n_stage = 0
        for k, i in enumerate(ins):
            op = i.split()[0]
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in i:
                    lds_read_pending = False
                if "vmcnt" in i:
                    dma_unwaited = False
            elif op.startswith("ds_read"):
                lds_read_pending = True
                n_stage += 1
            elif op == "global_load_lds_dwordx4":
                assert not lds_read_pending, (name, k, i)
                assert any(re.match(r"s_mov_b32 m0, s\d+", p) for p in ins[k - 2:k]), (name, ins[k - 3:k + 1])
                dma_unwaited = True
                n_dma += 1
        assert n_dma >= 2 and n_stage >= 1, (name, n_dma, n_stage)
        seen += 1
    assert seen >= 8  # 1 / 2 / 4 / 8 channels x the block sizes


def test_decode_kernels_do_not_spill(kernels):
    for name, ins in kernels.items():
        for i in ins:
            assert not i.startswith(("scratch_", "buffer_store", "buffer_load")), (name, i)


def test_k2u_flush_stores_issue_at_raised_priority(kernels):
    """K2u's line stores (1 / 2 / 4 channels) go out with the wavefront's issue priority raised (s_setprio 3 ... s_setprio 0:
    LABLOG.md K2u item 6) -- hipcc must not have moved the stores out of that window or dropped the pair."""
    seen = 0
    for name, ins in kernels.items():
        if "qoa_decode_uniform_kernel" not in name:
            continue
        m = re.search(r"qoa_decode_uniform_kernelILi(\d+)ELi(\d+)ELb([01])E", name)
        nch, absolute = int(m.group(1)), m.group(3) == "1"
        prio, raised_stores, plain_stores = 0, 0, 0
        for i in ins:
            op = i.split()[0]
            if op == "s_setprio":
                prio = int(i.split()[1])
            elif op.startswith(("s_cbranch", "s_branch")) and prio:
                prio = prio  # (the window is straight-line code; a branch inside it would show up as plain stores below)
            elif op == "global_store_dwordx4":   # streamed (nt: frames on the 128-byte grid) and write-back (off it) bursts alike
                if prio:
                    raised_stores += 1
                else:
                    plain_stores += 1
        # both store policies ship: streamed (nt) for frames on the 128-byte grid, write-back off it (LLVM once merged the
        # two bursts into one and dropped the nt hint with it)
        nt = sum(1 for i in ins if i.split()[0] == "global_store_dwordx4" and " nt" in i)
        wb = sum(1 for i in ins if i.split()[0] == "global_store_dwordx4" and " nt" not in i)
        if absolute:
            # the absolute-line variant (round 5): whole absolute lines only, every line store streamed and under a lane mask
            # (always issued: the static vmcnt count of the slice loads includes it) -- all but the frame's last line, which goes
            # out behind the loop, inside the raised-priority window
            # per round: phases x store slots x role groups (2 ch: 4 x 2 x 4, 4 ch: 2 x 3 x 2, 8 ch: 2 x 3 x 1), all at raised priority
            # (8 ch: no priority window at all); the frame's last line once per role group, behind the loop
            per_round, groups = {2: (32, 4), 4: (12, 2), 8: (6, 1)}[nch]
            assert nt == per_round + groups and wb == 0, (name, nt, wb)
            assert (raised_stores, plain_stores) == ((0, nt) if nch == 8 else (per_round, groups)), (name, raised_stores, plain_stores)
            for k, i in enumerate(ins):
                if i.split()[0] in ("global_store_dwordx4", "global_store_dwordx2"):
                    assert re.match(r"s_and_b64 exec, exec, s\[\d+:\d+\]", ins[k - 1]), (name, ins[k - 2:k + 2])
                    assert re.match(r"s_mov_b64 exec, s\[\d+:\d+\]", ins[k + 1]), (name, ins[k - 2:k + 2])
            seen += 1
            continue
        assert nt > 0 and wb > 0 and (nch not in (1, 2, 4, 8) or nt == wb), (name, nt, wb)
        if nch == 8:
            assert raised_stores == 0, name
        else:
            assert raised_stores > 0 and plain_stores == 0, (name, raised_stores, plain_stores)
            seen += 1
    assert seen >= 3


def test_k2u_masked_store_restores_exec(kernels):
    """3 / 5 / 6 / 7 channels: the last store-role group of a frame count that is no multiple of 8 runs under a lane mask inside
    one asm statement (`s_mov_b64 save, exec; s_and_b64 exec, exec, mask; global_store_dwordx4; s_mov_b64 exec, save`).  The
    statement declares SCC clobbered (advisor, round 4: `s_and_b64` writes it); what can be checked on the binary is that
    every narrowing of EXEC is undone from the SAME saved pair right behind the one store it covers, and that no SCC reader
    (`s_cselect`, `s_cbranch_scc*`, `s_addc`, `s_subb`) sits between the narrowing and the next SCC writer -- i.e. nobody
    consumes the SCC the statement left behind."""
    seen = 0
    scc_readers = ("s_cselect", "s_cbranch_scc", "s_addc_u32", "s_subb_u32", "s_cmov")
    for name, ins in kernels.items():
        m = re.search(r"qoa_decode_uniform_kernelILi(\d+)ELi(\d+)ELb0E", name)
        if not m or int(m.group(1)) not in (3, 5, 6, 7):
            continue
        n_masked = 0
        for k, i in enumerate(ins):
            if not re.match(r"s_and_b64 exec, exec, s\[\d+:\d+\]", i) or k + 1 >= len(ins):
                continue
            if ins[k + 1].split()[0] != "global_store_dwordx4":
                continue  # (hipcc's own control flow narrows EXEC too: `s_and_b64 exec, exec, ..` + `s_cbranch_execz`)
            save = re.match(r"s_mov_b64 (s\[\d+:\d+\]), exec", ins[k - 1])
            assert save, (name, ins[k - 2:k + 3])
            assert ins[k + 1].split()[0] == "global_store_dwordx4", (name, ins[k:k + 3])
            assert ins[k + 2] == f"s_mov_b64 exec, {save.group(1)}", (name, ins[k - 1:k + 3])
            for j in ins[k + 3:]:
                op = j.split()[0]
                assert not op.startswith(scc_readers), (name, k, j)
                if op.startswith(("s_cmp", "s_and_b", "s_or_b", "s_add_", "s_sub_", "s_lshl", "s_lshr", "s_bitcmp", "s_andn2", "s_xor_b")):
                    break  # SCC rewritten before anybody read it
                if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                    break  # (a branch that is not on SCC: the next block starts with its own compares)
            n_masked += 1
        assert n_masked > 0, name
        seen += 1
    assert seen >= 4
