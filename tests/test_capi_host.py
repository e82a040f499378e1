"""CPU-side checks of the product library: it loads, exports everything include/qoaudio_amd.h declares,
its host logic (sizes, validation order, stream walking) agrees with the oracle, and -- without a GPU --
its compute entry points fail loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rng_pcm

HEADER = os.path.join(ROOT, "include", "qoaudio_amd.h")


@pytest.fixture(scope="module")
def L():
    from qoaudio_amd import _lib
    return _lib.lib()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qoa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(L):
    from qoaudio_amd import _lib
    decl = declared_functions()
    assert len(decl) >= 30
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    for name in decl:
        assert re.search(rf" T {name}\b", exported), f"{name} declared in qoaudio_amd.h but not exported"
        getattr(L, name)
    assert sorted(_lib.SYMBOLS) == decl, "qoaudio_amd/_lib.py SYMBOLS out of sync with the header"


def test_library_has_gfx950_code_object():
    from qoaudio_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"qoa_encode_kernel" in blob and b"qoa_decode_kernel" in blob


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under qoaudio_amd/ may import, link or call it."""
    from qoaudio_amd import _lib
    for dirpath, _, files in os.walk(os.path.join(ROOT, "qoaudio_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("import oracle", "from oracle", "liboracle", "oq_", "oracle/", "oracle.py"):
                    assert needle not in text, (f, needle)
    assert b"oq_encode" not in open(_lib.LIB_PATH, "rb").read()


def test_encoded_size_and_frame_size(L, oracle):
    assert L.qoa_encoded_size(2, 2394122) == 1934040  # the fixture's length, src/lib.rs:374-378
    assert L.qoa_encoded_size(1, 1310720) == 530440 and L.qoa_encoded_size(2, 1310720) == 1058824
    for ch in range(1, 9):
        for n in (1, 19, 20, 21, 5119, 5120, 5121, 1310720, 28800000):
            assert L.qoa_encoded_size(ch, n) == oracle.encoded_size(ch, n)
        for slices in (1, 2, 255, 256):
            assert L.qoa_frame_size(ch, slices) == 8 + 16 * ch + 8 * slices * ch
    assert L.qoa_frame_size(8, 256) == 16520


def test_validation_order_without_touching_the_gpu(L):
    from qoaudio_amd import EncodeError, QoaDesc, QoaEncoder
    for desc, kind in ((QoaDesc(0, 44100, 1000), "InvalidChannels"), (QoaDesc(9, 44100, 1000), "InvalidChannels"),
                       (QoaDesc(1, 0, 1000), "InvalidSampleRate"), (QoaDesc(1, 44100, 0), "InvalidSamples"),
                       (QoaDesc(0, 0, 0), "InvalidChannels"), (QoaDesc(8, 0, 0), "InvalidSampleRate")):
        with pytest.raises(EncodeError) as e:
            QoaEncoder(desc)
        assert e.value.kind == kind
    assert str(EncodeError(1)) == "Invalid number of channels (must be 1-8)"  # src/lib.rs:900-902
    assert str(EncodeError(2)) == "Invalid sample rate (must be > 0)"
    assert str(EncodeError(3)) == "Invalid number of samples (must be > 0)"


def test_decode_error_strings(L):
    from qoaudio_amd import DecodeError
    assert str(DecodeError(1)) == "File is not a qoa file"  # src/lib.rs:880-883
    assert str(DecodeError(2)) == "File has no samples"
    assert str(DecodeError(3)) == "File has invalid frame header"
    assert str(DecodeError(4)) == "Incompatible frame header"
    assert str(DecodeError(5)).startswith("IO error: ")


def test_plan_validation(L):
    from qoaudio_amd import EncodeError, QoaDesc
    from qoaudio_amd import batch
    with pytest.raises(EncodeError) as e:
        batch.EncodePlan([(0, 0, QoaDesc(2, 44100, 100)), (200, 1000, QoaDesc(0, 44100, 100))])
    assert e.value.kind == "InvalidChannels"
    with pytest.raises(EncodeError) as e:
        batch.EncodePlan([(0, 4, QoaDesc(2, 44100, 100))])  # out_offset not 8-byte aligned
    assert e.value.kind == "IoError"


def test_no_gpu_means_loud_io_error(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    from qoaudio_amd import DecodeError, EncodeError, QoaDesc, encode_all, decode_all
    with pytest.raises(EncodeError) as e:
        encode_all(np.zeros(100, np.int16), QoaDesc(1, 44100, 100))
    assert e.value.kind == "IoError" and "IO error: " in str(e.value)


def scan(L, data):
    from qoaudio_amd import _lib
    info = _lib.StreamInfo()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    rc = L.qoa_stream_scan(buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size, C.byref(info))
    assert rc == 0
    return info


def oracle_walk(oracle, data):
    """(error raised by new or None, terminal error or None, #FrameHeader items, #Sample items) per the oracle."""
    try:
        dec = oracle.OracleDecoder(data)
    except oracle.OracleError as e:
        return e.kind, None, 0, 0
    streaming = dec.mode() == ("Streaming",)
    frames = samples = 0
    term = None
    first = True
    for kind, val in dec:
        if kind == "Sample":
            samples += 1
        elif kind == "FrameHeader":
            if first and streaming:
                assert val == (0, 0, 0)  # the default header, src/lib.rs:146-147,:674-678
            else:
                frames += 1
        else:
            term = val
        first = False
    return None, term, frames, samples


ERR = {0: None, 1: "NotQoaFile", 2: "NoSamples", 3: "InvalidFrameHeader", 4: "IncompatibleFrame", 5: "IoError"}


def check_scan(L, oracle, data):
    info = scan(L, data)
    new_err, term, frames, samples = oracle_walk(oracle, data)
    if new_err is not None:
        assert info.in_new == 1 and ERR[info.terminal] == new_err, (new_err, info.terminal)
    else:
        assert info.in_new == 0
        assert ERR[info.terminal] == term
        assert (info.n_frames, info.n_samples) == (frames, samples)


def test_stream_scan_matches_reference_reader_on_truncations(L, oracle):
    x = rng_pcm(11, 5200 * 2)
    enc = oracle.encode_all(x, 2, 32000, 5200)  # two frames: 5120 + 80 sample-times
    for cut in list(range(0, 80)) + list(range(len(enc) - 140, len(enc) + 1)) + [4136 + 8 + k for k in range(-9, 60)]:
        check_scan(L, oracle, enc[:cut])
    check_scan(L, oracle, enc + b"\0" * 5)  # trailing partial header bytes = clean EOF
    check_scan(L, oracle, enc + enc[8:])    # concatenated frames keep decoding


def test_stream_scan_matches_reference_reader_on_bad_headers(L, oracle):
    x = rng_pcm(12, 300 * 3)
    enc = bytearray(oracle.encode_all(x, 3, 8000, 300))
    def patched(off, val):
        b = bytearray(enc)
        b[off:off + len(val)] = val
        return bytes(b)
    cases = [
        patched(8, b"\0"),                    # channels = 0
        patched(9, b"\0\0\0"),                # sample rate = 0
        patched(14, (8 + 48).to_bytes(2, "big")),   # frame_size <= header + lms
        patched(14, (8 + 48 + 8 * 4).to_bytes(2, "big")),  # slices not a multiple of channels
        patched(14, (8 + 16 + 8 * 257).to_bytes(2, "big")),  # > 256 slices per channel (with ch patched below)
        patched(12, (100).to_bytes(2, "big")),   # fewer samples than the slices present: reader resyncs early
        patched(12, (320).to_bytes(2, "big")),   # more samples than slices present: runs into EOF
        patched(4, b"\0\0\0\0"),              # samples = 0 -> streaming mode
        patched(0, b"QOAF"),
    ]
    cases.append(bytes(enc) + patched(8, b"\2")[8:])     # second frame with different channel count (fixed mode)
    cases.append(patched(4, b"\0\0\0\0") + patched(8, b"\1")[8:60])  # streaming mode, channel change, then cut
    for c in cases:
        check_scan(L, oracle, c)


def test_stream_scan_random_corruption(L, oracle):
    r = np.random.default_rng(99)
    x = rng_pcm(13, 700 * 2)
    enc = oracle.encode_all(x, 2, 44100, 700)
    for _ in range(300):
        b = bytearray(enc)
        for _ in range(int(r.integers(1, 4))):
            b[int(r.integers(0, 64))] = int(r.integers(0, 256))  # headers live in the first bytes
        cut = int(r.integers(0, len(b) + 1)) if r.random() < 0.5 else len(b)
        check_scan(L, oracle, bytes(b[:cut]))


def test_streaming_decoder_handle_without_frames_needs_no_gpu(L):
    from qoaudio_amd import FrameHeader, QoaDecoder, Streaming, DecodeError
    d = QoaDecoder.new_streaming()
    assert d.mode() == Streaming()
    assert d.total_duration() is None
    assert d.next() == FrameHeader(0, 0, 0)  # the default header comes first in streaming mode
    assert d.next() is None
    for data, kind in ((b"", "IoError"), (b"qoa", "IoError"), (b"riff0000", "NotQoaFile"),
                       (b"qoaf\0\0\0\1", "NoSamples"), (b"qoaf\0\0\0\1" + bytes(8), "InvalidFrameHeader")):
        with pytest.raises(DecodeError) as e:
            QoaDecoder(data)
        assert e.value.kind == kind


# ---- the kernel's arithmetic identities, checked exhaustively on the CPU ----

QUANT_TAB = np.array([7, 7, 7, 5, 5, 3, 3, 1, 0, 0, 2, 2, 4, 4, 6, 6, 6])
RECIP = np.array([65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32])
DEQUANT = np.array([[1, -1, 3, -3, 5, -5, 7, -7], [5, -5, 18, -18, 32, -32, 49, -49],
                    [16, -16, 53, -53, 95, -95, 147, -147], [34, -34, 113, -113, 203, -203, 315, -315],
                    [63, -63, 210, -210, 378, -378, 588, -588], [104, -104, 345, -345, 621, -621, 966, -966],
                    [158, -158, 528, -528, 950, -950, 1477, -1477], [228, -228, 760, -760, 1368, -1368, 2128, -2128],
                    [316, -316, 1053, -1053, 1895, -1895, 2947, -2947], [422, -422, 1405, -1405, 2529, -2529, 3934, -3934],
                    [548, -548, 1828, -1828, 3290, -3290, 5117, -5117], [696, -696, 2320, -2320, 4176, -4176, 6496, -6496],
                    [868, -868, 2893, -2893, 5207, -5207, 8099, -8099], [1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933],
                    [1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005],
                    [1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336]])


def test_kernel_quantiser_closed_form_is_exact_for_every_residual():
    """qoa_kernels.hip replaces the QUANT/DEQUANT table lookups (src/lib.rs:569-572) by
    level = min(|scaled|,7)&6, code = level - (scaled>>31), dequant = +-mag[level/2], with the products
    taken on the 24-bit multiplier.  Checked for EVERY residual the encoder can see and all 16 scalefactors
    (|residual| <= 32767 + 2^18, incl. the i32 wrap of qoa_div at sf 0/1, src/lib.rs:613-617)."""
    res = np.arange(-(2 ** 18) - 32768, 2 ** 18 + 32768 + 1, dtype=np.int64)
    sext24 = lambda v: ((v + 2 ** 23) % 2 ** 24) - 2 ** 23
    wrap32 = lambda v: ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
    for sf in range(16):
        # reference formulation
        n = wrap32(wrap32(res * RECIP[sf]) + 32768) >> 16
        scaled = n + np.sign(res) - np.sign(n)
        c = np.clip(scaled, -8, 8)
        code_ref = QUANT_TAB[c + 8]
        deq_ref = DEQUANT[sf][code_ref]
        # kernel formulation
        n_k = wrap32(sext24(res) * sext24(RECIP[sf]) + 32768) >> 16   # v_mad_i32_i24: low 32 bits of the 48-bit product
        scaled_k = n_k + np.clip(res, -1, 1) - np.clip(n_k, -1, 1)     # v_med3_i32
        sneg = scaled_k >> 31
        a = (scaled_k ^ sneg) - sneg
        l2 = np.minimum(a, 7) & 6
        mag = DEQUANT[sf][l2]                                           # v_perm_b32 over the 4 packed magnitudes
        deq_k = (mag ^ sneg) - sneg
        code_k = l2 - sneg
        assert np.array_equal(code_ref, code_k), sf
        assert np.array_equal(deq_ref, deq_k), sf


def test_cheap_quotient_form_is_exact_wherever_it_does_not_mark():
    """K1's cheap-quotient step (tools/gen_slice_asm.py: cheap_quotient_note) for EVERY residual the packed forms can see
    (|residual| <= 2^18 + 32768: i16 weights and history) and all 16 scalefactors: a1 = sgn'(v), q = n * a1 with n taken as
    the signed high word of the 32-bit product, level = med3(q, -1, 7) & -10, code = level - (v >> 31), dequantised = mag * a1.
    Wherever q >= 0 code and dequantised value equal the reference's (src/lib.rs:613-617, :569-572); q < 0 (the lane is
    marked) happens only at scalefactors 0 and 1, only with |v| >= 32768, and -- the premise of the marked-candidate
    bound -- the reference's own dequantised value is then at most 49 in magnitude."""
    res = np.arange(-(2 ** 18) - 32768, 2 ** 18 + 32768 + 1, dtype=np.int64)
    sext24 = lambda v: ((v + 2 ** 23) % 2 ** 24) - 2 ** 23
    sext16 = lambda v: ((v + 2 ** 15) % 2 ** 16) - 2 ** 15
    wrap32 = lambda v: ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
    marked_total = 0
    for sf in range(16):
        n = wrap32(wrap32(res * RECIP[sf]) + 32768) >> 16
        scaled = n + np.sign(res) - np.sign(n)
        code_ref = QUANT_TAB[np.clip(scaled, -8, 8) + 8]
        deq_ref = DEQUANT[sf][code_ref]
        # the kernel's instructions
        u = wrap32(sext24(res) * sext24(RECIP[sf]) + 32768)            # v_mad_i32_i24
        sn = res >> 31                                                  # v_ashrrev_i32 31
        a1 = sn | 1                                                     # v_or_b32 1
        q = wrap32(sext24(sext16(u >> 16)) * sext24(a1))                # v_mul_i32_i24_sdwa, src0 = sext(WORD_1)
        t = np.clip(q, -1, 7)                                           # v_med3_i32 q, -1, 7
        level = t & -10                                                 # v_and_b32 -10
        ok = q >= 0
        mag = DEQUANT[sf][np.where(ok, level, 0)]
        deq_k = wrap32(sext24(mag) * sext24(a1))                        # v_mul_i32_i24
        code_k = level - sn
        assert np.array_equal(code_ref[ok], code_k[ok]) and np.array_equal(deq_ref[ok], deq_k[ok]), sf
        marked = ~ok
        # a marked step's "level" is 0xfffffff6: OR-ed into a slice-word half it sets bits 31..4 (test below: the sign bit stays)
        assert np.array_equal(level[marked] & 0xffffffff, np.full(marked.sum(), 0xfffffff6))
        if sf >= 2:
            assert not marked.any(), sf
        else:
            assert marked.any() and np.abs(res[marked]).min() >= (32768 if sf == 0 else 229355), sf
            assert np.abs(deq_ref[marked]).max() <= 49
        marked_total += int(marked.sum())
    assert marked_total > 0


def test_cheap_quotient_mark_is_the_sign_bit_of_a_slice_word_half():
    """The cheap forms accumulate a slice-word half as acc = (acc << 3 | level) - (v >> 31) over 10 steps from 0
    (tools/gen_slice_asm.py: pk_step).  Unmarked steps only: the 30-bit string of codes, bits 31 and 30 clear.  A marked step
    ORs 0xfffffff6 in; whatever the other steps do, bit 31 is then set at the end -- the mark the election reads instead of
    round 3's v_or3 per step pair.  Every position of one marked step, random other steps, and several marked steps."""
    r = np.random.default_rng(11)
    for trial in range(4000):
        n_marked = trial % 4                                            # 0, 1, 2, 3 marked steps
        where = set(r.choice(10, n_marked, replace=False).tolist()) if n_marked else set()
        if trial < 10:
            where = {trial}                                             # every single position
        acc, codes = 0, []
        for i in range(10):
            neg = int(r.integers(0, 2))
            if i in where:
                level = 0xfffffff6
            else:
                level = int(r.integers(0, 4)) * 2
                codes.append(level + neg)
            acc = (((acc << 3) & 0xffffffff) | level)                   # v_lshl_or_b32
            acc = (acc + neg) & 0xffffffff                              # v_sub_u32 acc, acc, (v >> 31)
        assert bool(acc >> 31) == bool(where), (trial, where, hex(acc))
        if not where:
            want = 0
            for c in codes:
                want = want << 3 | c
            assert acc == want and acc < 1 << 30


def test_mul24_equals_wrapping_mul_inside_the_guard():
    """v_mul_i32_i24 returns the low 32 bits of sext24(a)*sext24(b): equal to the wrapping i32 product for
    every operand pair the fast path can produce (|w| < 2^23 at all 20 steps, |h| <= 2^15)."""
    r = np.random.default_rng(7)
    lim = 2 ** 23 - 20 * 896
    w = np.concatenate([r.integers(-lim - 17920, lim + 17920, 200000), [lim + 17919, -lim - 17919, 0, 1, -1]])
    h = np.concatenate([r.integers(-32768, 32768, 200000), [-32768, 32767, 0, 1, -1]])
    wrap32 = lambda v: ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
    sext24 = lambda v: ((v + 2 ** 23) % 2 ** 24) - 2 ** 23
    assert np.array_equal(wrap32(w * h), wrap32(sext24(w) * sext24(h)))
    assert np.array_equal(wrap32(w * w), wrap32(sext24(w) * sext24(w)))


def test_generated_slice_asm_is_current():
    """qoa_slice_asm.inc is generated (tools/gen_slice_asm.py) and committed: the committed text must be what the
    generator produces now, and the scheduled block body must respect the wait-state rules it was built under."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_slice_asm", os.path.join(root, "tools", "gen_slice_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    text, counts = g.render()
    assert text == open(g.INC_PATH).read()
    # independent re-check of the hazards on the final order of the block loop's straight-line body
    for body in ([l for l in g.pk_block() if not l.endswith(":")], g.pk_slice(),
                 [l for l in g.pk_block(spec=True) if not l.endswith(":")]):
        _check_wait_states(g, body)


def _check_wait_states(g, body):
    last_write = {}
    pos = 0
    for line in body:
        pos += 1
        if line.split()[0].startswith(("s_", "ds_")):
            if line.startswith("s_cbranch") or line.startswith("s_branch"):
                last_write.clear()
            if line.startswith("s_nop"):
                pos += int(line.split()[1])  # s_nop N = N + 1 wait states
            continue
        op, dst, src, acc = g._parse(line)
        for r in src:
            if r in last_write:
                wpos, wop = last_write[r]
                if wop.startswith("v_dot2") and not (op == wop and r in acc):
                    assert pos - wpos >= 4, line
                if wop.startswith("v_pk_"):
                    assert pos - wpos >= 2, line
                if "_dpp" in op and r in g._regs(line.split()[2].rstrip(",")):
                    assert pos - wpos >= 3, line
        for r in dst:
            if r in last_write and last_write[r][1].startswith("v_dot2") and not op.startswith("v_dot2"):
                assert pos - last_write[r][0] >= 5, line
            last_write[r] = (pos, op)


def test_multi_gpu_shards_equal_dist_shard_range():
    """qoa_multi_shard (the C ABI's block partition) == qoaudio_amd.dist.shard_range (the torch.distributed form)."""
    from qoaudio_amd import batch, dist
    for n in (0, 1, 7, 8, 1024, 8191, 8192):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                first, count = batch.multi_shard(n, world, r)
                assert (first, first + count) == dist.shard_range(n, r, world)
                covered += count
            assert covered == n


def test_multi_gpu_and_lazy_reader_fail_loudly_without_a_device():
    """No CPU fallback: on a box without a GPU qoa_multi_create returns IoError with a detail string; the lazy reader's
    checks that precede any decoding (magic, header length) give the reference's errors."""
    import torch
    import qoaudio_amd as q
    from qoaudio_amd import batch, _lib
    import io
    if not torch.cuda.is_available():
        with pytest.raises(q.EncodeError) as e:
            batch.MultiGpu([0])
        assert e.value.kind == "IoError" and _lib.lib().qoa_last_error_detail()
    for data, kind in ((b"riff\0\0\0\1" + bytes(64), "NotQoaFile"), (b"qo", "IoError"), (b"qoaf\0\0", "IoError"),
                       (b"qoaf\0\0\0\1", "NoSamples"), (b"qoaf\0\0\0\1" + bytes(8), "InvalidFrameHeader")):
        with pytest.raises(q.DecodeError) as e:
            q.QoaDecoder(io.BytesIO(data))
        assert e.value.kind == kind


_OOM_CHILD = r"""
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, {root!r})
from qoaudio_amd import _lib
L = _lib.lib()
n_frames = 3_000_000                                   # 1 channel, 1 slice of 20 samples per frame: 32 bytes each
frame = np.zeros(32, np.uint8)
frame[0] = 1; frame[1:4] = (0, 0xac, 0x44); frame[4:6] = (0, 20); frame[6:8] = (0, 32)
buf = np.empty(8 + 32 * n_frames, np.uint8)
buf[:4] = np.frombuffer(b"qoaf", np.uint8); buf[4:8] = np.frombuffer((20 * n_frames).to_bytes(4, "big"), np.uint8)
buf[8:] = np.tile(frame, n_frames)
info = _lib.StreamInfo()
assert L.qoa_stream_scan(buf.ctypes.data_as(C.c_void_p), 8 + 32 * 1000, C.byref(info)) == 0   # (warm: everything loaded)
vm = [int(l.split()[1]) * 1024 for l in open("/proc/self/status") if l.startswith("VmSize")][0]
soft0, hard0 = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (24 << 20), hard0))   # soft limit only; the frame table needs far more than 24 MB
rc = L.qoa_stream_scan(buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(info))
detail = L.qoa_last_error_detail().decode()
resource.setrlimit(resource.RLIMIT_AS, (soft0, hard0))
print("RC", rc, detail)
rc2 = L.qoa_stream_scan(buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(info))       # and the library still works
print("AGAIN", rc2, info.terminal)
"""


def test_no_exception_crosses_the_c_abi_when_host_memory_runs_out():
    """"never abort".  The host side keeps its tables in std::vector; with the address space capped, the walk of a
    stream of 3,000,000 frames cannot grow its frame table: the entry point must come back with the family's IoError and a
    reason -- not std::terminate -- and work again once memory is there.  (Child process: the cap must not hit pytest.)"""
    import sys
    out = subprocess.run([sys.executable, "-c", _OOM_CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["RC"].split(" ", 1) == ["5", "out of host memory"], out.stdout       # QOA_DECODE_IO_ERROR
    assert lines["AGAIN"].split()[0] == "0"


def test_every_int_entry_point_is_a_function_try_block():
    """... and that holds for every `int` entry point of the C ABI, mechanically: `extern "C" int qoa_x(...) try {{ ... }}
    catch (...) {{ return on_exception(...); }}` (one-line bodies that cannot allocate are exempt)."""
    src = open(os.path.join(ROOT, "qoaudio_amd", "csrc", "qoa_capi.cpp")).read()
    heads = re.findall(r'^extern "C" int (qoa_\w+)\([^{;]*\)\s*(try\s*)?\{(.*)$', src, flags=re.M)
    assert len(heads) >= 40
    for name, is_try, rest in heads:
        one_line = rest.rstrip().endswith("}")
        assert is_try or one_line, f"{name} can throw across the C ABI"
    # (qoa_decoder_next reports through its item, as Iterator::next does: its handler fills in Some(Err(IoError)))
    assert src.count("} catch (...) { return on_exception(QOA_") + 1 == sum(1 for _, t, _ in heads if t)
    assert "item->kind = QOA_ITEM_ERROR; item->error = rc;" in src
