"""Parity of the HIP path against the CPU oracle, bit for bit (integer work: no tolerance anywhere).
Every call goes through the C ABI (qoaudio_amd/api.py and batch.py are ctypes over libqoaudio_amd.so)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import rng_pcm

pytestmark = pytest.mark.gpu

LENGTHS = [1, 2, 19, 20, 21, 39, 40, 41, 45, 319, 320, 321, 339, 5119, 5120, 5121, 5140, 10241]


@pytest.mark.parametrize("channels", range(1, 9))
def test_encode_all_matches_oracle(gpu, oracle, channels):
    """fuzz-style inputs over the edge lengths SURVEY section 4 lists (1, 19/20/21, 5119/5120/5121, tails)."""
    for n in LENGTHS:
        for kind in ("mix", "white") if n > 100 else ("mix", "white", "loud", "square", "silence"):
            x = rng_pcm(7919 * channels + n, n * channels, kind)
            got = gpu.encode_all(x, gpu.QoaDesc(channels, 44100, n))
            want = oracle.encode_all(x, channels, 44100, n)
            assert got == want, (channels, n, kind)


@pytest.mark.parametrize("channels", [1, 2, 5, 8])
def test_decode_all_matches_oracle(gpu, oracle, channels):
    for n in LENGTHS:
        x = rng_pcm(31 * channels + n, n * channels, "mix")
        enc = oracle.encode_all(x, channels, 48000, n)
        d = gpu.decode_all(enc)
        ch, sr, want = oracle.decode_all(enc)
        assert (d.num_channels, d.sample_rate) == (ch, sr) == (channels, 48000)
        assert np.array_equal(d.samples, want), (channels, n)
        d2 = gpu.decode_all_capi(enc)
        assert np.array_equal(d2.samples, want)


def test_fixture_config1(gpu, oracle, fixture_bytes, fixture_pcm):
    """BASELINE config 1: fixture -> PCM -> QOA.  KAT-3 (decoded PCM hash), KAT-1 (fixed-point prefix)."""
    d = gpu.decode_all(fixture_bytes)
    assert (d.num_channels, d.sample_rate, d.samples.size) == (2, 44100, 2394122 * 2)
    assert hashlib.sha256(d.samples.astype("<i2").tobytes()).hexdigest() == \
        "2ed7d9a42d13659a6ff51b04728499d99ac99686326b87f215fdeabe05522619"
    assert np.array_equal(d.samples, fixture_pcm)
    enc = gpu.encode_all(d.samples, gpu.QoaDesc(2, 44100, 2394122))
    assert len(enc) == 1934040
    assert enc[:1_073_768] == fixture_bytes[:1_073_768]
    assert enc == oracle.encode_all(fixture_pcm, 2, 44100, 2394122)


def test_loud_signals_hit_the_wrapping_divide(gpu, oracle):
    """|residual| >= 32768 makes qoa_div wrap for scalefactor 0 (src/lib.rs:606-617): must match bit for bit."""
    for seed in range(4):
        x = rng_pcm(seed, 4000, "loud")
        assert gpu.encode_all(x, gpu.QoaDesc(1, 44100, 4000)) == oracle.encode_all(x, 1, 44100, 4000)
    x = np.tile(np.array([32767, -32768, -32768, 32767, 0, 32767], dtype=np.int16), 700)
    assert gpu.encode_all(x, gpu.QoaDesc(2, 44100, 2100)) == oracle.encode_all(x, 2, 44100, 2100)


def _oracle_with_state(oracle, channels, rate, samples, state):
    e = oracle.OracleEncoder(channels, rate, samples)
    for c in range(channels):
        for i in range(4):
            e.e.lms[c].history[i] = int(state[c][i])
            e.e.lms[c].weights[i] = int(state[c][4 + i])
    return e


@pytest.mark.parametrize("big", [2 ** 23 - 20 * 896 - 1, 2 ** 23 - 20 * 896, 2 ** 23, 2 ** 24 + 12345, 2 ** 30, -2 ** 31])
def test_huge_weights_take_the_generic_path(gpu, oracle, big):
    """Weights are unbounded i32 in principle; beyond the 24-bit guard the kernel must switch to full
    32-bit multiplies and still match the reference's wrapping arithmetic (src/lib.rs:800-816, :823-826)."""
    channels, n = 2, 900
    x = rng_pcm(big % 1000, n * channels, "mix")
    state = np.zeros((channels, 8), dtype=np.int64)
    state[0] = [100, -200, 300, -400, big, -big // 3, 77, -16384]
    state[1] = [-5, 6, -7, 8, 5, big, -(big // 2), 3]
    state = ((state + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int32)
    enc = gpu.QoaEncoder(gpu.QoaDesc(channels, 44100, n))
    enc.set_lms_state(state)
    fl, data = enc.encode_frame(x)
    ref = _oracle_with_state(oracle, channels, 44100, n, state)
    want, _ = ref.encode_frame(x)
    assert fl == n and data == want
    got_state = enc.lms_state()
    want_state = np.array([h + w for h, w in ref.lms_state()], dtype=np.int32)
    assert np.array_equal(got_state, want_state)


def test_streaming_encode_equals_one_shot(gpu, oracle, fixture_pcm):
    """src/lib.rs:1261-1297 on the first 12 frames of the fixture + a short tail frame; state is carried in
    device memory as full i32 (never through the 16-bit header image)."""
    n = 5120 * 12 + 777
    pcm = fixture_pcm[: n * 2]
    desc = gpu.QoaDesc(2, 44100, n)
    oneshot = gpu.QoaEncoder(desc).encode(pcm)
    enc = gpu.QoaEncoder(desc)
    streamed = bytearray()
    enc.write_header(streamed)
    off = 0
    while off < n:
        fl = min(5120, n - off)
        assert enc.encode_frame(pcm[off * 2:(off + fl) * 2], streamed) == fl
        off += fl
    assert bytes(streamed) == oneshot == oracle.encode_all(pcm, 2, 44100, n)
    for bad in (np.zeros(0, np.int16), np.zeros(3, np.int16), np.zeros(5121 * 2, np.int16)):  # :424-430
        with pytest.raises(gpu.EncodeError) as e:
            enc.encode_frame(bad)
        assert e.value.kind == "InvalidSamples"
    with pytest.raises(gpu.EncodeError) as e:  # :368-370
        gpu.QoaEncoder(desc).encode(pcm[:-2])
    assert e.value.kind == "InvalidSamples"


def test_config5_eight_channel_streaming(gpu, oracle):
    """BASELINE config 5 in miniature: 8 channels, 48 kHz, frame-at-a-time, LMS carried across frames."""
    frames, ch = 6, 8
    n = 5120 * frames
    x = rng_pcm(5, n * ch, "mix")
    enc = gpu.QoaEncoder(gpu.QoaDesc(ch, 48000, n))
    out = bytearray()
    enc.write_header(out)
    for f in range(frames):
        _, data = enc.encode_frame(x[f * 5120 * ch:(f + 1) * 5120 * ch])
        assert len(data) == 16520
        out += data
    assert bytes(out) == oracle.encode_all(x, ch, 48000, n)


@pytest.mark.parametrize("flags", ["default", "classic", "cpw1", "cpw2", "cpw2_classic", "cpw4", "cpw4_classic", "cpw4_classic_step32", "cpw4_step32", "generic", "step32", "step32_slices", "noblock32", "nospec", "exactq", "ldsq",
                                   "ldsq_cpw1", "nowide"])
def test_ragged_batch_all_paths(gpu, oracle, flags):
    """One plan with clips of different lengths AND channel counts; every chains-per-wavefront packing and the
    forced generic path must all produce the oracle's bytes."""
    from qoaudio_amd import batch
    fl = {"default": 0, "cpw1": batch.PLAN_CPW_1, "cpw2": batch.PLAN_CPW_2, "cpw4": batch.PLAN_CPW_4,   # cpw4: helper wavefronts
          "classic": batch.PLAN_NO_HELPER, "cpw2_classic": batch.PLAN_CPW_2 | batch.PLAN_NO_HELPER,
          "cpw4_classic": batch.PLAN_CPW_4 | batch.PLAN_NO_HELPER,
          "cpw4_classic_step32": batch.PLAN_CPW_4 | batch.PLAN_NO_HELPER | batch.PLAN_FORCE_32BIT_STEP,
          "cpw4_step32": batch.PLAN_CPW_4 | batch.PLAN_FORCE_32BIT_STEP | batch.PLAN_NO_BLOCK32,
          "generic": batch.PLAN_FORCE_GENERIC_PATH | batch.PLAN_CPW_4, "step32": batch.PLAN_FORCE_32BIT_STEP,
          "step32_slices": batch.PLAN_FORCE_32BIT_STEP | batch.PLAN_NO_BLOCK32, "noblock32": batch.PLAN_NO_BLOCK32,
          "nospec": batch.PLAN_NO_SPECULATION, "exactq": batch.PLAN_EXACT_QUOTIENT,
          "ldsq": batch.PLAN_LDS_TABLES | batch.PLAN_CPW_4,
          "ldsq_cpw1": batch.PLAN_LDS_TABLES | batch.PLAN_CPW_1, "nowide": batch.PLAN_NO_WIDE_FORM}[flags]
    shapes = [(2, 5300), (1, 41), (2, 20), (3, 999), (2, 12000), (1, 6000), (8, 333), (2, 1), (1, 5120), (2, 5121),
              (5, 777), (2, 640), (1, 10240), (2, 319)]
    clips, pcm_parts, pcm_off, out_off = [], [], 0, 0
    for i, (ch, n) in enumerate(shapes):
        x = rng_pcm(100 + i, n * ch, "mix" if i % 3 else "white")
        pcm_parts.append(x)
        clips.append((pcm_off, out_off, gpu.QoaDesc(ch, 44100 + i, n)))
        pcm_off += x.size
        out_off += gpu.encoded_size(ch, n)
    pcm = torch.from_numpy(np.concatenate(pcm_parts)).cuda()
    out = torch.zeros(out_off, dtype=torch.uint8, device="cuda")
    plan = batch.EncodePlan(clips, fl)
    assert plan.samples == pcm.numel() and plan.out_bytes == out_off
    plan.run(pcm, out)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for (po, oo, d), x in zip(clips, pcm_parts):
        want = oracle.encode_all(x, d.channels, d.sample_rate, d.samples)
        assert host[oo:oo + len(want)].tobytes() == want, (d, flags)
    # decode the batch back with the plan built from the descs, and with the plan built by walking the bytes
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.zeros_like(pcm)
    dplan.run(out, back)
    assert dplan.mismatches() == 0
    streams = [host[oo:oo + gpu.encoded_size(d.channels, d.samples)].tobytes() for (_, oo, d) in clips]
    splan = batch.DecodePlan.from_streams(streams, [oo for (_, oo, _) in clips], [po for (po, _, _) in clips])
    assert splan.status == [0] * len(clips)
    back2 = torch.zeros_like(pcm)
    splan.run(out, back2)
    torch.cuda.synchronize()
    for (po, oo, d), s in zip(clips, streams):
        _, _, want = oracle.decode_all(s)
        assert np.array_equal(back[po:po + want.size].cpu().numpy(), want)
        assert np.array_equal(back2[po:po + want.size].cpu().numpy(), want)


def test_decoder_item_order_matches_reference_iterator(gpu, oracle):
    """QoaItem sequence (FrameHeader, n*ch Samples, ..., None / Err) on well-formed, partial, truncated,
    streaming-mode and channel-changing streams, item for item against the oracle's iterator."""
    x2 = rng_pcm(1, 5200 * 2)
    enc2 = oracle.encode_all(x2, 2, 32000, 5200)
    x3 = rng_pcm(2, 130 * 3)
    enc3 = oracle.encode_all(x3, 3, 8000, 130)
    streaming = b"qoaf\0\0\0\0" + enc2[8:] + enc3[8:] + enc2[8:]      # channel count changes per frame
    short_n = bytearray(enc3)
    short_n[12:14] = (55).to_bytes(2, "big")                           # fewer samples than the slices present
    cases = [enc2, enc3, enc2[:-5], enc2[: 8 + 4136 + 30], enc2[: 8 + 4136 + 8 + 32 + 16 * 3 + 7],
             streaming, streaming[:-9], bytes(short_n), enc2 + enc3[8:]]
    for data in cases:
        ref = oracle.OracleDecoder(data)
        dec = gpu.QoaDecoder(data)
        m = dec.mode()
        assert (("Streaming",) if isinstance(m, gpu.Streaming) else ("FixedSamples", m.channels, m.sample_rate, m.samples)) \
            == ref.mode()
        while True:
            want = ref.next()
            try:
                got = dec.next()
            except gpu.DecodeError as e:
                got = ("Err", e.kind)
            if want is None:
                assert got is None
                break
            if want[0] == "Sample":
                assert got == gpu.Sample(want[1])
            elif want[0] == "FrameHeader":
                assert got == gpu.FrameHeader(*want[1])
                assert dec.current_frame_header() == got
            else:
                assert got == want
                break


class _ChunkReader:
    """io::Read stand-in: hands out `sizes[k % len]` bytes per call (short reads), optionally fails at byte `fail_at`."""

    def __init__(self, data, sizes, fail_at=None):
        self.data, self.sizes, self.fail_at, self.pos, self.calls = bytes(data), sizes, fail_at, 0, 0

    def read(self, n):
        if self.fail_at is not None and self.pos >= self.fail_at:
            raise OSError("reader failed")
        k = min(n, self.sizes[self.calls % len(self.sizes)], len(self.data) - self.pos)
        if self.fail_at is not None:
            k = min(k, self.fail_at - self.pos) or k
        self.calls += 1
        out = self.data[self.pos:self.pos + k]
        self.pos += len(out)
        return out


def _items(dec, gpu):
    out = []
    while True:
        try:
            it = dec.next()
        except gpu.DecodeError as e:
            out.append(("Err", e.kind))
            return out
        if it is None:
            return out
        out.append(("FrameHeader", (it.num_channels, it.sample_rate, it.num_samples_per_channel))
                   if isinstance(it, gpu.FrameHeader) else ("Sample", it.value))


def test_lazy_reader_item_order_and_position(gpu, oracle):
    """qoa_decoder_new_reader (QoaDecoder::new over io::Read, src/lib.rs:127-178, :668-692): the item sequence over a
    reader that returns 1..7 bytes per call equals the oracle's iterator over the same bytes -- well-formed, truncated
    inside a row, cut at a frame boundary, streaming mode with changing channel counts; the reader is pulled frame by frame
    (never further than the frame whose samples are being handed out), and into_inner() returns it there."""
    x2 = rng_pcm(21, 5200 * 2)
    enc2 = oracle.encode_all(x2, 2, 32000, 5200)      # two frames: 5120 + 80 sample-times
    x3 = rng_pcm(22, 130 * 3)
    enc3 = oracle.encode_all(x3, 3, 8000, 130)
    streaming = b"qoaf\0\0\0\0" + enc2[8:] + enc3[8:] + enc2[8:]
    for data in (enc2, enc3, enc2[:-5], enc2[: 8 + 4136], enc2[: 8 + 4136 + 3], streaming, streaming[:-9]):
        want = [w for w in oracle.OracleDecoder(data)]
        rd = _ChunkReader(data, [1, 7, 3, 5, 2, 6, 4])
        dec = gpu.QoaDecoder(rd)
        got = _items(dec, gpu)
        assert got == want, (len(data), got[:3], want[:3])
    # laziness: after new() only the file header and the first frame have been pulled; the second frame arrives when the
    # iterator has handed out the first one's last sample
    rd = _ChunkReader(enc2, [4096])
    dec = gpu.QoaDecoder(rd)
    first_frame_end = 8 + 4136
    assert rd.pos == first_frame_end
    assert isinstance(dec.next(), gpu.FrameHeader)
    for _ in range(5120 * 2):
        assert isinstance(dec.next(), gpu.Sample)
    assert rd.pos == first_frame_end
    assert dec.next() == gpu.FrameHeader(2, 32000, 80)
    assert rd.pos == len(enc2)
    inner = dec.into_inner()
    assert inner is rd and inner.pos == len(enc2)
    # a reader that FAILS (not: ends) -- in the middle of the second frame, and exactly at a frame boundary: the rows that
    # arrived are yielded, then IoError (src/lib.rs:206-216 treats only UnexpectedEof as a clean end)
    for fail_at, n_samples in ((8 + 4136 + 8 + 32 + 16 * 2 + 5, (5120 + 40) * 2), (8 + 4136, 5120 * 2)):
        got = _items(gpu.QoaDecoder(_ChunkReader(enc2, [7, 1, 64], fail_at=fail_at)), gpu)
        assert got[-1] == ("Err", "IoError")
        assert sum(1 for k, _ in got if k == "Sample") == n_samples
    # errors of new(): bad magic, nothing but the file header, reader failing inside the first 8 bytes
    for data, kind in ((b"riff" + enc2[4:], "NotQoaFile"), (enc2[:8], "NoSamples"), (enc2[:3], "IoError")):
        with pytest.raises(gpu.DecodeError) as e:
            gpu.QoaDecoder(_ChunkReader(data, [2]))
        assert e.value.kind == kind
    with pytest.raises(gpu.DecodeError) as e:
        gpu.QoaDecoder(_ChunkReader(enc2, [3], fail_at=5))
    assert e.value.kind == "IoError"


def test_reader_exceptions_are_io_errors_not_a_clean_end(gpu, oracle, tmp_path):
    """Whatever a Python reader raises -- not only OSError -- is an I/O error of the stream (io::Error -> DecodeError::IoError,
    src/lib.rs:889-893), never a silent truncation; read() returning None (a non-blocking reader without data) likewise.
    QoaDecoder.open() closes its file when the constructor fails and when the decoder is closed; decode_all(reader) drains
    the reader and decodes the stream in one piece (advisor findings, round 3)."""
    x = rng_pcm(8, 5200 * 2)
    enc = oracle.encode_all(x, 2, 32000, 5200)

    class Raises(_ChunkReader):
        def __init__(self, data, at, exc):
            super().__init__(data, [512])
            self.at, self.exc = at, exc

        def read(self, n=1 << 30):
            if self.pos >= self.at:
                if self.exc is None:
                    return None
                raise self.exc
            return super().read(min(n, self.at - self.pos))

    for exc in (ValueError("I/O operation on closed file"), RuntimeError("custom reader"), None):
        got = _items(gpu.QoaDecoder(Raises(enc, 8 + 4136, exc)), gpu)   # fails at the boundary behind the first frame
        assert got[-1] == ("Err", "IoError") and sum(1 for k, _ in got if k == "Sample") == 5120 * 2
        dec = gpu.QoaDecoder(Raises(enc, 8 + 4136 + 100, exc))
        with pytest.raises(gpu.DecodeError) as e:
            for _ in dec:
                pass
        assert e.value.kind == "IoError"
        assert isinstance(e.value.__cause__, type(exc) if exc is not None else BlockingIOError)
        with pytest.raises(gpu.DecodeError) as e:
            gpu.QoaDecoder(Raises(enc, 5, exc))                          # inside new()
        assert e.value.kind == "IoError" and e.value.__cause__ is not None
        with pytest.raises(gpu.DecodeError) as e:
            gpu.decode_all(Raises(enc, 100, exc))
        assert e.value.kind == "IoError"
    # open(): the file is closed when new() fails, and with the decoder
    bad = tmp_path / "bad.qoa"
    bad.write_bytes(b"riff" + enc[4:])
    import gc, warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", ResourceWarning)
        with pytest.raises(gpu.DecodeError):
            gpu.QoaDecoder.open(str(bad))
        gc.collect()
    good = tmp_path / "good.qoa"
    good.write_bytes(enc)
    dec = gpu.QoaDecoder.open(str(good))
    f = dec._owned
    assert f is not None and not f.closed
    n = sum(1 for it in dec if isinstance(it, gpu.Sample))
    assert n == 5200 * 2
    dec.close()
    assert f.closed
    dec = gpu.QoaDecoder.open(str(good))
    f = dec.into_inner()                                                 # handed over: the caller's to close
    assert not f.closed
    f.close()
    # decode_all over a file object: one drain, the same samples as over the bytes
    with open(good, "rb") as fh:
        d = gpu.decode_all(fh)
        assert fh.read() == b""
    assert d.num_channels == 2 and np.array_equal(d.samples, oracle.decode_all(enc)[2])


def test_streaming_decoder_decode_frame(gpu, oracle, fixture_bytes):
    """src/lib.rs:1034-1066, plus sample values and a multi-frame append."""
    dec = gpu.QoaDecoder.new_streaming()
    ref = oracle.OracleDecoder(streaming=True)
    assert isinstance(dec.mode(), gpu.Streaming)
    pos = 8
    for _ in range(3):
        fs = int.from_bytes(fixture_bytes[pos + 6:pos + 8], "big")
        got = dec.decode_frame(fixture_bytes[pos:pos + fs])
        want = ref.decode_frame(fixture_bytes[pos:pos + fs])
        assert got.size == 5120 * 2 and np.array_equal(got, want)
        pos += fs
    two = fixture_bytes[pos:pos + 2 * 4136]
    assert np.array_equal(dec.decode_frame(two), ref.decode_frame(two))
    with pytest.raises(gpu.DecodeError) as e:
        dec.decode_frame(fixture_bytes[pos:pos + 100])  # truncated frame
    assert e.value.kind == "IoError"


def test_decode_plan_status_for_bad_streams(gpu, oracle):
    from qoaudio_amd import batch
    x = rng_pcm(3, 700 * 2)
    good = oracle.encode_all(x, 2, 44100, 700)
    streams = [good, good[:-3], b"riff" + good[4:], good[:8], b"qoaf\0\0\0\1" + bytes(8)]
    offs_in, offs_out, a, b = [], [], 0, 0
    for s in streams:
        offs_in.append(a)
        offs_out.append(b)
        a += (len(s) + 7) // 8 * 8
        b += 1400
    plan = batch.DecodePlan.from_streams(streams, offs_in, offs_out)
    assert plan.status == [0, 5, 1, 2, 3]
    assert plan.samples_out == [1400, 680 * 2, 0, 0, 0]
    buf = np.zeros(a, np.uint8)
    for s, o in zip(streams, offs_in):
        buf[o:o + len(s)] = np.frombuffer(s, np.uint8)
    out = torch.zeros(b, dtype=torch.int16, device="cuda")
    plan.run(torch.from_numpy(buf).cuda(), out)
    torch.cuda.synchronize()
    _, _, want = oracle.decode_all(good)
    host = out.cpu().numpy()
    assert np.array_equal(host[:1400], want)
    assert np.array_equal(host[1400:1400 + 1360], want[:1360])


def test_decode_plan_detects_header_mismatch(gpu, oracle):
    from qoaudio_amd import batch
    x = rng_pcm(4, 6000 * 2)
    desc = gpu.QoaDesc(2, 44100, 6000)
    enc = np.frombuffer(oracle.encode_all(x, 2, 44100, 6000), np.uint8).copy()
    clips, _, per_enc = batch.uniform_clips(1, desc)
    enc[8 + 1] ^= 0x40  # corrupt the sample rate of frame 0
    plan = batch.DecodePlan.from_clips(clips)
    pcm = torch.zeros(12000, dtype=torch.int16, device="cuda")
    plan.run(torch.from_numpy(enc).cuda(), pcm)
    assert plan.mismatches() == 1
    with pytest.raises(gpu.DecodeError):
        batch.decode_batch(torch.from_numpy(enc).cuda().reshape(1, -1), desc)


def test_full_size_batch_properties(gpu, oracle):
    """BASELINE config 3 at FULL size (1024 stereo clips x 256 frames = 2.68 G samples, 5.4 GB of PCM in HBM),
    through size-independent properties: (1) every copy of a clip encodes to identical bytes (checksum of
    checksums over 1024 clips), (2) the distinct clips equal the oracle byte for byte, (3) decode(encode(x))
    equals the oracle's decode of the same bytes and every header verifies."""
    from qoaudio_amd import batch, synth
    n_clips, distinct, frames, ch = 1024, 8, 256, 2
    samples = frames * 5120
    base = synth.make_clips(distinct, samples, ch, device="cuda")
    pcm = base.repeat((n_clips // distinct, 1, 1)).contiguous()
    desc = gpu.QoaDesc(ch, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    assert per_enc == 1058824
    plan = batch.EncodePlan(clips)
    assert (plan.pcm_bytes, plan.out_bytes, plan.samples) == (5368709120, 1084235776, 2684354560)
    out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    plan.run(pcm, out)
    torch.cuda.synchronize()
    ref_block = out[:distinct]
    for rep in range(n_clips // distinct):
        assert torch.equal(out[rep * distinct:(rep + 1) * distinct], ref_block)
    host_pcm = base[:2].cpu().numpy()
    host_enc = out[:2].cpu().numpy()
    for i in range(2):
        assert host_enc[i].tobytes() == oracle.encode_all(host_pcm[i].reshape(-1), ch, 44100, samples)
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.empty_like(pcm)
    dplan.run(out, back)
    assert dplan.mismatches() == 0
    for rep in (0, n_clips // distinct - 1):
        assert torch.equal(back[rep * distinct:(rep + 1) * distinct], back[:distinct])
    _, _, want = oracle.decode_all(host_enc[1].tobytes())
    assert np.array_equal(back[1].cpu().numpy().reshape(-1), want)
    err = (back[:distinct].to(torch.float32) - base.to(torch.float32))
    assert float(err.pow(2).mean().sqrt()) < 500.0  # the reference's round-trip bound, src/lib.rs:1258


def test_mono_config2_slice(gpu, oracle):
    """BASELINE config 2 shape (mono, 1024 clips) at 4 frames per clip, every clip distinct."""
    from qoaudio_amd import batch, synth
    n_clips, samples = 1024, 4 * 5120
    pcm = synth.make_clips(n_clips, samples, 1, device="cuda", kind="white")
    enc = batch.encode_batch(pcm, 44100)
    torch.cuda.synchronize()
    hp, he = pcm.cpu().numpy(), enc.cpu().numpy()
    for i in (0, 1, 511, 1023):
        assert he[i].tobytes() == oracle.encode_all(hp[i].reshape(-1), 1, 44100, samples)
    back = batch.decode_batch(enc, gpu.QoaDesc(1, 44100, samples))
    _, _, want = oracle.decode_all(he[1023].tobytes())
    assert np.array_equal(back[1023].cpu().numpy().reshape(-1), want)


def _gpu_slice(state, smp, generic=0):
    import ctypes as C
    from qoaudio_amd import _lib
    st = np.asarray(state, np.int32)
    s = np.asarray(smp, np.int16)
    tot, fst, sl = np.zeros(16, np.uint64), np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    es = np.zeros((16, 8), np.int32)
    rc = _lib.lib().qoa_debug_encode_slice(st.ctypes.data, s.ctypes.data, len(s), generic, tot.ctypes.data,
                                           fst.ctypes.data, sl.ctypes.data, es.ctypes.data)
    assert rc == 0
    # fst[i] = (the device's own winner of the row << 32) | |err0| of candidate i; bit 63 = packed path declined
    wins = {(int(v) >> 32) & 15 for v in fst}
    assert len(wins) == 1, "every lane must agree on the winner"
    declined = {int(v) >> 63 for v in fst}
    assert len(declined) == 1
    _gpu_slice.declined = bool(declined.pop())
    return [int(v) for v in tot], [int(v) & 0xffffffff for v in fst], [int(v) for v in sl], es, wins.pop()


def test_kat2_slice_search_lane_by_lane(gpu, oracle):
    """KAT-2: all 16 candidates' total ranks of the fixture's first divergent slice, on the device."""
    state = [-2655, -1631, -965, -806, -5363, 11763, -15085, 16783]
    smp = [-857, -665, -532, -205, 97, 453, 1182, 2249, 3636, 4729, 5942, 7621, 8992, 9889, 10639, 11477, 12145,
           12013, 11809, 11685]
    want = [522117267, 525200405, 119383785, 43336318, 1587308, 1582747, 1887874, 1946592, 2513046, 3221590, 4272064,
            3964616, 6487859, 13440220, 24107150, 16356911]
    for generic in (0, 1, 2, 3):  # one-asm slice, rolled tail path, 20 separate step statements, i16-packed slice
        tot, aerr0, sl, es, dev_win = _gpu_slice(state, smp, generic)
        assert tot == want
        assert not _gpu_slice.declined
        win = min((tot[i], aerr0[i], i) for i in range(16))[2]
        assert (win, dev_win, hex(sl[win])) == (5, 5, "0x56484f1ce06cd14a")


def test_fixture_slices_teacher_forced_on_device(gpu, oracle, fixture_walk):
    """Device mirror of tests/test_oracle_golden.py::test_fixture_slices_teacher_forced: slices of the reference-held
    fixture, each from the entry state of the fixture's own decode trajectory, through the device functions of K1 in all
    four forms of qoa_debug_encode_slice.  All 1,055 slices where the fixture's older, penalty-free encoder and the crate's
    rule differ (the only reference-adjacent penalty-decided cases that exist) plus every 97th slice and the 2-sample
    tail.  Per slice: all 16 total ranks equal the oracle's, the device's winner is the oracle's, and the device's
    candidate for the FIXTURE's scalefactor is the fixture's slice word bit for bit."""
    import fixture_walk as W
    slices, _ = fixture_walk
    others = []
    for i, s in enumerate(slices):
        word, sf, _st, _ = oracle.encode_slice(s["entry"][0], s["entry"][1], s["pcm"], oracle.EXHAUSTIVE)
        if (word << ((20 - s["n"]) * 3)) & (2 ** 64 - 1) != s["word"]:
            others.append(i)
    assert len(others) == 1055
    other_set = set(others)
    picks = sorted(set(others) | set(range(0, len(slices), 97)) | {len(slices) - 1, len(slices) - 2})
    declined = 0
    for i in picks:
        s = slices[i]
        state = list(s["entry"][0]) + list(s["entry"][1])
        word, sf, (eh, ew), totals = oracle.encode_slice(s["entry"][0], s["entry"][1], s["pcm"], oracle.REF_STYLE)
        sf_fix = s["word"] >> 60
        assert (sf != sf_fix) == (i in other_set)
        for generic in (0, 1, 2, 3):
            tot, aerr0, sl, es, dev_win = _gpu_slice(state, s["pcm"], generic)
            assert tot == totals, (i, generic)
            assert dev_win == sf, (i, generic)
            assert sl[sf_fix] == s["word"], (i, generic)
            assert list(es[sf]) == eh + ew
            declined += generic == 3 and _gpu_slice.declined and i not in other_set
    # the packed form must carry practically all of the ordinary slices (real music): a silent fall-back would halve K1's
    # speed.  (The 1,055 penalty-decided ones sit at the weight-energy threshold by definition and mostly decline.)
    assert declined < (len(picks) - len(others)) // 10


def test_candidate_ranks_match_oracle_on_hard_slices(gpu, oracle):
    """err^2 reaches 2^32 and the 64-bit rank sums carry on full-scale input; weights/history at extremes."""
    r = np.random.default_rng(5)
    for trial in range(60):
        hist = [int(v) for v in r.integers(-32768, 32768, 4)]
        scale = [20000, 40000, 2 ** 20, 2 ** 22][trial % 4]
        w = [int(v) for v in r.integers(-scale, scale, 4)]
        kind = ["white", "loud", "mix"][trial % 3]
        length = 20 if trial % 5 else int(r.integers(1, 20))
        smp = rng_pcm(trial, length, kind)
        s, sf, (eh, ew), totals = oracle.encode_slice(hist, w, smp)
        for generic in (0, 1, 2, 3):
            tot, aerr0, sl, es, dev_win = _gpu_slice(hist + w, smp, generic)
            assert tot == totals, (trial, generic)
            win = min((tot[i], aerr0[i], i) for i in range(16))[2]
            assert win == sf == dev_win
            shift = (20 - length) * 3
            assert sl[win] == (s << shift) & (2 ** 64 - 1)
            assert list(es[win]) == eh + ew


def test_packed_fast_path_lane_by_lane(gpu, oracle):
    """The i16-packed step (v_dot2_i32_i16 / v_pk_mad_i16) against the oracle, candidate by candidate: ordinary
    weights, weights whose energy sits around the penalty threshold (sum w^2 ~ 0x8ff << 18, penalty active but the
    slice still provably exact), weights at the edge of the admitted window and beyond it (the slice must then be
    declined and re-done by the 32-bit path), history at the i16 extremes, residuals that wrap qoa_div at sf 0."""
    r = np.random.default_rng(11)
    accepted = declined = penalised = 0
    for trial in range(240):
        hist = [int(v) for v in r.integers(-32768, 32768, 4)]
        if trial % 8 == 7:
            hist = [[-32768, 32767][int(v)] for v in r.integers(0, 2, 4)]
        amp = [3000, 9000, 12280, 12400, 13500, 15900, 16500, 31000][trial % 8]
        w = [int(v) * int(sgn) for v, sgn in zip(r.integers(amp - amp // 8, amp + 1, 4), r.choice([-1, 1], 4))]
        if trial % 16 == 0:
            w = [0, 0, -8192, 16384]
        smp = rng_pcm(1000 + trial, 20, ["white", "loud", "mix", "square"][trial % 4])
        s, sf, (eh, ew), totals = oracle.encode_slice(hist, w, smp)
        tot, aerr0, sl, es, dev_win = _gpu_slice(hist + w, smp, 3)
        assert tot == totals, (trial, hist, w)
        assert dev_win == sf and sl[sf] == s and list(es[sf]) == eh + ew
        energy = sum(v * v for v in w)
        if energy >= 3875 << 18:
            assert _gpu_slice.declined
        declined += _gpu_slice.declined
        accepted += not _gpu_slice.declined
        tot32, _, sl32, es32, _ = _gpu_slice(hist + w, smp, 0)
        assert tot32 == tot and sl32 == sl and np.array_equal(es32, es)
        penalised += (not _gpu_slice.declined) and energy >= (0x900 << 18)
    assert accepted >= 100 and declined >= 40 and penalised >= 5, (accepted, declined, penalised)


def test_packed_path_is_taken_on_real_music(gpu, oracle, fixture_pcm):
    """The fast path's admission statistics on the reference's own fixture (real music, 2 x 239,414 slices): the
    bytes are the oracle's either way, and the share of slices handed to the 32-bit path must stay negligible --
    otherwise the kernel silently runs at half speed."""
    from qoaudio_amd import batch
    n = len(fixture_pcm) // 2
    clips = [(0, 0, gpu.QoaDesc(2, 44100, n))]
    pcm = torch.from_numpy(fixture_pcm).cuda()
    outs = []
    for flags in (batch.PLAN_DEFAULT, batch.PLAN_NO_SPECULATION, batch.PLAN_FORCE_32BIT_STEP, batch.PLAN_EXACT_QUOTIENT,
                  batch.PLAN_NO_PENALTY_FORM, batch.PLAN_NO_BLOCK32, batch.PLAN_FORCE_32BIT_STEP | batch.PLAN_NO_BLOCK32,
                  batch.PLAN_NO_HELPER, batch.PLAN_NO_WIDE_FORM):
        plan = batch.EncodePlan(clips, flags | batch.PLAN_COUNTERS)
        out = torch.zeros(plan.out_bytes, dtype=torch.uint8, device="cuda")
        plan.run(pcm, out)
        declined, unpacked, misses = plan.stats()
        cheap_declined, marked = plan.counters()[3:5]
        outs.append(out.cpu().numpy().tobytes())
        if flags == batch.PLAN_EXACT_QUOTIENT:
            assert cheap_declined == 0
            continue
        if flags in (batch.PLAN_DEFAULT, batch.PLAN_NO_HELPER, batch.PLAN_NO_WIDE_FORM):
            print(f"fixture: {declined} declined + {unpacked} unpacked, {misses} speculation misses, {cheap_declined} "
                  f"cheap-quotient declines of {2 * ((n + 19) // 20)} slices")
            # the cheap-quotient form must carry real music: a scalefactor-0 candidate's quotient wraps to the opposite sign
            # in 0.6 % of the fixture's slices (tools/k1lab/wrap_study.c) and nearly all of those are proven losers in place
            assert cheap_declined <= 2 * ((n + 19) // 20) // 50
            print(f"fixture: a candidate was marked in {marked} slices")
            # (every speculation miss is a cheap-form decline first); marked candidates are nearly all settled in place
            assert marked > 0 and 0 <= cheap_declined - misses <= marked // 10
            assert declined + unpacked <= 2 * ((n + 19) // 20) // 200
            assert 0 < misses <= 2 * ((n + 19) // 20) // 20  # tonal passages: a few per cent, never the bulk
        elif flags == batch.PLAN_NO_SPECULATION:
            assert misses == 0 and declined + unpacked <= 2 * ((n + 19) // 20) // 200
        elif flags == batch.PLAN_NO_PENALTY_FORM:
            assert plan.counters()[7] == 0 and 0 < misses <= 2 * ((n + 19) // 20) // 20
        elif flags == batch.PLAN_NO_BLOCK32:
            assert declined + unpacked <= 2 * ((n + 19) // 20) // 200
        else:
            assert (declined, unpacked, misses) == (0, 0, 0)
    assert all(o == outs[0] for o in outs[1:]) and outs[0] == oracle.encode_all(fixture_pcm, 2, 44100, n)


@pytest.mark.parametrize("gain", [1.5, 3.0, 8.0])
def test_cheap_quotient_on_loud_material(gpu, oracle, fixture_pcm, gain):
    """The cheap-quotient step form (level from |n|, sign from the residual) against its hard case: amplified, clipped music,
    where the scalefactor-0 candidate's quotient wraps to the opposite sign in 5-25 % of the slices (tools/k1lab/wrap_study.c).
    Marked candidates must be proven losers or the slice repeated with the exact qoa_div: bytes equal the oracle's and the
    exact-quotient form's, and the decline path is actually taken."""
    from qoaudio_amd import batch
    n = min(len(fixture_pcm) // 2, 40 * 5120)
    x = np.clip(np.rint(fixture_pcm[:2 * n].astype(np.float64) * gain), -32768, 32767).astype(np.int16)
    clips = [(0, 0, gpu.QoaDesc(2, 44100, n))]
    pcm = torch.from_numpy(x).cuda()
    outs, declines, marks = [], [], []
    for flags in (batch.PLAN_DEFAULT, batch.PLAN_EXACT_QUOTIENT, batch.PLAN_CPW_1, batch.PLAN_NO_BLOCK32):
        plan = batch.EncodePlan(clips, flags | batch.PLAN_COUNTERS)
        out = torch.zeros(plan.out_bytes, dtype=torch.uint8, device="cuda")
        plan.run(pcm, out)
        c = plan.counters()
        declines.append(c[3])
        marks.append(c[4])
        outs.append(out.cpu().numpy().tobytes())
    assert outs[0] == outs[1] == outs[2] == outs[3] == oracle.encode_all(x, 2, 44100, n)
    print(f"gain {gain}: a candidate marked in {marks[0]} slices, {declines[0]} of them declined, of {2 * n // 20} slices")
    assert declines[1] == 0 and marks[1] == 0
    assert marks[0] > 0 and marks[2] > 0        # the marking path runs ...
    if gain >= 3.0:
        assert declines[0] > 0                  # ... and so does the decline


def test_rank_ties_resolved_like_the_reference(gpu, oracle):
    """Ties on the total rank must go to the candidate that is first in (first-sample rank, sf) order
    (strict '<' at src/lib.rs:587, stable insertion sort at :536).  Tied slices are found by searching small
    signals with the oracle (about 1 % of short low-amplitude slices tie); the device's own winner must be
    the oracle's on every one of them."""
    r = np.random.default_rng(1)
    ties = first_rank_decides = 0
    for trial in range(20000):
        amp = int(r.integers(1, 60))
        length = 20 if trial % 3 else int(r.integers(1, 6))
        smp = r.integers(-amp, amp + 1, length).astype(np.int16)
        st = [int(v) for v in r.integers(-3, 4, 4)] + [int(v) for v in r.integers(-300, 300, 4)]
        s, sf, _, totals = oracle.encode_slice(st[:4], st[4:], smp)
        tied = [i for i in range(16) if totals[i] == min(totals)]
        if len(tied) < 2:
            continue
        ties += 1
        tot, aerr0, sl, es, dev_win = _gpu_slice(st, smp, 0)
        assert tot == totals
        assert dev_win == sf, (list(smp), st, totals)
        assert sf == min(tied, key=lambda i: (aerr0[i], i))
        first_rank_decides += sf != min(tied)
        if ties >= 120:
            break
    assert ties >= 100 and first_rank_decides >= 1


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pipeline_equals_encode_all(gpu, oracle, pinned):
    """qoa_encode_batch_host (time-sliced pipeline, LMS state carried across slices in device memory): uniform
    batch (2-D copies) and ragged mixed-channel batch (per-clip copies), several time slices each."""
    from qoaudio_amd import batch
    # uniform: 24 stereo clips of 7 frames + a tail, 3 frames per time slice
    n, samples, ch = 24, 7 * 5120 + 1234, 2
    desc = gpu.QoaDesc(ch, 32000, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n, desc)
    pcm = torch.from_numpy(np.stack([rng_pcm(500 + i, samples * ch) for i in range(n)]))
    out = torch.zeros(n * per_enc, dtype=torch.uint8)
    if pinned:
        pcm, out = pcm.pin_memory(), out.pin_memory()
    batch.encode_batch_host(pcm, clips, out, chunk_frames=3)
    for i in (0, 7, 23):
        assert out[i * per_enc:(i + 1) * per_enc].numpy().tobytes() == \
            oracle.encode_all(pcm[i].numpy().reshape(-1), ch, 32000, samples)
    # ragged: different lengths and channel counts, 2 frames per slice; some clips end slices earlier than others
    shapes = [(2, 5120 * 5 + 7), (1, 100), (3, 5120 * 2), (2, 5120 * 2 + 1), (8, 5120 * 3 + 19), (1, 5120 * 6), (2, 20)]
    rclips, parts, po, oo = [], [], 0, 0
    for i, (c, s) in enumerate(shapes):
        x = rng_pcm(900 + i, s * c, "mix" if i % 2 else "white")
        parts.append(x)
        rclips.append((po, oo, gpu.QoaDesc(c, 44100, s)))
        po += x.size
        oo += gpu.encoded_size(c, s)
    rp = torch.from_numpy(np.concatenate(parts))
    ro = torch.zeros(oo, dtype=torch.uint8)
    if pinned:
        rp, ro = rp.pin_memory(), ro.pin_memory()
    batch.encode_batch_host(rp, rclips, ro, chunk_frames=2)
    host = ro.numpy()
    for (p0, o0, d), x in zip(rclips, parts):
        want = oracle.encode_all(x, d.channels, d.sample_rate, d.samples)
        assert host[o0:o0 + len(want)].tobytes() == want, d


def _crafted_stream(rng, frames, samples_field=0):
    """Hand-built QOA bytes: frames = [(channels, rate, nspc, rows_present)], random LMS and slices.  The
    encoder cannot produce >8 channels or frames whose sample count disagrees with their slice count, but the
    reference decoder accepts both (src/lib.rs:227-244 validates only frame_size; :320-328 drives the reader)."""
    out = bytearray(b"qoaf" + int(samples_field).to_bytes(4, "big"))
    for ch, rate, nspc, rows in frames:
        fsize = 8 + 16 * ch + 8 * min(rows, 256) * ch
        out += bytes([ch]) + rate.to_bytes(3, "big") + nspc.to_bytes(2, "big") + (fsize & 0xffff).to_bytes(2, "big")
        out += rng.integers(0, 256, 16 * ch, dtype=np.uint8).tobytes()
        out += rng.integers(0, 256, 8 * rows * ch, dtype=np.uint8).tobytes()
    return bytes(out)


def _same_items(gpu, oracle, data):
    try:
        ref = oracle.OracleDecoder(data)
    except oracle.OracleError as want:  # QoaDecoder::new itself fails (src/lib.rs:127-169): same error kind
        with pytest.raises(gpu.DecodeError) as got:
            gpu.QoaDecoder(data)
        assert got.value.kind == want.args[0]
        return 0
    dec = gpu.QoaDecoder(data)
    n = 0
    while True:
        want = ref.next()
        try:
            got = dec.next()
        except gpu.DecodeError as e:
            got = ("Err", e.kind)
        if want is None:
            assert got is None
            return n
        if want[0] == "Sample":
            assert got == gpu.Sample(want[1]), n
        elif want[0] == "FrameHeader":
            assert got == gpu.FrameHeader(*want[1])
        else:
            assert got == want
            return n
        n += 1


def test_decoder_more_than_eight_channels(gpu, oracle):
    """Up to 255 channels are legal for the decoder (u8 in the frame header); they take the generic kernel."""
    r = np.random.default_rng(21)
    for ch, rows in ((9, 3), (16, 5), (31, 2), (255, 1)):
        nspc = rows * 20 - 7
        data = _crafted_stream(r, [(ch, 44100, nspc, rows), (ch, 44100, 20, 1)], samples_field=nspc + 20)
        n = _same_items(gpu, oracle, data)
        assert n == 2 + (nspc + 20) * ch
        d = gpu.decode_all(data)
        _, _, want = oracle.decode_all(data)
        assert d.num_channels == ch and np.array_equal(d.samples, want)


def test_decoder_frame_longer_than_its_slices(gpu, oracle):
    """num_samples_per_channel drives the reader, not frame_size (src/lib.rs:320-328): a frame may claim up to
    65535 samples (3277 slice rows) and the decoder keeps reading slices from whatever follows."""
    r = np.random.default_rng(22)
    # 1 channel, claims 65535 samples, header says 256 slices, stream really holds 3277 rows
    data = _crafted_stream(r, [(1, 8000, 65535, 3277)], samples_field=65535)
    assert _same_items(gpu, oracle, data) == 1 + 65535
    # 2 channels, claims 5000 samples but only 100 rows present -> 100 full rows, then IoError
    data = _crafted_stream(r, [(2, 8000, 5000, 100)], samples_field=5000)
    assert _same_items(gpu, oracle, data) == 1 + 100 * 20 * 2
    # streaming mode (samples = 0), frames of different channel counts and odd lengths back to back
    data = _crafted_stream(r, [(3, 8000, 41, 3), (1, 48000, 1, 1), (8, 22050, 5120, 256), (2, 8000, 39, 2)])
    assert _same_items(gpu, oracle, data) == 1 + 4 + 41 * 3 + 1 + 5120 * 8 + 39 * 2
    # extreme LMS weights in the header (all 0x8000 / 0x7fff) and all-ones slices
    hostile = bytearray(_crafted_stream(r, [(2, 44100, 5120, 256)], samples_field=5120))
    hostile[16:48] = bytes([0x80, 0x00, 0x7f, 0xff] * 8)
    hostile[48:] = b"\xff" * (len(hostile) - 48)
    assert _same_items(gpu, oracle, bytes(hostile)) == 1 + 5120 * 2


def test_batch_decode_mixes_packed_and_32bit_rows(gpu, oracle):
    """K2 decodes a row on i16 pairs when every chain of the wave passes the per-row weight guard and on 32-bit
    weights otherwise.  One plan, one wave: ordinary music, a frame whose weights are driven out of i16 for a while
    (all-ones slices from extreme header weights) and then left alone, random garbage -- every stream must still
    equal the oracle sample for sample."""
    from qoaudio_amd import batch
    r = np.random.default_rng(77)
    music = oracle.encode_all(rng_pcm(9, 3 * 5120 * 2), 2, 44100, 3 * 5120)
    hostile = bytearray(_crafted_stream(r, [(2, 44100, 5120, 256), (2, 44100, 5120, 256)], samples_field=10240))
    hostile[16:48] = bytes([0x80, 0x00, 0x7f, 0xff] * 8)
    body = 48 + 40 * 16
    hostile[48:body] = b"\xff" * (body - 48)                      # 40 rows of sf 15 / code 7: weights leave i16
    hostile[body:8 + 4136] = bytes(8 + 4136 - body)                 # then sf 0 / code 0: they stay where they are
    garbage = _crafted_stream(r, [(2, 8000, 5120, 256)] * 3, samples_field=15360)
    streams = [music, bytes(hostile), garbage, music]
    offs_in, offs_out, a, b = [], [], 0, 0
    for st in streams:
        offs_in.append(a)
        offs_out.append(b)
        a += (len(st) + 7) // 8 * 8
        b += 3 * 5120 * 2
    plan = batch.DecodePlan.from_streams(streams, offs_in, offs_out)
    assert plan.status == [0, 0, 0, 0]
    buf = np.zeros(a, np.uint8)
    for st, o in zip(streams, offs_in):
        buf[o:o + len(st)] = np.frombuffer(st, np.uint8)
    out = torch.zeros(b, dtype=torch.int16, device="cuda")
    plan.run(torch.from_numpy(buf).cuda(), out)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for st, o, n in zip(streams, offs_out, plan.samples_out):
        _, _, want = oracle.decode_all(st)
        assert n == len(want) and np.array_equal(host[o:o + n], want)


@pytest.mark.parametrize("channels", [3, 5, 6, 7])
def test_decode_channel_counts_that_do_not_divide_64(gpu, oracle, channels):
    """64 / channels frames share a wavefront and the remaining lanes idle; with more frames than one wavefront holds
    the idle lanes of wave k used to write into wave k+1's LDS image (first channels of its first frame)."""
    from qoaudio_amd import batch
    fpw = 64 // channels
    shapes = [(channels, 5120 * (2 * fpw + 1) + 77), (channels, 300), (channels, 5120 * fpw + 5)]
    clips, parts, po, oo = [], [], 0, 0
    for i, (ch, ns) in enumerate(shapes):
        x = rng_pcm(700 + i, ns * ch, "mix")
        parts.append(x)
        clips.append((po, oo, gpu.QoaDesc(ch, 32000, ns)))
        po += x.size
        oo += gpu.encoded_size(ch, ns)
    enc = np.concatenate([np.frombuffer(oracle.encode_all(x, d.channels, d.sample_rate, d.samples), np.uint8)
                          for (_, _, d), x in zip(clips, parts)])
    plan = batch.DecodePlan.from_clips(clips)
    back = torch.zeros(po, dtype=torch.int16, device="cuda")
    for _ in range(3):  # the overwrite was a race between wavefronts: give it a few chances
        plan.run(torch.from_numpy(enc).cuda(), back)
        torch.cuda.synchronize()
        host = back.cpu().numpy()
        for (p_, o_, d), x in zip(clips, parts):
            _, _, want = oracle.decode_all(enc[o_:o_ + gpu.encoded_size(d.channels, d.samples)].tobytes())
            assert np.array_equal(host[p_:p_ + len(want)], want), (d.channels, d.samples)


def test_hypothesis_encode_decode_parity(gpu, oracle):
    """Property test (hypothesis): arbitrary small int16 signals, 1-8 channels: GPU bytes == oracle bytes and
    GPU decode == oracle decode (the reference's fuzz target, with parity on top)."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.integers(1, 8), st.integers(1, 130), st.integers(0, 2 ** 32 - 1),
           st.sampled_from(["white", "mix", "loud", "square", "silence"]), st.integers(1, 2 ** 24 - 1))
    def check(channels, n, seed, kind, rate):
        x = rng_pcm(seed, n * channels, kind)
        enc = gpu.encode_all(x, gpu.QoaDesc(channels, rate, n))
        assert enc == oracle.encode_all(x, channels, rate, n)
        d = gpu.decode_all(enc)
        _, _, want = oracle.decode_all(enc)
        assert (d.num_channels, d.sample_rate) == (channels, rate) and np.array_equal(d.samples, want)

    check()


def test_sample_rate_above_24_bits_overlaps_the_channel_byte(gpu, oracle):
    """The encoder ORs the unmasked u32 rate into the frame header (src/lib.rs:448-449): rates >= 2^24 spill
    into the channel byte.  Reproduced bit for bit; such a stream then fails to decode in the reference too."""
    x = rng_pcm(9, 300)
    for rate in (2 ** 24, 2 ** 24 + 44100, 2 ** 32 - 1):
        assert gpu.encode_all(x, gpu.QoaDesc(1, rate, 300)) == oracle.encode_all(x, 1, rate, 300)


def test_encoder_state_carries_across_calls_like_the_reference(gpu, oracle):
    """QoaEncoder keeps its LMS state in `self` (src/lib.rs:89): a second encode() on the same object, or an
    encode() after encode_frame() calls, continues from the carried state.  Same on the device handle."""
    n, ch = 5120 + 300, 2
    a, b = rng_pcm(1, n * ch), rng_pcm(2, n * ch, "white")
    enc = gpu.QoaEncoder(gpu.QoaDesc(ch, 44100, n))
    ref = oracle.OracleEncoder(ch, 44100, n)
    assert enc.encode(a) == ref.encode(a)
    assert enc.encode(b) == ref.encode(b)          # second one-shot on the same encoder: state is NOT reset
    fl, frame = enc.encode_frame(a[: 700 * ch])
    want, _ = ref.encode_frame(a[: 700 * ch])
    assert (fl, frame) == (700, want)
    assert enc.encode(a) == ref.encode(a)
    assert np.array_equal(enc.lms_state(), np.array([h + w for h, w in ref.lms_state()], dtype=np.int32))
    assert gpu.QoaEncoder(gpu.QoaDesc(ch, 44100, n)).write_header() == ref.write_header() == b"qoaf" + n.to_bytes(4, "big")


def test_decoder_handle_flows(gpu, oracle):
    """decode_frame() on a fixed-mode decoder, next() resuming after None once more bytes arrive, a frame split
    across two decode_frame() calls, current_frame_header() tracking."""
    x = rng_pcm(3, 5200 * 2)
    stream = oracle.encode_all(x, 2, 32000, 5200)   # frames: 5120 + 80 sample-times
    f0_end = 8 + 4136
    # fixed-mode decoder built from the first frame only, second frame appended later
    dec, ref = gpu.QoaDecoder(stream[:f0_end]), oracle.OracleDecoder(stream[:f0_end])
    got = [dec.next() for _ in range(3)]
    assert got[0] == gpu.FrameHeader(2, 32000, 5120) and isinstance(got[1], gpu.Sample)
    want = [ref.next() for _ in range(3)]
    assert [g.value for g in got[1:]] == [w[1] for w in want[1:]]
    rest_gpu = dec.decode_frame(stream[f0_end:])     # drains frame 0's remaining samples + the appended frame
    rest_ref = ref.decode_frame(stream[f0_end:])
    assert np.array_equal(rest_gpu, rest_ref) and rest_gpu.size == 5200 * 2 - 2
    assert dec.next() is None and dec.current_frame_header() == gpu.FrameHeader(2, 32000, 80)
    # streaming decoder fed half a frame: IoError, as in the reference (the cursor is then at EOF)
    sd, sr = gpu.QoaDecoder.new_streaming(), oracle.OracleDecoder(streaming=True)
    with pytest.raises(gpu.DecodeError) as e:
        sd.decode_frame(stream[8:8 + 2000])
    assert e.value.kind == "IoError"
    with pytest.raises(oracle.OracleError):
        sr.decode_frame(stream[8:8 + 2000])
    # iterator resumes after None when the reader grows (Cursor<Vec<u8>> semantics)
    it = gpu.QoaDecoder.new_streaming()
    assert it.next() == gpu.FrameHeader(0, 0, 0) and it.next() is None
    tail = it.decode_frame(stream[f0_end:])
    _, _, full = oracle.decode_all(stream)
    assert np.array_equal(tail, full[5120 * 2:])
    assert it.next() is None


def test_degenerate_shapes(gpu, oracle):
    for ch, n in ((8, 1), (1, 1), (8, 20), (7, 21), (3, 5121), (8, 5120)):
        x = rng_pcm(ch * 100 + n, ch * n, "white")
        enc = gpu.encode_all(x, gpu.QoaDesc(ch, 1, n))  # sample_rate 1 is the smallest legal value
        assert enc == oracle.encode_all(x, ch, 1, n) and len(enc) == gpu.encoded_size(ch, n)
        d = gpu.decode_all(enc)
        _, _, want = oracle.decode_all(enc)
        assert np.array_equal(d.samples, want)


@pytest.mark.parametrize("pinned", [False, True])
def test_host_decode_pipeline_and_persistent_handle(gpu, oracle, pinned):
    """qoa_decode_batch_host / qoa_host_pipeline_* (decode half + persistent handle): uniform batches
    (2-D copies; full frames go through K2u, the short last frame through K2) and a ragged mixed-channel batch (per-clip
    copies), several time slices each, against the oracle's decoder; one pipeline handle reused for encode and decode of
    differently shaped batches; a damaged frame header is reported."""
    from qoaudio_amd import batch
    pipe = batch.HostPipeline()

    def buf(t):
        return t.pin_memory() if pinned else t

    # uniform: 40 stereo clips of 7 frames + a tail (40 * 7 = 280 full frames: 8 whole wavefronts for K2u + a rest)
    n, samples, ch = 40, 7 * 5120 + 1234, 2
    desc = gpu.QoaDesc(ch, 32000, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n, desc)
    pcm = buf(torch.from_numpy(np.stack([rng_pcm(500 + i, samples * ch) for i in range(n)])))
    enc = buf(torch.zeros(n * per_enc, dtype=torch.uint8))
    pipe.encode(pcm, clips, enc, chunk_frames=3)
    back = buf(torch.zeros(n * per_pcm, dtype=torch.int16))
    pipe.decode(enc, clips, back, chunk_frames=3)
    for i in (0, 13, 39):
        stream = enc[i * per_enc:(i + 1) * per_enc].numpy().tobytes()
        assert stream == oracle.encode_all(pcm[i].numpy().reshape(-1), ch, 32000, samples)
        _, _, want = oracle.decode_all(stream)
        assert np.array_equal(back[i * per_pcm:(i + 1) * per_pcm].numpy(), want)
    back2 = buf(torch.zeros(n * per_pcm, dtype=torch.int16))
    batch.decode_batch_host(enc, clips, back2, chunk_frames=2)  # the one-shot form, another slice size
    assert torch.equal(back2, back)
    # ragged, same handle again: different lengths and channel counts; some clips end slices earlier than others
    shapes = [(2, 5120 * 5 + 7), (1, 100), (3, 5120 * 2), (2, 5120 * 2 + 1), (8, 5120 * 3 + 19), (1, 5120 * 6), (2, 20)]
    rclips, parts, po, oo = [], [], 0, 0
    for i, (c, s) in enumerate(shapes):
        x = rng_pcm(900 + i, s * c, "mix" if i % 2 else "white")
        parts.append(x)
        rclips.append((po, oo, gpu.QoaDesc(c, 44100, s)))
        po += x.size
        oo += gpu.encoded_size(c, s)
    rp = buf(torch.from_numpy(np.concatenate(parts)))
    ro = buf(torch.zeros(oo, dtype=torch.uint8))
    pipe.encode(rp, rclips, ro, chunk_frames=2)
    rb = buf(torch.zeros(po, dtype=torch.int16))
    pipe.decode(ro, rclips, rb, chunk_frames=2)
    host_enc, host_back = ro.numpy(), rb.numpy()
    for (p0, o0, d), x in zip(rclips, parts):
        want = oracle.encode_all(x, d.channels, d.sample_rate, d.samples)
        assert host_enc[o0:o0 + len(want)].tobytes() == want, d
        _, _, dec = oracle.decode_all(want)
        assert np.array_equal(host_back[p0:p0 + x.size], dec), d
    # a damaged frame header (sample rate byte of clip 4's second frame) is detected on the device
    bad = ro.clone()
    o4 = rclips[4][1]
    bad[o4 + 8 + (8 + 16 * 8 + 8 * 256 * 8) + 2] ^= 0x40
    with pytest.raises(gpu.DecodeError) as e:
        pipe.decode(bad, rclips, rb, chunk_frames=2)
    assert e.value.kind == "InvalidFrameHeader"
    pipe.decode(ro, rclips, rb, chunk_frames=5)  # the handle is still good
    assert np.array_equal(rb.numpy()[:parts[0].size], oracle.decode_all(host_enc[:rclips[1][1]].tobytes())[2])


@pytest.mark.parametrize("channels", [1, 2, 3, 4, 5, 6, 7, 8])
def test_k2u_hostile_full_frames(gpu, oracle, channels):
    """K2u (the fast path for full 256-row frames; round 4: every channel count 1 .. 8 -- with 3 / 5 / 6 / 7 channels a wavefront
    holds floor(64 / channels) frames and its spare lanes mirror lanes of the last one) on streams no encoder writes: random LMS headers and
    random slices (weights race out of i16 and back), frames that start from extreme header weights and push them further
    with all-ones slices, quiet music in between -- enough full frames for several wavefronts of the fast path plus a rest
    that takes the general kernel.  Every stream must equal the oracle's decoder sample for sample: the generated row's
    slack guard may only ever decline a row, never mis-prove one."""
    from qoaudio_amd import batch
    r = np.random.default_rng(2024 + channels)
    fs = 8 + 16 * channels + 8 * 256 * channels
    music = oracle.encode_all(rng_pcm(9, 21 * 5120 * channels), channels, 44100, 21 * 5120)
    hostile = bytearray(_crafted_stream(r, [(channels, 44100, 5120, 256)] * 12, samples_field=12 * 5120))
    for f in range(12):
        base = 8 + f * fs
        hostile[base + 8:base + 8 + 16 * channels] = bytes([0x80, 0x00, 0x7f, 0xff] * (4 * channels))
        body = base + 8 + 16 * channels
        n_ff = (10 + 3 * f) * 8 * channels                       # 10..43 rows of sf 15 / code 7: weights leave i16
        hostile[body:body + n_ff] = b"\xff" * n_ff
        quiet = body + n_ff + (f % 3) * 40 * 8 * channels          # then stretches of sf 0 / code 0 where they stay put
        hostile[quiet:base + fs] = bytes(base + fs - quiet)
    garbage = _crafted_stream(r, [(channels, 8000, 5120, 256)] * 80, samples_field=80 * 5120)
    streams = [music, bytes(hostile), garbage, music]
    offs_in, offs_out, a, b = [], [], 0, 0
    for st, frames in zip(streams, (21, 12, 80, 21)):
        offs_in.append(a)
        offs_out.append(b)
        a += (len(st) + 7) // 8 * 8
        b += frames * 5120 * channels
    plan = batch.DecodePlan.from_streams(streams, offs_in, offs_out)
    assert plan.status == [0, 0, 0, 0]
    buf = np.zeros(a, np.uint8)
    for st, o in zip(streams, offs_in):
        buf[o:o + len(st)] = np.frombuffer(st, np.uint8)
    out = torch.zeros(b, dtype=torch.int16, device="cuda")
    plan.run(torch.from_numpy(buf).cuda(), out)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for st, o, n in zip(streams, offs_out, plan.samples_out):
        _, _, want = oracle.decode_all(st)
        assert n == len(want) and np.array_equal(host[o:o + n], want)


@pytest.mark.parametrize("channels", [1, 2, 3, 5, 6, 7, 8])
def test_k2u_partial_last_wavefront_beside_the_general_kernel(gpu, oracle, channels):
    """Uniform clips whose full frames do not fill whole wavefronts (K2u's last wavefront repeats its last frame in the spare
    frame slots: the same bytes to the same addresses) and that end in a short frame (the general kernel, launched BESIDE K2u
    on the plan's side stream and joined before the caller's stream goes on): the same plan run several times back to back
    into fresh buffers, a consumer kernel on the caller's stream right behind every run."""
    from qoaudio_amd import batch
    fpw = 64 // channels
    n_clips, frames = 3, fpw + 1 + (channels % 3)                # 3 * frames full frames: never a multiple of fpw here
    assert (n_clips * frames) % fpw != 0
    samples = frames * 5120 + 3082 - (3082 % 8)                  # + a short last frame; clip PCM stays on the 16-byte grid
    desc = gpu.QoaDesc(channels, 32000, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    assert (per_pcm * 2) % 16 == 0
    pcm = np.stack([rng_pcm(500 + 7 * channels + i, per_pcm, "mix") for i in range(n_clips)])
    enc = np.stack([np.frombuffer(oracle.encode_all(pcm[i], channels, 32000, samples), np.uint8) for i in range(n_clips)])
    want = np.stack([oracle.decode_all(enc[i].tobytes())[2] for i in range(n_clips)])
    d_enc = torch.from_numpy(enc).cuda()
    plan = batch.DecodePlan.from_clips(clips)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for rep in range(4):
            back = torch.full((n_clips, per_pcm), 12345, dtype=torch.int16, device="cuda")
            plan.run(d_enc, back, stream)
            copy = back.clone()                                   # on the caller's stream, right behind the run
            assert np.array_equal(copy.cpu().numpy(), want), (channels, rep)
    assert plan.mismatches() == 0


def test_decode_plan_with_a_pcm_pointer_off_the_16_byte_grid(gpu, oracle):
    """K2u stores 16-byte pieces at frame offsets that are multiples of 16; a caller may hand the plan a PCM sub-buffer whose
    base is not.  The plan then sends its uniform-frame table through the general kernel: same samples either way."""
    from qoaudio_amd import batch
    n, samples, ch = 40, 2 * 5120, 2   # 80 full stereo frames: two wavefronts of the fast path + a rest
    desc = gpu.QoaDesc(ch, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n, desc)
    pcm = np.stack([rng_pcm(3000 + i, samples * ch) for i in range(n)])
    enc = np.stack([np.frombuffer(oracle.encode_all(pcm[i], ch, 44100, samples), np.uint8) for i in range(n)])
    d_enc = torch.from_numpy(enc).cuda()
    plan = batch.DecodePlan.from_clips(clips)
    big = torch.zeros(n * per_pcm + 8, dtype=torch.int16, device="cuda")
    want = np.concatenate([oracle.decode_all(enc[i].tobytes())[2] for i in (0, 17, 39)])
    for shift in (0, 1, 3, 8):          # element offsets: 0 and 8 keep 16-byte alignment, 1 and 3 do not
        view = big[shift:shift + n * per_pcm]
        view.zero_()
        plan.run(d_enc, view)
        assert plan.mismatches() == 0
        host = view.cpu().numpy().reshape(n, -1)
        assert np.array_equal(np.concatenate([host[i] for i in (0, 17, 39)]), want), shift


@pytest.mark.parametrize("flags", ["helper", "classic"])
def test_many_small_workgroups_helper_and_classic(gpu, oracle, flags):
    """1,500 short stereo clips, one chain per wavefront: 3,000 compute wavefronts -- with helper wavefronts forced that is
    6,000 wavefronts in two-wave workgroups, several per SIMD, barriers and all; without, the classic four-wave workgroups.
    Both against the oracle (lengths ragged around the block and frame boundaries)."""
    from qoaudio_amd import batch
    fl = batch.PLAN_CPW_1 | (batch.PLAN_FORCE_HELPER if flags == "helper" else batch.PLAN_NO_HELPER)
    r = np.random.default_rng(5)
    lens = [int(v) for v in r.choice([1, 19, 20, 21, 319, 320, 321, 640, 5119, 5120, 5121, 5440], 1500)]
    clips, parts, po, oo = [], [], 0, 0
    for i, ns in enumerate(lens):
        x = rng_pcm(1000 + i, ns * 2, "mix" if i % 4 else "white")
        parts.append(x)
        clips.append((po, oo, gpu.QoaDesc(2, 44100, ns)))
        po += x.size
        oo += gpu.encoded_size(2, ns)
    pcm = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.zeros(oo, dtype=torch.uint8, device="cuda")
    batch.EncodePlan(clips, fl).run(pcm, out)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for (p_, o_, d), x in zip(clips[::7], parts[::7]):
        want = oracle.encode_all(x, 2, 44100, d.samples)
        assert host[o_:o_ + len(want)].tobytes() == want, d.samples
    # every clip: the two launches' bytes are compared by the other parametrisation through the oracle on a sample; here all
    # of them against a second run of the same plan (determinism under the barriers)
    out2 = torch.zeros_like(out)
    batch.EncodePlan(clips, fl).run(pcm, out2)
    assert torch.equal(out, out2)


def test_plan_packing_keeps_helper_wavefronts_available(gpu):
    """qoa_encode_plan_launch_info: a plan packs 4 chains per wavefront at every size (round 3's 1 / 2 per wavefront for small
    launches ran up to 2x slower with helper wavefronts: tools/k1lab/small_launch.py); launches of up to 512 compute wavefronts
    (2048 chains) report helper wavefronts, larger ones and QOA_PLAN_NO_HELPER plans do not; the QOA_PLAN_CPW_* flags
    override the packing."""
    from qoaudio_amd import batch
    for n_clips, ch in ((1, 2), (256, 2), (257, 2), (512, 2), (513, 2), (1024, 2), (1025, 2), (1024, 1), (512, 1), (300, 8), (1, 1)):
        clips = [(i * 40 * ch, i * gpu.encoded_size(ch, 40), gpu.QoaDesc(ch, 44100, 40)) for i in range(n_clips)]
        waves, cpw, lds, helper = batch.EncodePlan(clips).launch_shape()
        chains = n_clips * ch
        assert cpw == 4 and waves == (chains + 3) // 4 and lds == 0, (n_clips, ch, waves, cpw)
        assert helper == (waves <= 512), (n_clips, ch, waves, helper)
        assert batch.EncodePlan(clips, batch.PLAN_NO_HELPER).launch_shape()[3] is False
        assert batch.EncodePlan(clips, batch.PLAN_FORCE_HELPER).launch_shape()[3] is True
    clips = [(i * 80, i * gpu.encoded_size(2, 40), gpu.QoaDesc(2, 44100, 40)) for i in range(64)]
    assert batch.EncodePlan(clips, batch.PLAN_CPW_4).launch_info()[:2] == (32, 4)
    assert batch.EncodePlan(clips, batch.PLAN_CPW_1).launch_shape() == (128, 1, False, True)
    assert batch.EncodePlan(clips, batch.PLAN_CPW_2).launch_info()[:2] == (64, 2)


def test_wave_stats_tell_what_a_batch_cost(gpu, oracle):
    """QOA_PLAN_WAVE_STATS / qoa_encode_plan_wave_stats (round 5): every compute wavefront records, once, at its exit, the
    slices it ran per step form and its cycles -- the recording instantiation of K1, no atomics.  Same bytes as the default
    plan; ordinary music sits on the cheap-quotient form, music amplified into hard clipping reaches the 32-bit block statement
    (src/lib.rs:613-617, :823-826) and its slowest wavefront stays within the bound the record states; plans that force
    another packing are refused."""
    from qoaudio_amd import batch, synth
    n_clips, ch, frames = 96, 2, 8
    samples = frames * 5120
    desc = gpu.QoaDesc(ch, 44100, samples)
    clips, _, per_enc = batch.uniform_clips(n_clips, desc)
    music = synth.make_clips(n_clips, samples, ch, 44100, device="cuda:0")
    loud = (music.to(torch.int32) * 3).clamp_(-32768, 32767).to(torch.int16)
    for name, pcm in (("music", music), ("loud", loud)):
        out = torch.zeros((n_clips, per_enc), dtype=torch.uint8, device="cuda")
        batch.EncodePlan(clips).run(pcm, out)
        out2 = torch.zeros_like(out)
        for flags in (batch.PLAN_WAVE_STATS, batch.PLAN_WAVE_STATS | batch.PLAN_NO_HELPER):
            plan = batch.EncodePlan(clips, flags)
            plan.run(pcm, out2)
            plan.run(pcm, out2)  # (the records are the LAST run's, not a sum)
            ws = plan.wave_stats()
            assert torch.equal(out, out2), (name, flags)
            assert ws["wavefronts"] == n_clips * ch // 4
            total = frames * 256
            mix = ws["slowest_wavefront"]["form_mix"]
            assert ws["slices"]["blocks"] == ws["wavefronts"] * frames * 16 and mix["blocks"] == frames * 16
            done = sum(v for k, v in ws["slices"].items() if k != "blocks")
            assert ws["wavefronts"] * total <= done <= ws["wavefronts"] * total * 1.25, (name, ws)  # (declined attempts count twice)
            cps = ws["cycles_per_slice"]
            assert 1500 < cps["min"] <= cps["p50"] <= cps["max"] < 1.1 * ws["bound_cycles_per_slice"], (name, cps)
            if name == "music":
                assert ws["slices"]["cheap_quotient"] > 0.98 * ws["wavefronts"] * total and ws["slices"]["block_32bit"] == 0
            else:
                assert ws["slices"]["block_32bit"] > 0 and cps["max"] > 1.2 * cps["min"]
        k = n_clips // 2
        assert out[k].cpu().numpy().tobytes() == oracle.encode_all(pcm[k].cpu().numpy().reshape(-1), ch, 44100, samples)
    with pytest.raises(gpu.EncodeError):
        batch.EncodePlan(clips, batch.PLAN_WAVE_STATS | batch.PLAN_CPW_2)
    with pytest.raises(gpu.EncodeError):
        batch.EncodePlan(clips).wave_stats()


@pytest.mark.parametrize("channels", [2, 4, 8])
@pytest.mark.parametrize("n_streams", [3, 70])
def test_k2u_absolute_line_flush_at_every_offset(gpu, oracle, n_streams, channels):
    """K2u's absolute-line variant (round 5; 2 / 4 / 8 channels): full frames whose PCM is off the 128-byte grid but on the 8-byte grid are
    flushed as whole ABSOLUTE cache lines -- the frame's first and last line partial, 8 single bytes at either end where the
    address is no multiple of 16.  Streams of 2 .. 5 full frames (music, hostile headers + all-ones slices that take the 32-bit
    rows, random garbage) at PCM offsets that run through EVERY multiple of 4 bytes modulo 128 (offsets that are 4 mod 8 stay on
    K2u's write-back path, offsets 0 mod 128 on its streamed path: all three launches in one plan), packed so tightly that a
    frame's first / last line is shared with its neighbour stream's last / first; the output buffer is pre-filled with a
    pattern and every byte outside the streams must still hold it (no masked piece may leak).  Both block shapes (n_streams = 3:
    one wavefront per workgroup; 70: four, several workgroups).  Sample for sample against the oracle's decoder
    (src/lib.rs:291-330)."""
    from qoaudio_amd import batch
    r = np.random.default_rng(77 + n_streams + 1000 * channels)
    fs = 8 + 16 * channels + 8 * 256 * channels
    streams, n_frames = [], []
    music = oracle.encode_all(rng_pcm(31, 5 * 5120 * channels), channels, 44100, 5 * 5120)
    for k in range(n_streams):
        nf = 2 + k % 4
        if k % 3 == 0:
            st = oracle.encode_all(rng_pcm(100 + k, nf * 5120 * channels), channels, 44100, nf * 5120)
        elif k % 3 == 1:
            st = bytearray(_crafted_stream(r, [(channels, 44100, 5120, 256)] * nf, samples_field=nf * 5120))
            for f in range(nf):
                base = 8 + f * fs
                st[base + 8:base + 8 + 16 * channels] = bytes([0x80, 0x00, 0x7f, 0xff] * (4 * channels))
                body = base + 8 + 16 * channels
                st[body:body + 30 * 8 * channels] = b"\xff" * (30 * 8 * channels)
            st = bytes(st)
        else:
            st = _crafted_stream(r, [(channels, 8000, 5120, 256)] * nf, samples_field=nf * 5120)
        streams.append(st)
        n_frames.append(nf)
    offs_in, offs_out, a, b = [], [], 0, 64
    for k, (st, nf) in enumerate(zip(streams, n_frames)):
        delta = (4 * (k * 7 + 1)) % 128 if k else 0               # every multiple of 4 modulo 128 over 32 streams
        b = b + (delta // 2 - b) % 64                               # int16 index with (2 b) % 128 == delta, no gap of a line
        offs_in.append(a)
        offs_out.append(b)
        a += (len(st) + 7) // 8 * 8
        b += nf * 5120 * channels
    total = b + 64
    plan = batch.DecodePlan.from_streams(streams, offs_in, offs_out)
    assert plan.status == [0] * n_streams
    buf = np.zeros(a, np.uint8)
    for st, o in zip(streams, offs_in):
        buf[o:o + len(st)] = np.frombuffer(st, np.uint8)
    pattern = ((np.arange(total) * 7 + 3) % 251 - 125).astype(np.int16)
    d_in = torch.from_numpy(buf).cuda()
    for rep in range(2):
        out = torch.from_numpy(pattern.copy()).cuda()
        assert out.data_ptr() % 128 == 0
        plan.run(d_in, out)
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        covered = np.zeros(total, bool)
        for st, o, n in zip(streams, offs_out, plan.samples_out):
            _, _, want = oracle.decode_all(st)
            assert n == len(want)
            bad = np.nonzero(host[o:o + n] != want)[0]
            assert bad.size == 0, (o, (2 * o) % 128, bad[:8], n)
            covered[o:o + n] = True
        assert np.array_equal(host[~covered], pattern[~covered])
    assert plan.mismatches() == 0
