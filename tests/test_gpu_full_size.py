"""BASELINE configs 2 and 5 at their FULL sizes on the device (config 3 at full size: test_gpu_parity.py::
test_full_size_batch_properties; config 4 = config 3 per rank: test_gpu_multi.py).  Bit-exact against the oracle where the
CPU finishes in seconds, size-independent properties (replica equality, streamed == one-shot, verified headers,
round-trip bound) over everything else."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config2_full_size_mono_batch(gpu, oracle):
    """Config 2: 1024 mono 44.1 kHz clips x 256 frames (1,342,177,280 samples; 2.68 GB of PCM, 543 MB of QOA resident).
    16 distinct clips x 64 replicas: 4 clips byte for byte against the oracle (encode and decode), every replica equal to
    its original in both directions, every frame header verified, the reference's round-trip bound (src/lib.rs:1258)."""
    from qoaudio_amd import batch, synth
    n_clips, distinct, frames, ch = 1024, 16, 256, 1
    samples = frames * 5120
    base = synth.make_clips(distinct, samples, ch, device="cuda")
    pcm = base.repeat((n_clips // distinct, 1, 1)).contiguous()
    desc = gpu.QoaDesc(ch, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    assert per_enc == 530440  # cfg2 530,440 B per clip
    plan = batch.EncodePlan(clips)
    assert (plan.pcm_bytes, plan.out_bytes, plan.samples) == (2684354560, 543170560, 1342177280)
    out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    plan.run(pcm, out)
    torch.cuda.synchronize()
    for rep in range(n_clips // distinct):
        assert torch.equal(out[rep * distinct:(rep + 1) * distinct], out[:distinct])
    check = (0, 5, 10, 15)
    host_pcm = base.cpu().numpy()
    host_enc = out[:distinct].cpu().numpy()
    for i in check:
        assert host_enc[i].tobytes() == oracle.encode_all(host_pcm[i].reshape(-1), ch, 44100, samples)
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.empty_like(pcm)
    dplan.run(out, back)
    assert dplan.mismatches() == 0
    for rep in range(n_clips // distinct):
        assert torch.equal(back[rep * distinct:(rep + 1) * distinct], back[:distinct])
    host_back = back[:distinct].cpu().numpy()
    for i in check:
        _, _, want = oracle.decode_all(host_enc[i].tobytes())
        assert np.array_equal(host_back[i].reshape(-1), want)
    err = back[:distinct].to(torch.float32) - base.to(torch.float32)
    assert float(err.pow(2).mean().sqrt()) < 500.0


def test_config5_full_length_streaming(gpu, oracle):
    """Config 5: ONE 8-channel 48 kHz clip of 10 minutes (28,800,000 sample-times = exactly 5625 frames) through the
    frame-at-a-time API -- write_header + 5625 x encode_frame, the LMS state carried in device memory between the calls
    (src/lib.rs:404-439) -- against the oracle's one-shot encode_all of the same PCM over all 92,925,008 bytes, and the
    encoder's final LMS state (full i32, src/lib.rs:482) against the oracle encoder's."""
    from qoaudio_amd import synth
    ch, rate, frames = 8, 48000, 5625
    n = frames * 5120
    x = synth.make_clips(1, n, ch, rate, device="cuda")[0].cpu().numpy().reshape(-1)
    assert x.size == 230400000
    enc = gpu.QoaEncoder(gpu.QoaDesc(ch, rate, n))
    parts = [enc.write_header()]
    for f in range(frames):
        fl, data = enc.encode_frame(x[f * 5120 * ch:(f + 1) * 5120 * ch])
        assert fl == 5120 and len(data) == 16520
        parts.append(data)
    got = b"".join(parts)
    assert len(got) == 92925008 == gpu.encoded_size(ch, n)
    ref = oracle.OracleEncoder(ch, rate, n)
    want = ref.encode(x)
    assert got == want
    state = enc.lms_state().reshape(ch, 8)
    for c, (hist, weights) in enumerate(ref.lms_state()):
        assert list(state[c, :4]) == hist and list(state[c, 4:]) == weights


@pytest.mark.parametrize("channels", [3, 5, 6, 7])
def test_full_chip_decode_of_channel_counts_that_do_not_divide_a_wavefront(gpu, oracle, channels):
    """3 / 5 / 6 / 7 channels through K2u's four-wavefront workgroups (>= 1024 wavefronts: the small tests above take its
    one-wavefront form): mirror lanes, 4-row staging units with spare room per load instruction, masked last store group,
    XCD-contiguous block order, partial last wavefront.  Every clip against the general kernel K2 (the same plan into a PCM
    base 2 bytes off the dword grid), 24 clips spread over the batch against the oracle, every frame header verified."""
    from qoaudio_amd import batch, synth
    fpw = 64 // channels
    frames = 16
    n_clips = (1024 * fpw + 7 * frames) // frames + 1          # a little over 1024 wavefronts, and no multiple of fpw frames
    assert (n_clips * frames + fpw - 1) // fpw >= 1024 and (n_clips * frames) % fpw != 0
    samples = frames * 5120
    distinct = 48
    base = synth.make_clips(distinct, samples, channels, 48000, device="cuda", first_index=777)
    pcm = base.repeat(((n_clips + distinct - 1) // distinct, 1, 1))[:n_clips].contiguous()
    desc = gpu.QoaDesc(channels, 48000, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    batch.EncodePlan(clips).run(pcm, enc)
    dplan = batch.DecodePlan.from_clips(clips)
    buf = torch.full((n_clips * per_pcm + 8,), 4321, dtype=torch.int16, device="cuda")
    fast = buf[:n_clips * per_pcm]
    dplan.run(enc, fast)
    assert dplan.mismatches() == 0
    fast = fast.clone()
    slow = buf[1:1 + n_clips * per_pcm]                        # 2 bytes off the dword grid: the general kernel
    dplan.run(enc, slow)
    assert dplan.mismatches() == 0
    assert torch.equal(fast, slow)
    host = fast.view(n_clips, per_pcm).cpu().numpy()
    host_enc = enc.cpu().numpy()
    for i in np.linspace(0, n_clips - 1, 24).astype(int):
        _, _, want = oracle.decode_all(host_enc[i].tobytes())
        assert np.array_equal(host[i], want), (channels, int(i))


@pytest.mark.parametrize("channels,frames,tail", [(2, 16, 3082), (4, 8, 1217), (8, 4, 777)])
def test_full_chip_decode_of_packed_clips_with_a_short_last_frame(gpu, oracle, channels, frames, tail):
    """The reference fixture's shape (full frames + one short frame, src/lib.rs:1001-1031) packed back to back, a full chip of
    it: clip i's PCM starts `i * tail * channels * 2 mod 128` bytes past a 128-byte line, so the batch runs through every offset
    the channel count allows (stereo: all 16 multiples of 8), and ONE plan run takes K2u's absolute-line launch (off-grid full
    frames + the on-grid ones that do not fill the chip alone), the general kernel K2 on the side stream (the short last
    frames) and their shared cache lines (a clip's last line is its neighbour's first).  Every clip against the general
    kernel alone (the same plan into a PCM base 2 bytes off the dword grid), the pattern around the batch untouched, 24 clips
    spread over the batch against the oracle (src/lib.rs:291-330), every frame header verified."""
    from qoaudio_amd import batch, synth
    fpw = 64 // channels
    n_clips = (2048 + 64) * fpw // frames                       # > 2048 wavefronts of full frames
    samples = frames * 5120 + tail
    step = (samples * channels * 2) % 128
    assert step % 8 == 0 and step != 0
    distinct = 32
    base = synth.make_clips(distinct, samples, channels, 44100, device="cuda", first_index=4242)
    pcm = base.repeat(((n_clips + distinct - 1) // distinct, 1, 1))[:n_clips].contiguous()
    desc = gpu.QoaDesc(channels, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    assert per_pcm == samples * channels
    enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    batch.EncodePlan(clips).run(pcm, enc)
    del pcm
    dplan = batch.DecodePlan.from_clips(clips)
    pad = 64
    buf = torch.full((pad + n_clips * per_pcm + pad,), -4321, dtype=torch.int16, device="cuda")
    assert buf.data_ptr() % 128 == 0
    fast = buf[pad:pad + n_clips * per_pcm]                    # 128-byte aligned base: the absolute-line launch
    dplan.run(enc, fast)
    assert dplan.mismatches() == 0
    assert bool((buf[:pad] == -4321).all()) and bool((buf[pad + n_clips * per_pcm:] == -4321).all())
    fast = fast.clone()
    buf.fill_(-4321)
    slow = buf[pad + 1:pad + 1 + n_clips * per_pcm]            # 2 bytes off the dword grid: the general kernel
    dplan.run(enc, slow)
    assert dplan.mismatches() == 0
    assert torch.equal(fast, slow)
    host = fast.view(n_clips, per_pcm).cpu().numpy()
    host_enc = enc.cpu().numpy()
    seen = set()
    for i in np.linspace(0, n_clips - 1, 24).astype(int):
        seen.add((int(i) * step) % 128)
        _, _, want = oracle.decode_all(host_enc[i].tobytes())
        assert np.array_equal(host[i], want), (channels, int(i))
    assert len(seen) >= 128 // np.gcd(step, 128) // 2          # the sampled clips sit at many different offsets
    assert host_enc[0].tobytes() == oracle.encode_all(base[0].cpu().numpy().reshape(-1), channels, 44100, samples)
