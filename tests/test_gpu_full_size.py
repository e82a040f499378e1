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
