"""Randomised parity against the oracle: ragged mixed-channel batches through K1 (both step forms) and K2, slices
lane by lane around the packed path's admission window, hand-built hostile streams through the batch decoder.
This is the test that found the idle-lane LDS overwrite in K2 (channel counts that do not divide 64).
QOA_FUZZ_SCALE (default 1) multiplies the amount of work for a long run on the GPU box."""
import os

import numpy as np
import pytest
import torch

from conftest import rng_pcm
from test_gpu_parity import _crafted_stream, _gpu_slice

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("QOA_FUZZ_SCALE", "1"))
KINDS = ["white", "loud", "mix", "square", "silence"]


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_slices_lane_by_lane(gpu, oracle, seed):
    r = np.random.default_rng(seed)
    declined = 0
    for _ in range(400 * SCALE):
        amp = int(r.choice([100, 2000, 8000, 12000, 12290, 14000, 15800, 15950, 16100, 20000, 31000, 32767, 40000]))
        w = [int(v) for v in r.integers(-amp, amp + 1, 4)]
        if r.random() < 0.3:
            w = [int(np.sign(v) or 1) * amp for v in w]
        hist = [int(v) for v in r.integers(-32768, 32768, 4)]
        if r.random() < 0.2:
            hist = [int(r.choice([-32768, 32767, 0, -1])) for _ in range(4)]
        smp = rng_pcm(int(r.integers(1 << 30)), 20, str(r.choice(KINDS)))
        if r.random() < 0.3:
            smp = (smp.astype(np.int32) // int(r.choice([1, 7, 100, 3000]))).astype(np.int16)
        s, sf, (eh, ew), totals = oracle.encode_slice(hist, w, smp)
        tot, aerr0, sl, es, win = _gpu_slice(hist + w, smp, 3)
        assert tot == totals and win == sf and sl[sf] == s and list(es[sf]) == eh + ew, (hist, w, list(smp))
        declined += _gpu_slice.declined
    assert 0 < declined < 400 * SCALE


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_fuzz_ragged_batches_round_trip(gpu, oracle, seed):
    from qoaudio_amd import batch
    r = np.random.default_rng(seed)
    for _ in range(2 * SCALE):
        shapes = [(int(r.integers(1, 9)), int(r.integers(1, 30000))) for _ in range(int(r.integers(1, 40)))]
        clips, parts, po, oo = [], [], 0, 0
        for ch, ns in shapes:
            x = rng_pcm(int(r.integers(1 << 30)), ns * ch, str(r.choice(KINDS)))
            parts.append(x)
            clips.append((po, oo, gpu.QoaDesc(ch, int(r.integers(1, 200000)), ns)))
            po += x.size
            oo += gpu.encoded_size(ch, ns)
        pcm = torch.from_numpy(np.concatenate(parts)).cuda()
        outs = []
        for fl in (0, batch.PLAN_NO_SPECULATION, batch.PLAN_FORCE_32BIT_STEP, batch.PLAN_EXACT_QUOTIENT, batch.PLAN_NO_BLOCK32,
                   batch.PLAN_FORCE_32BIT_STEP | batch.PLAN_NO_BLOCK32, batch.PLAN_CPW_4, batch.PLAN_CPW_4 | batch.PLAN_NO_HELPER,
                   batch.PLAN_CPW_4 | batch.PLAN_NO_SPECULATION, batch.PLAN_NO_HELPER, batch.PLAN_CPW_2 | batch.PLAN_NO_HELPER,
                   batch.PLAN_NO_WIDE_FORM):
            out = torch.zeros(oo, dtype=torch.uint8, device="cuda")
            batch.EncodePlan(clips, fl).run(pcm, out)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
        for (p_, o_, d), x in zip(clips, parts):
            want = oracle.encode_all(x, d.channels, d.sample_rate, d.samples)
            assert outs[0][o_:o_ + len(want)].tobytes() == want, shapes
        back = torch.zeros_like(pcm)
        batch.DecodePlan.from_clips(clips).run(torch.from_numpy(outs[0]).cuda(), back)
        torch.cuda.synchronize()
        hb = back.cpu().numpy()
        for (p_, o_, d), x in zip(clips, parts):
            _, _, want = oracle.decode_all(outs[0][o_:o_ + gpu.encoded_size(d.channels, d.samples)].tobytes())
            assert np.array_equal(hb[p_:p_ + len(want)], want), shapes


@pytest.mark.parametrize("seed", [6, 7])
def test_fuzz_hostile_streams_batch_decode(gpu, oracle, seed):
    from qoaudio_amd import batch
    r = np.random.default_rng(seed)
    for _ in range(4 * SCALE):
        streams = []
        for _ in range(int(r.integers(1, 12))):
            k = int(r.integers(1, 4))
            streams.append(_crafted_stream(r, [(int(r.integers(1, 9)), 8000, 5120, 256)] * k, samples_field=5120 * k))
        wants = [oracle.decode_all(st)[2] for st in streams]
        offs_in, offs_out, a, b = [], [], 0, 0
        for st, w in zip(streams, wants):
            offs_in.append(a)
            offs_out.append(b)
            a += (len(st) + 7) // 8 * 8
            b += len(w)
        plan = batch.DecodePlan.from_streams(streams, offs_in, offs_out)
        buf = np.zeros(a, np.uint8)
        for st, o in zip(streams, offs_in):
            buf[o:o + len(st)] = np.frombuffer(st, np.uint8)
        out = torch.zeros(b, dtype=torch.int16, device="cuda")
        plan.run(torch.from_numpy(buf).cuda(), out)
        torch.cuda.synchronize()
        ho = out.cpu().numpy()
        for w, o in zip(wants, offs_out):
            assert np.array_equal(ho[o:o + len(w)], w)
        assert np.array_equal(gpu.decode_all(streams[0]).samples, wants[0])


@pytest.mark.parametrize("seed", [8, 9])
def test_fuzz_streaming_encoder_frames_of_any_length(gpu, oracle, seed):
    """encode_frame accepts 1..5120 sample-times per call (src/lib.rs:424-430); the LMS state is carried in device
    memory from call to call.  Random channel counts and frame lengths, state compared after every call."""
    r = np.random.default_rng(seed)
    for _ in range(3 * SCALE):
        ch = int(r.integers(1, 9))
        lens = [int(r.choice([1, 19, 20, 21, 319, 320, 321, 5119, 5120, int(r.integers(1, 5121))])) for _ in range(int(r.integers(2, 7)))]
        total = sum(lens)
        enc = gpu.QoaEncoder(gpu.QoaDesc(ch, 48000, total))
        ref = oracle.OracleEncoder(ch, 48000, total)
        kind = str(r.choice(KINDS))
        for n in lens:
            x = rng_pcm(int(r.integers(1 << 30)), n * ch, kind)
            fl, data = enc.encode_frame(x)
            want, wfl = ref.encode_frame(x)
            assert (fl, data) == (wfl, want), (ch, lens, n)
            want_state = np.array([h + w for h, w in ref.lms_state()], dtype=np.int32)
            assert np.array_equal(enc.lms_state(), want_state)


@pytest.mark.parametrize("seed", [10, 11])
def test_fuzz_decoder_iterator_on_damaged_streams(gpu, oracle, seed):
    """QoaDecoder's item sequence (frame headers, samples, the error that ends it) on hand-built streams that are
    truncated at random positions and have random bytes flipped in their headers -- fixed and streaming mode."""
    from test_gpu_parity import _same_items
    r = np.random.default_rng(seed)
    for _ in range(12 * SCALE):
        frames = []
        for _ in range(int(r.integers(1, 4))):
            ch = int(r.integers(1, 11))
            nspc = int(r.choice([1, 20, 41, 5120, int(r.integers(1, 5121))]))
            rows = (nspc + 19) // 20 if r.random() < 0.8 else int(r.integers(1, 257))
            frames.append((ch, int(r.integers(1, 1 << 24)), nspc, rows))
        fixed = r.random() < 0.6
        if fixed:  # a fixed-mode file: every frame must repeat channels and rate (src/lib.rs:246-262)
            frames = [(frames[0][0], frames[0][1], f[2], f[3]) for f in frames]
        data = bytearray(_crafted_stream(r, frames, samples_field=sum(f[2] for f in frames) if fixed else 0))
        if r.random() < 0.5:
            del data[int(r.integers(0, len(data))):]
        for _ in range(int(r.integers(0, 3))):
            if len(data) > 16:
                data[int(r.integers(0, 16))] ^= 1 << int(r.integers(0, 8))
        _same_items(gpu, oracle, bytes(data))


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_block_loop_from_hostile_states(gpu, oracle, seed):
    """K1's block loop (cheap-quotient / exact speculative / full forms, chosen slice by slice on the device) entered from
    random LMS states whose weights fit i16 -- up to the edge of the window the speculative forms accept, where predictions
    reach +-250 k and the scalefactor-0 AND scalefactor-1 candidates' quotients wrap to the opposite sign -- on signals that
    sit on the rails, alternate between them, or are ordinary: two blocks of 16 full slices per call, bytes and end state
    against the oracle."""
    from test_gpu_parity import _oracle_with_state
    r = np.random.default_rng(seed)
    channels, n = 2, 640
    for trial in range(60 * SCALE):
        amp = int(r.choice([3000, 9000, 12000, 12288, 14000, 15900, 16300]))
        state = np.zeros((channels, 8), dtype=np.int32)
        for c in range(channels):
            w = r.integers(-amp, amp + 1, 4)
            if r.random() < 0.5:   # all four products the same sign: the largest predictions the energy allows
                w = np.abs(w) * np.sign(r.integers(0, 2) * 2 - 1)
            h = r.integers(-32768, 32768, 4)
            if r.random() < 0.5:
                h = np.full(4, int(r.choice([-32768, 32767])))
            state[c] = list(h) + list(w)
        kind = str(r.choice(["rail", "flip", "white", "mix", "ramp"]))
        if kind == "rail":
            x = np.full(n * channels, int(r.choice([-32768, 32767])), dtype=np.int16)
        elif kind == "flip":
            x = rng_pcm(int(r.integers(1 << 30)), n * channels, "loud")
        elif kind == "ramp":
            x = np.clip(np.linspace(-40000, 40000, n * channels) * r.choice([-1, 1]), -32768, 32767).astype(np.int16)
        else:
            x = rng_pcm(int(r.integers(1 << 30)), n * channels, kind)
        enc = gpu.QoaEncoder(gpu.QoaDesc(channels, 44100, n))
        enc.set_lms_state(state)
        fl, data = enc.encode_frame(x)
        ref = _oracle_with_state(oracle, channels, 44100, n, state)
        want, _ = ref.encode_frame(x)
        assert fl == n and data == want, (trial, kind, state.tolist())
        assert np.array_equal(enc.lms_state(), np.array([h + w for h, w in ref.lms_state()], dtype=np.int32))


@pytest.mark.parametrize("seed", [21, 22])
def test_marked_candidates_in_rank_ties(gpu, oracle, seed):
    """Signals that sit on a rail, entered with predictions far beyond it: many candidates reach the SAME total rank (often
    0: the reconstruction clamps onto the sample whatever the quantiser says), the scalefactor-0 candidate's quotient wraps
    (it is marked in the cheap-quotient form) and the slice's sample bound is 0 -- the case in which a marked candidate and an
    unmarked one tie and only the reference's (first-sample rank, scalefactor) order decides.  Found by the long fuzz run
    (QOA_FUZZ_SCALE=8, seed 11, trial 391) against round 3's first branch-free marking, which let the unmarked one win."""
    from test_gpu_parity import _oracle_with_state
    r = np.random.default_rng(seed)
    channels, n = 2, 640
    for trial in range(150 * SCALE):
        state = np.zeros((channels, 8), dtype=np.int32)
        rail = int(r.choice([-32768, 32767]))
        for c in range(channels):
            amp = int(r.choice([3000, 9000, 12288, 16300]))
            w = np.abs(r.integers(-amp, amp + 1, 4)) * (1 if r.random() < 0.7 else -1)
            h = np.full(4, rail if r.random() < 0.7 else -rail - 1) if r.random() < 0.8 else r.integers(-32768, 32768, 4)
            state[c] = list(h) + list(w)
        x = np.full(n * channels, rail, dtype=np.int16)
        if r.random() < 0.5:   # a few samples off the rail: bounds of 1, 4, 9 ...
            idx = r.integers(0, x.size, 12)
            x[idx] = np.clip(x[idx].astype(np.int32) - np.sign(rail) * r.integers(0, 4, 12), -32768, 32767).astype(np.int16)
        enc = gpu.QoaEncoder(gpu.QoaDesc(channels, 44100, n))
        enc.set_lms_state(state)
        fl, data = enc.encode_frame(x)
        ref = _oracle_with_state(oracle, channels, 44100, n, state)
        want, _ = ref.encode_frame(x)
        assert fl == n and data == want, (trial, state.tolist())
        assert np.array_equal(enc.lms_state(), np.array([h + w for h, w in ref.lms_state()], dtype=np.int32))


@pytest.mark.parametrize("seed", [31, 32])
def test_fuzz_frames_from_hostile_states(gpu, oracle, seed):
    """Like test_fuzz_block_loop_from_hostile_states but three whole frames per trial through the frame-at-a-time encoder: 48
    blocks from a hostile state, so the ladder's cool-downs, the steady-state dispatch and the hand-over of the LMS state
    between calls (device memory, full i32) all take part -- bytes of every frame and the final state against the oracle."""
    from test_gpu_parity import _oracle_with_state
    r = np.random.default_rng(seed)
    channels = 2
    for trial in range(6 * SCALE):
        n = 5120 * 3 - int(r.integers(0, 2)) * int(r.integers(1, 5000))     # sometimes a short last frame
        amp = int(r.choice([3000, 12288, 16300, 30000]))
        state = np.zeros((channels, 8), dtype=np.int32)
        for c in range(channels):
            w = r.integers(-amp, amp + 1, 4)
            if r.random() < 0.5:
                w = np.abs(w) * np.sign(r.integers(0, 2) * 2 - 1)
            h = r.integers(-32768, 32768, 4)
            state[c] = list(h) + list(w)
        kind = str(r.choice(["rail", "loud", "white", "mix", "square", "clipped"]))
        if kind == "rail":
            x = np.full(n * channels, int(r.choice([-32768, 32767])), dtype=np.int16)
            x[r.integers(0, x.size, 40)] = r.integers(-32768, 32768, 40).astype(np.int16)
        elif kind == "clipped":
            x = np.clip(rng_pcm(int(r.integers(1 << 30)), n * channels, "mix").astype(np.int32) * 5, -32768, 32767).astype(np.int16)
        else:
            x = rng_pcm(int(r.integers(1 << 30)), n * channels, kind)
        enc = gpu.QoaEncoder(gpu.QoaDesc(channels, 44100, n))
        enc.set_lms_state(state)
        ref = _oracle_with_state(oracle, channels, 44100, n, state)
        for f0 in range(0, n, 5120):
            piece = x[f0 * channels:min(f0 + 5120, n) * channels]
            fl, data = enc.encode_frame(piece)
            want, _ = ref.encode_frame(piece)
            assert fl == len(piece) // channels and data == want, (trial, kind, f0, state.tolist())
        assert np.array_equal(enc.lms_state(), np.array([h + w for h, w in ref.lms_state()], dtype=np.int32))
