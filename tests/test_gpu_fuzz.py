"""Randomised parity against the oracle: ragged mixed-channel batches through K1 (every step form) and K2, slices lane by
lane around the packed path's admission window, hand-built hostile streams through the batch decoder.
This is the test that found the idle-lane LDS overwrite in K2 (channel counts that do not divide 64) and, in a long
builder-side run, round 3's marked-candidate tie bug.

Seeding (round 5): every case draws from np.random.default_rng([FUZZ_SEED, case...]) with FUZZ_SEED = a digest of the
source tree (tests/conftest.py) -- the driver's round-end run therefore explores inputs no earlier run has seen.  The
seed is printed at the end of every run and with every failure; QOA_FUZZ_SEED=<n> reproduces.  QOA_FUZZ_SCALE (default 1)
multiplies the amount of work; the default is sized so that `pytest -m gpu` takes about five minutes on the GPU box.

Invariants: /root/reference/fuzz/fuzz_targets/encode_round_trip.rs:5-47 (encode never fails on a valid desc, decode of
the result yields the desc's sample count), strengthened to byte equality with the oracle; the search under test is
src/lib.rs:516-593."""
import os

import numpy as np
import pytest
import torch

from conftest import FUZZ_SEED, fuzz_rng, rng_pcm
from test_gpu_parity import _crafted_stream, _gpu_slice

pytestmark = pytest.mark.gpu
SCALE = float(os.environ.get("QOA_FUZZ_SCALE", "1"))
KINDS = ["white", "loud", "mix", "square", "silence"]

L_EXACT = 3875 << 18    # the packed forms' exactness window: weight energy below it (packed form)
P_EDGE = 0x900 << 18    # weights penalty becomes non-zero from here (src/lib.rs:814-816)
RANK_SAT = 32767 ** 2   # the speculative forms' saturating 32-bit rank is exact below this
DEQ_MAX = [7, 49, 147, 315, 588, 966, 1477, 2128, 2947, 3934, 5117, 6496, 8099, 9933, 12005, 14336]  # src/lib.rs:847-864


def n_trials(base):
    return max(1, int(round(base * SCALE)))


def all_forms():
    """Every step-form switch of qoa_encode_plan_create (same bytes by construction; include/qoaudio_amd.h)."""
    from qoaudio_amd import batch as B
    return [0, B.PLAN_NO_SPECULATION, B.PLAN_FORCE_32BIT_STEP, B.PLAN_EXACT_QUOTIENT, B.PLAN_NO_BLOCK32,
            B.PLAN_FORCE_32BIT_STEP | B.PLAN_NO_BLOCK32, B.PLAN_CPW_4 | B.PLAN_NO_HELPER, B.PLAN_CPW_4 | B.PLAN_NO_SPECULATION,
            B.PLAN_NO_HELPER, B.PLAN_CPW_2 | B.PLAN_NO_HELPER, B.PLAN_CPW_1, B.PLAN_CPW_2 | B.PLAN_FORCE_HELPER, B.PLAN_NO_WIDE_FORM,
            B.PLAN_NO_PENALTY_FORM, B.PLAN_LDS_TABLES, B.PLAN_FORCE_GENERIC_PATH, B.PLAN_NO_PENALTY_FORM | B.PLAN_NO_WIDE_FORM]


def g_step(sf):
    """the most one LMS update can add to the weight energy at this scalefactor (|w| <= 16 k in the window)"""
    d = DEQ_MAX[sf] >> 4
    return 8 * 16000 * d + 4 * d * d


def weights_with_energy(r, target):
    """four weights of random signs and proportions whose energy sum w^2 is within a few thousand of `target`"""
    target = max(int(target), 4)
    frac = r.dirichlet(np.ones(4) * float(r.choice([0.3, 1.0, 5.0])))
    w = np.sqrt(frac * target).astype(np.int64)
    rest = target - int((w * w).sum())
    k = int(np.argmax(w))
    w[k] = int(np.sqrt(max(int(w[k]) ** 2 + rest, 0)))
    w = np.minimum(w, 46000)
    return [int(v) * int(r.choice([-1, 1])) for v in w]


def hostile_state(r, channels, klass):
    """[channels, 8] LMS entry states of one of the classes the exactness proofs of K1's step forms are weakest on"""
    state = np.zeros((channels, 8), dtype=np.int64)
    for c in range(channels):
        if klass == "energy_edge":      # weight energy at the packed window's limit or the penalty threshold, +- a few steps' growth
            sf = int(r.integers(0, 16))
            target = int(r.choice([L_EXACT, P_EDGE])) + int(r.integers(-4 * g_step(sf), 4 * g_step(sf) + 1))
            w = weights_with_energy(r, target)
        elif klass == "i16_edge":       # one or more weights at the i16 limit (pk_try_pack's boundary), the rest anything
            edge = [32767, -32768, 32766, -32767, 32767 - 896, 31871, -31871, 31872, 32768, -32769, 33000, 32767 - int(r.integers(0, 20 * 896))]
            w = [int(r.choice(edge)) if r.random() < 0.5 else int(r.integers(-20000, 20001)) for _ in range(4)]
            if all(abs(v) < 31000 for v in w):
                w[int(r.integers(0, 4))] = int(r.choice(edge))
        elif klass == "aligned":        # all four products of one sign: the largest predictions the energy allows (sf-0/1 quotients wrap)
            amp = int(r.choice([9000, 12288, 14500, 15000, 15900, 16300]))
            w = list(np.abs(r.integers(amp - amp // 8, amp + 1, 4)) * (1 if r.random() < 0.5 else -1))
        else:                           # "random"
            amp = int(r.choice([100, 3000, 9000, 12000, 12288, 14000, 15900, 16300, 30000]))
            w = list(r.integers(-amp, amp + 1, 4))
        if klass == "aligned" or r.random() < 0.3:
            h = np.full(4, int(r.choice([-32768, 32767])))
            if klass == "aligned":      # history signs follow the weights' so that the products add up
                h = np.abs(h.astype(np.int64)).clip(max=32767) * np.sign(w) * (1 if r.random() < 0.5 else -1)
                h = np.clip(h, -32768, 32767)
        else:
            h = r.integers(-32768, 32768, 4)
        state[c] = list(h) + list(w)
    return ((state + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int32)


def hostile_signal(r, n, kind):
    if kind == "rail":
        x = np.full(n, int(r.choice([-32768, 32767])), dtype=np.int16)
        if r.random() < 0.5:            # a few samples off the rail: sample bounds of 1, 4, 9 ...
            idx = r.integers(0, n, max(1, n // 100))
            x[idx] = np.clip(x[idx].astype(np.int32) - np.sign(int(x[0])) * r.integers(0, 4, idx.size), -32768, 32767).astype(np.int16)
        return x
    if kind == "ramp":
        return np.clip(np.linspace(-40000, 40000, n) * r.choice([-1, 1]), -32768, 32767).astype(np.int16)
    if kind == "clipped":
        return np.clip(rng_pcm(int(r.integers(1 << 30)), n, "mix").astype(np.int32) * int(r.choice([3, 5, 8])), -32768, 32767).astype(np.int16)
    if kind == "white_edge":            # white noise whose winners' total ranks straddle 32767^2 (the saturating rank's exact range)
        a = int(r.integers(28500, 32768))
        return r.integers(-a, a + 1, n).astype(np.int16)
    if kind == "flip":
        return rng_pcm(int(r.integers(1 << 30)), n, "loud")
    return rng_pcm(int(r.integers(1 << 30)), n, kind)


def encode_from_state(gpu, oracle, channels, n, state, x, forms, frame=5120):
    """`x` (n sample-times x channels) through the frame-at-a-time encoder from `state`, once per step form: bytes of every
    frame and the final i32 state against the oracle's."""
    from test_gpu_parity import _oracle_with_state
    ref = _oracle_with_state(oracle, channels, 44100, n, state)
    wants = []
    for f0 in range(0, n, frame):
        piece = x[f0 * channels:min(f0 + frame, n) * channels]
        wants.append((piece, ref.encode_frame(piece)[0]))
    want_state = np.array([h + w for h, w in ref.lms_state()], dtype=np.int32)
    for fl in forms:
        enc = gpu.QoaEncoder(gpu.QoaDesc(channels, 44100, n))
        enc.debug_set_plan_flags(fl)
        enc.set_lms_state(state)
        for k, (piece, want) in enumerate(wants):
            got_len, data = enc.encode_frame(piece)
            assert got_len == len(piece) // channels and data == want, (hex(fl), k, state.tolist())
        assert np.array_equal(enc.lms_state(), want_state), (hex(fl), state.tolist())


def pick_forms(r, trial, every=6, extra=3):
    forms = all_forms()
    if trial % every == 0:
        return forms
    return [0] + [forms[int(i)] for i in r.choice(np.arange(1, len(forms)), extra, replace=False)]


@pytest.mark.parametrize("case", [1, 2])
def test_fuzz_slices_lane_by_lane(gpu, oracle, case):
    r = fuzz_rng(case)
    declined = 0
    n = n_trials(60000)
    for _ in range(n):
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
    assert 0 < declined < n


@pytest.mark.parametrize("case", [40, 41])
def test_fuzz_slices_at_the_proofs_edges(gpu, oracle, case):
    """One slice, all 16 candidates, lane by lane (every form of qoa_debug_encode_slice) from the states and signals the
    exactness arguments of K1's packed forms are weakest on: weight energy at `3875 << 18` +- a few steps' growth G(sf) and at
    the penalty threshold `0x900 << 18`; weights at the i16 limit; samples on a rail with aligned maximal predictions
    (scalefactor-0 AND -1 quotients wrap: marked candidates); winners whose total rank is about 32767^2."""
    r = fuzz_rng(case)
    near_sat = 0
    for trial in range(n_trials(40000)):
        klass = str(r.choice(["energy_edge", "i16_edge", "aligned", "random"], p=[0.4, 0.25, 0.25, 0.1]))
        st = hostile_state(r, 1, klass)[0]
        kind = str(r.choice(["rail", "white_edge", "flip", "clipped", "white", "mix", "ramp"]))
        smp = hostile_signal(r, 20, kind)
        if r.random() < 0.15:
            smp = smp[: int(r.integers(1, 20))]
        hist, w = [int(v) for v in st[:4]], [int(v) for v in st[4:]]
        s, sf, (eh, ew), totals = oracle.encode_slice(hist, w, smp)
        near_sat += 0.8 * RANK_SAT < min(totals) < 1.25 * RANK_SAT
        for generic in ((3, 0) if trial % 4 else (3, 0, 1, 2)):
            tot, aerr0, sl, es, win = _gpu_slice(hist + w, smp, generic)
            assert tot == totals and win == sf and list(es[sf]) == eh + ew, (klass, kind, generic, hist, w, list(smp))
            assert sl[sf] == (s << ((20 - len(smp)) * 3)) & (2 ** 64 - 1), (klass, kind, generic, hist, w, list(smp))
    assert near_sat > 0  # the class is being hit


@pytest.mark.parametrize("case", [3, 4, 5])
def test_fuzz_ragged_batches_round_trip(gpu, oracle, case):
    from qoaudio_amd import batch
    r = fuzz_rng(case)
    for _ in range(n_trials(40)):
        shapes = [(int(r.integers(1, 9)), int(r.integers(1, 30000))) for _ in range(int(r.integers(1, 40)))]
        clips, parts, po, oo = [], [], 0, 0
        for ch, ns in shapes:
            kind = str(r.choice(KINDS + ["white_edge", "clipped"]))
            x = hostile_signal(r, ns * ch, kind)
            parts.append(x)
            clips.append((po, oo, gpu.QoaDesc(ch, int(r.integers(1, 200000)), ns)))
            po += x.size
            oo += gpu.encoded_size(ch, ns)
        pcm = torch.from_numpy(np.concatenate(parts)).cuda()
        outs = []
        for fl in all_forms():
            out = torch.zeros(oo, dtype=torch.uint8, device="cuda")
            batch.EncodePlan(clips, fl).run(pcm, out)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        for (p_, o_, d), x in zip(clips, parts):
            want = oracle.encode_all(x, d.channels, d.sample_rate, d.samples)
            assert outs[0][o_:o_ + len(want)].tobytes() == want, shapes
        for fl, o in zip(all_forms()[1:], outs[1:]):
            assert np.array_equal(outs[0], o), (hex(fl), shapes)
        back = torch.zeros_like(pcm)
        batch.DecodePlan.from_clips(clips).run(torch.from_numpy(outs[0]).cuda(), back)
        torch.cuda.synchronize()
        hb = back.cpu().numpy()
        for (p_, o_, d), x in zip(clips, parts):
            _, _, want = oracle.decode_all(outs[0][o_:o_ + gpu.encoded_size(d.channels, d.samples)].tobytes())
            assert np.array_equal(hb[p_:p_ + len(want)], want), shapes
            assert len(want) == d.samples * d.channels  # encode_round_trip.rs:40-46


@pytest.mark.parametrize("case", [6, 7])
def test_fuzz_hostile_streams_batch_decode(gpu, oracle, case):
    from qoaudio_amd import batch
    r = fuzz_rng(case)
    for _ in range(n_trials(400)):
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
            b += len(w) + int(r.choice([0, 0, 2, 4, 6, 32]))  # PCM on and off the 16-byte / 128-byte grids
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


@pytest.mark.parametrize("case", [8, 9])
def test_fuzz_streaming_encoder_frames_of_any_length(gpu, oracle, case):
    """encode_frame accepts 1..5120 sample-times per call (src/lib.rs:424-430); the LMS state is carried in device
    memory from call to call.  Random channel counts and frame lengths, state compared after every call."""
    r = fuzz_rng(case)
    for _ in range(n_trials(800)):
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


@pytest.mark.parametrize("case", [10, 11])
def test_fuzz_decoder_iterator_on_damaged_streams(gpu, oracle, case):
    """QoaDecoder's item sequence (frame headers, samples, the error that ends it) on hand-built streams that are
    truncated at random positions and have random bytes flipped in their headers -- fixed and streaming mode."""
    from test_gpu_parity import _same_items
    r = fuzz_rng(case)
    for _ in range(n_trials(600)):
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


@pytest.mark.parametrize("case", [12, 13, 14])
def test_fuzz_block_loop_from_hostile_states(gpu, oracle, case):
    """K1's block loop (cheap-quotient / exact speculative / penalty / wide / full / 32-bit forms, chosen slice by slice on the
    device, and every QOA_PLAN_* switch that removes one of them) entered from the states the proofs are weakest on (see
    test_fuzz_slices_at_the_proofs_edges) -- two to four blocks of 16 full slices per call, 1 to 8 channels, bytes and end state
    against the oracle."""
    r = fuzz_rng(case)
    for trial in range(n_trials(8000)):
        channels = int(r.choice([1, 2, 2, 2, 3, 4, 6, 8]))
        n = 320 * int(r.integers(2, 5)) + (int(r.integers(0, 20)) if r.random() < 0.2 else 0)
        klass = str(r.choice(["energy_edge", "i16_edge", "aligned", "random"], p=[0.35, 0.2, 0.3, 0.15]))
        state = hostile_state(r, channels, klass)
        kind = str(r.choice(["rail", "flip", "white", "mix", "ramp", "white_edge", "clipped"]))
        x = hostile_signal(r, n * channels, kind)
        encode_from_state(gpu, oracle, channels, n, state, x, pick_forms(r, trial))


@pytest.mark.parametrize("case", [21, 22])
def test_marked_candidates_in_rank_ties(gpu, oracle, case):
    """Signals that sit on a rail, entered with predictions far beyond it: many candidates reach the SAME total rank (often
    0: the reconstruction clamps onto the sample whatever the quantiser says), the scalefactor-0 (and, with aligned maximal
    weights, scalefactor-1) candidate's quotient wraps (it is marked in the cheap-quotient form) and the slice's sample bound is
    0 -- the case in which a marked candidate and an unmarked one tie and only the reference's (first-sample rank, scalefactor)
    order decides.  Found by the long fuzz run (QOA_FUZZ_SCALE=8, seed 11, trial 391) against round 3's first branch-free
    marking, which let the unmarked one win."""
    r = fuzz_rng(case)
    channels, n = 2, 640
    for trial in range(n_trials(6000)):
        state = np.zeros((channels, 8), dtype=np.int32)
        rail = int(r.choice([-32768, 32767]))
        for c in range(channels):
            amp = int(r.choice([3000, 9000, 12288, 15000, 15900, 16300]))
            w = np.abs(r.integers(-amp, amp + 1, 4)) * (1 if r.random() < 0.7 else -1)
            if r.random() < 0.3:   # all four at the amplitude: predictions of +-250 k, the scalefactor-1 quotient wraps too
                w = np.full(4, amp) * (1 if r.random() < 0.5 else -1)
            h = np.full(4, rail if r.random() < 0.7 else -rail - 1) if r.random() < 0.8 else r.integers(-32768, 32768, 4)
            state[c] = list(h) + list(w)
        x = hostile_signal(r, n * channels, "rail")
        x[:] = np.where(x == x[0], rail, x) if r.random() < 0.5 else x
        encode_from_state(gpu, oracle, channels, n, state, x, pick_forms(r, trial, every=10, extra=2))


@pytest.mark.parametrize("case", [31, 32])
def test_fuzz_frames_from_hostile_states(gpu, oracle, case):
    """Like test_fuzz_block_loop_from_hostile_states but three whole frames per trial through the frame-at-a-time encoder: 48
    blocks from a hostile state, so the ladder's cool-downs, the steady-state dispatch and the hand-over of the LMS state
    between calls (device memory, full i32) all take part -- bytes of every frame and the final state against the oracle."""
    r = fuzz_rng(case)
    for trial in range(n_trials(800)):
        channels = int(r.choice([1, 2, 2, 4]))
        n = 5120 * 3 - int(r.integers(0, 2)) * int(r.integers(1, 5000))     # sometimes a short last frame
        klass = str(r.choice(["energy_edge", "i16_edge", "aligned", "random"]))
        state = hostile_state(r, channels, klass)
        kind = str(r.choice(["rail", "loud", "white", "mix", "square", "clipped", "white_edge"]))
        x = hostile_signal(r, n * channels, kind)
        if kind == "rail":
            x[r.integers(0, x.size, 40)] = r.integers(-32768, 32768, 40).astype(np.int16)
        encode_from_state(gpu, oracle, channels, n, state, x, pick_forms(r, trial, every=4, extra=2))
