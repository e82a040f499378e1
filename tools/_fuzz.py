import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from test_gpu_parity import _gpu_slice, _crafted_stream
from conftest import rng_pcm
from oracle import oracle
import qoaudio_amd as gpu
from qoaudio_amd import batch

r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0 = time.time(); n = dec = 0
# 1. slices lane by lane, packed path with fall-back, weights all over the admitted window and around its edges
while time.time() - t0 < 60:
    amp = int(r.choice([100, 2000, 8000, 12000, 12290, 14000, 15800, 15950, 16100, 20000, 31000, 32767, 40000]))
    w = [int(v) for v in r.integers(-amp, amp + 1, 4)]
    if r.random() < 0.3: w = [int(np.sign(v) or 1) * amp for v in w]
    hist = [int(v) for v in r.integers(-32768, 32768, 4)]
    if r.random() < 0.2: hist = [int(r.choice([-32768, 32767, 0, -1])) for _ in range(4)]
    kind = str(r.choice(["white", "loud", "mix", "square", "silence"]))
    smp = rng_pcm(int(r.integers(1 << 30)), 20, kind)
    if r.random() < 0.3: smp = (smp.astype(np.int32) // int(r.choice([1, 7, 100, 3000]))).astype(np.int16)
    s, sf, (eh, ew), totals = oracle.encode_slice(hist, w, smp)
    tot, aerr0, sl, es, win = _gpu_slice(hist + w, smp, 3)
    assert tot == totals and win == sf and sl[sf] == s and list(es[sf]) == eh + ew, (hist, w, list(smp))
    n += 1; dec += _gpu_slice.declined
print("slices", n, "declined", dec)
# 2. ragged batches of random signals through the production kernel, both step forms
for it in range(12):
    shapes = [(int(r.integers(1, 9)), int(r.integers(1, 30000))) for _ in range(int(r.integers(1, 40)))]
    clips, parts, po, oo = [], [], 0, 0
    for ch, ns in shapes:
        kind = str(r.choice(["white", "loud", "mix", "square", "silence"]))
        x = rng_pcm(int(r.integers(1 << 30)), ns * ch, kind)
        parts.append(x); clips.append((po, oo, gpu.QoaDesc(ch, int(r.integers(1, 200000)), ns)))
        po += x.size; oo += gpu.encoded_size(ch, ns)
    pcm = torch.from_numpy(np.concatenate(parts)).cuda()
    outs = []
    for fl in (0, batch.PLAN_FORCE_32BIT_STEP):
        out = torch.zeros(oo, dtype=torch.uint8, device="cuda")
        plan = batch.EncodePlan(clips, fl); plan.run(pcm, out); torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    for (p_, o_, d), x in zip(clips, parts):
        want = oracle.encode_all(x, d.channels, d.sample_rate, d.samples)
        assert outs[0][o_:o_ + len(want)].tobytes() == want
    # decode the batch back and compare with the oracle's decoder
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.zeros_like(pcm); dplan.run(torch.from_numpy(outs[0]).cuda(), back); torch.cuda.synchronize()
    hb = back.cpu().numpy()
    for (p_, o_, d), x in zip(clips, parts):
        _, _, want = oracle.decode_all(outs[0][o_:o_ + gpu.encoded_size(d.channels, d.samples)].tobytes())
        assert np.array_equal(hb[p_:p_ + len(want)], want)
print("batches ok")
# 3. random garbage streams through the batch decoder (hostile weights, every sf)
for it in range(30):
    streams = [_crafted_stream(r, [(int(r.integers(1, 9)), 8000, 5120, 256)] * int(r.integers(1, 4)), samples_field=0) for _ in range(int(r.integers(1, 12)))]
    for st in streams:
        d = gpu.decode_all(st); _, _, want = oracle.decode_all(st)
        assert np.array_equal(d.samples, want)
print("garbage streams ok")
