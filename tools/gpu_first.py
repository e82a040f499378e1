"""First contact with the GPU: parity spot checks + rough timings (development aid, not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import qoaudio_amd as q
from qoaudio_amd import batch, synth
from oracle import oracle as O

fx = open("tests/golden/julien_baker_sprained_ankle.qoa", "rb").read()
ch, sr, pcm = O.decode_all(fx)
t = time.time(); d = q.decode_all(fx); print("gpu decode_all", time.time() - t, d.num_channels, d.sample_rate, d.samples.size)
print("decode parity:", np.array_equal(d.samples, pcm))
t = time.time(); enc = q.encode_all(pcm, q.QoaDesc(ch, sr, pcm.size // ch)); print("gpu encode_all", time.time() - t, len(enc))
ref = O.encode_all(pcm, ch, sr, pcm.size // ch)
print("encode parity:", enc == ref, "fixture prefix:", enc[:1073768] == fx[:1073768])
if enc != ref:
    a = np.frombuffer(enc, np.uint8); b = np.frombuffer(ref, np.uint8)
    print("first diff at", np.nonzero(a != b)[0][:10])

for nch in (1, 2, 3, 8):
    for n in (1, 19, 20, 21, 45, 5119, 5120, 5121, 12345):
        x = np.random.default_rng(n * 10 + nch).integers(-20000, 20000, n * nch).astype(np.int16)
        a = q.encode_all(x, q.QoaDesc(nch, 48000, n)); b = O.encode_all(x, nch, 48000, n)
        if a != b: print("MISMATCH enc", nch, n)
        dd = q.decode_all(a); _, _, oo = O.decode_all(a)
        if not np.array_equal(dd.samples, oo): print("MISMATCH dec", nch, n)
print("small cases done")

dev = torch.device("cuda:0")
for (n_clips, frames, nch, flags) in [(1024, 8, 2, 0), (1024, 8, 1, 0), (4096, 4, 2, 0), (4096, 4, 2, batch.PLAN_CPW_1), (4096, 4, 2, batch.PLAN_CPW_4), (1024, 8, 2, batch.PLAN_CPW_2), (1024, 8, 2, batch.PLAN_CPW_4)]:
    samples = frames * 5120
    x = synth.make_clips(16, samples, nch, device=dev)
    x = x.repeat((n_clips // 16, 1, 1)).contiguous()
    desc = q.QoaDesc(nch, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    plan = batch.EncodePlan(clips, flags)
    out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
    plan.run(x, out); torch.cuda.synchronize()
    t = time.time(); plan.run(x, out); torch.cuda.synchronize(); dt = time.time() - t
    print(f"encode clips={n_clips} frames={frames} ch={nch} flags={flags}: {dt*1e3:.2f} ms  {plan.samples/dt/1e9:.2f} Gsamples/s")
    hb = out[:16].cpu().numpy()
    xs = x[:16].cpu().numpy()
    ok = all(hb[i].tobytes() == O.encode_all(xs[i].reshape(-1), nch, 44100, samples, O.EXHAUSTIVE) for i in range(2))
    print("   parity(2 clips):", ok)
    dplan = batch.DecodePlan.from_clips(clips)
    pcm2 = torch.empty_like(x)
    dplan.run(out, pcm2); torch.cuda.synchronize()
    t = time.time(); dplan.run(out, pcm2); torch.cuda.synchronize(); dt = time.time() - t
    print(f"   decode: {dt*1e3:.2f} ms {dplan.samples/dt/1e9:.2f} Gsamples/s  {(dplan.in_bytes+dplan.pcm_bytes)/dt/1e9:.1f} GB/s mism={dplan.mismatches()}")
    _, _, o0 = O.decode_all(hb[0].tobytes())
    print("   decode parity:", np.array_equal(pcm2[0].cpu().numpy().reshape(-1), o0))
