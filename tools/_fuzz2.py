import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import rng_pcm
from oracle import oracle
import qoaudio_amd as gpu
from qoaudio_amd import batch
r = np.random.default_rng(5)
for it in range(40):
    shapes = [(int(r.integers(1, 9)), int(r.integers(1, 30000))) for _ in range(int(r.integers(1, 40)))]
    clips, parts, po, oo = [], [], 0, 0
    for ch, ns in shapes:
        kind = str(r.choice(["white", "loud", "mix", "square", "silence"]))
        x = rng_pcm(int(r.integers(1 << 30)), ns * ch, kind)
        parts.append(x); clips.append((po, oo, gpu.QoaDesc(ch, 44100, ns)))
        po += x.size; oo += gpu.encoded_size(ch, ns)
    enc = np.concatenate([np.frombuffer(oracle.encode_all(x, d.channels, d.sample_rate, d.samples), np.uint8) for (p_, o_, d), x in zip(clips, parts)])
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.zeros(po, dtype=torch.int16, device="cuda"); dplan.run(torch.from_numpy(enc).cuda(), back); torch.cuda.synchronize()
    hb = back.cpu().numpy()
    bad = 0
    for ci, ((p_, o_, d), x) in enumerate(zip(clips, parts)):
        _, _, want = oracle.decode_all(enc[o_:o_ + gpu.encoded_size(d.channels, d.samples)].tobytes())
        got = hb[p_:p_ + len(want)]
        if not np.array_equal(got, want):
            idx = np.nonzero(got != want)[0]
            print("iter", it, "clip", ci, "ch", d.channels, "samples", d.samples, "n_bad", len(idx), "first", idx[:6], "sample-time", idx[0] // d.channels, "got", got[idx[:4]], "want", want[idx[:4]], "pcm_off", p_)
            bad += 1
    if bad:
        print("shapes", shapes)
        break
print("done")
