import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import rng_pcm
from oracle import oracle
import qoaudio_amd as gpu
from qoaudio_amd import batch
def run(shapes, label):
    clips, parts, po, oo = [], [], 0, 0
    for i, (ch, ns) in enumerate(shapes):
        x = rng_pcm(100 + i, ns * ch, "mix")
        parts.append(x); clips.append((po, oo, gpu.QoaDesc(ch, 44100, ns)))
        po += x.size; oo += gpu.encoded_size(ch, ns)
    enc = np.concatenate([np.frombuffer(oracle.encode_all(x, d.channels, d.sample_rate, d.samples), np.uint8) for (p_, o_, d), x in zip(clips, parts)])
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.zeros(po, dtype=torch.int16, device="cuda"); dplan.run(torch.from_numpy(enc).cuda(), back); torch.cuda.synchronize()
    hb = back.cpu().numpy()
    for ci, ((p_, o_, d), x) in enumerate(zip(clips, parts)):
        _, _, want = oracle.decode_all(enc[o_:o_ + gpu.encoded_size(d.channels, d.samples)].tobytes())
        got = hb[p_:p_ + len(want)]
        idx = np.nonzero(got != want)[0]
        chans = sorted(set(int(i) % d.channels for i in idx))
        frames = sorted(set(int(i) // d.channels // 5120 for i in idx))
        print(label, "clip", ci, d.channels, d.samples, "bad", len(idx), "channels", chans, "frames", frames)
run([(7, 17152)], "A")
run([(7, 352), (7, 24961), (7, 17152)], "B")
run([(7, 5120 * 9 + 100)], "C")        # 10 frames of one clip: wave 1 holds the short last frame alone
run([(7, 5120 * 10)], "D")
run([(3, 5120 * 21 + 33)], "E")        # FPW = 21
run([(5, 5120 * 12 + 1000)], "F")      # FPW = 12
run([(6, 5120 * 10 + 1000)], "G")      # FPW = 10
run([(2, 5120 * 32 + 100)], "H")
run([(1, 5120 * 64 + 100)], "I")
run([(8, 5120 * 8 + 100)], "J")
run([(4, 5120 * 16 + 100)], "K")
