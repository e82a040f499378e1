#!/usr/bin/env python3
"""K2u on channel counts that do not divide a wavefront: where do decoded samples differ from the oracle's?  (debug aid)
Run on the GPU box: python tools/k2lab/check_general.py [channels [clips [frames]]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402
from oracle import oracle as O  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
samples = frames * 5120
desc = gpu.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda")
enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
batch.EncodePlan(clips).run(pcm, enc)
back = torch.full_like(pcm, 12345)
batch.DecodePlan.from_clips(clips).run(enc, back)
torch.cuda.synchronize()
he, hb = enc.cpu().numpy(), back.cpu().numpy()
bad_total = 0
for k in range(n_clips):
    want = O.decode_all(he[k].tobytes())[2].reshape(samples, ch)
    got = hb[k]
    bad = np.argwhere(want != got)
    bad_total += len(bad)
    if len(bad) and bad_total == len(bad):
        t, c = bad[:, 0], bad[:, 1]
        print(f"clip {k}: {len(bad)} wrong of {want.size}; first (t, c) {bad[:8].tolist()}")
        print("  frames hit:", sorted(set((t // 5120).tolist()))[:20], " channels hit:", sorted(set(c.tolist())))
        print("  rows hit (t % 5120 // 20):", sorted(set((t % 5120 // 20).tolist()))[:40])
        print("  sample-in-row hit:", sorted(set((t % 20).tolist())))
        print("  untouched (12345):", int((got[t, c] == 12345).sum()))
        byte = (t % 5120) * ch * 2 + c * 2
        print("  byte offset in frame % 128 hit:", sorted(set((byte % 128).tolist()))[:64])
        print("  line in frame hit:", sorted(set((byte // 128).tolist()))[:40])
print("wrong samples in total:", bad_total)
