#!/usr/bin/env python3
"""The general decode kernel (K2) alone on the short last frames of the bench's `stereo_short_last_frame*` shapes: 1024 stereo
clips of ONE 3,082-sample frame, PCM packed back to back (every clip 40 bytes further off the 128-byte grid), 16-byte aligned,
or on 128-byte lines.  Kernel ms by HIP events (best of 8).  python tools/k2lab/short_only.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

n_clips, ch, samples = 1024, 2, 3082
desc = gpu.QoaDesc(ch, 44100, samples)
per_enc = gpu.encoded_size(ch, samples)
pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda")
clips0, _, _ = batch.uniform_clips(n_clips, desc)
enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
batch.EncodePlan(clips0).run(pcm, enc)
torch.cuda.synchronize()
for name, pad in (("packed (40 bytes further off the grid per clip)", 0), ("16-byte aligned", 2), ("on 128-byte lines", 22)):
    stride = (samples + pad) * ch
    clips = [(k * stride, k * per_enc, desc) for k in range(n_clips)]
    out = torch.zeros(n_clips * stride + 64, dtype=torch.int16, device="cuda")
    plan = batch.DecodePlan.from_clips(clips)
    for _ in range(4):
        plan.run(enc, out)
    torch.cuda.synchronize()
    ms = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(enc, out); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print(f"{name:50s} {min(ms):.3f} ms (mean {sum(ms) / len(ms):.3f})", flush=True)
