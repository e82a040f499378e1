#!/usr/bin/env python3
"""Where a stereo batch whose clips end in a short frame (like the reference's fixture: 3,082 sample-times) spends its decode
time: full frames only (K2u), short frames only (the general kernel), both in one plan.  python tools/k2lab/short_frames.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

n_clips, ch = int(os.environ.get("QOA_CLIPS", "1024")), 2
only = os.environ.get("QOA_ONLY")
for idx, (name, samples) in enumerate((("63 full frames", 63 * 5120), ("one 3082-sample frame", 3082), ("one 3080-sample frame (whole 16-byte pieces)", 3080),
                      ("63 full frames + 3082", 63 * 5120 + 3082), ("255 full frames + 3082", 255 * 5120 + 3082))):
    if only is not None and str(idx) not in only.split(","):
        continue
    desc = gpu.QoaDesc(ch, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    pad = (-samples) % 4 if os.environ.get("QOA_PACKED") != "1" else 0   # stereo: 4 sample-times = 16 bytes
    clips = [(k * (samples + pad) * ch, oo, dd) for k, (po, oo, dd) in enumerate(clips)]
    pcm = synth.make_clips(n_clips, samples + pad, ch, 44100, device="cuda") if samples < 70 * 5120 else \
        synth.make_clips(32, samples + pad, ch, 44100, device="cuda").repeat((n_clips // 32, 1, 1)).contiguous()
    enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    batch.EncodePlan(clips).run(pcm, enc)
    dplan = batch.DecodePlan.from_clips(clips)
    back = torch.empty_like(pcm)
    for _ in range(8):
        dplan.run(enc, back)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dplan.run(enc, back)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = enc.numel() + back.numel() * 2
    print(f"{name:45s} {ms:7.3f} ms  {nbytes / ms / 1e9:5.2f} TB/s  frac {nbytes / ms / 1e9 / 8:.3f}  equal {bool(torch.equal(back, pcm)) if False else ''}", flush=True)
    del pcm, enc, back, dplan
