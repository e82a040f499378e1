#!/usr/bin/env python3
"""Decode throughput per channel count (which kernel carries it, time, fraction of the HBM peak): python tools/k2lab/channels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

frames = 64
for ch in [int(c) for c in os.environ.get('QOA_CHANNELS', '1,2,3,4,5,6,7,8').split(',')]:
    n_clips = 4096 // ch
    samples = frames * 5120
    pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda")
    per, enc = samples * ch, gpu.encoded_size(ch, samples)
    clips = [(i * per, i * enc, gpu.QoaDesc(ch, 44100, samples)) for i in range(n_clips)]
    eplan = batch.EncodePlan(clips)
    out = torch.zeros(eplan.out_bytes, dtype=torch.uint8, device="cuda")
    eplan.run(pcm.view(-1), out)
    dplan = batch.DecodePlan.from_clips(clips)
    buf = torch.zeros(pcm.numel() + 8, dtype=torch.int16, device="cuda")
    import hashlib
    # the same plan into a 16-byte aligned buffer (K2u: uniform full frames) and into one 8 bytes off (the general kernel K2)
    for name, back in (("K2u", buf[:pcm.numel()]), ("K2u, PCM base 8 bytes off the 16-byte grid (write-back stores)", buf[4:4 + pcm.numel()]),
                       ("K2 (PCM base 2 bytes off the dword grid)", buf[1:1 + pcm.numel()])):
        for _ in range(6):
            dplan.run(out, back)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dplan.run(out, back)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = out.numel() + back.numel() * 2
        digest = hashlib.sha256(back.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{ch} ch: {n_clips} clips x {frames} frames  decode {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s  frac {nbytes / ms / 1e9 / 8:.3f}  "
              f"pcm sha {digest}  {name}", flush=True)
