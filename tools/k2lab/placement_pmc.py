#!/usr/bin/env python3
"""One decode launch per (input, output) pair after a warm-up each -- to be run under rocprofv3 --pmc: the kernel-trace order
is the order printed here, HIP-event times beside it.  python tools/k2lab/placement_pmc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

n_clips, frames, ch = 1024, 256, 2
samples = frames * 5120
desc = gpu.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(64, samples, ch, device="cuda").repeat((16, 1, 1)).contiguous()
enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
batch.EncodePlan(clips).run(pcm, enc)
torch.cuda.synchronize()
del pcm
dplan = batch.DecodePlan.from_clips(clips)
outs, junk = [], []
for k in range(6):
    junk.append(torch.empty((k + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device="cuda"))
    outs.append(torch.empty((n_clips, samples, ch), dtype=torch.int16, device="cuda"))
for o in range(len(outs)):
    for rep in range(3):   # launches 3o .. 3o+2 of the decode kernel: the last one is the measured one
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dplan.run(enc, outs[o]); e1.record(); torch.cuda.synchronize()
    print(f"PAIR out{o}: {e0.elapsed_time(e1):.3f} ms", flush=True)
