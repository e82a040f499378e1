#!/usr/bin/env python3
"""K1's cycles per slice on the headline geometry (1024 stereo clips: 512 compute wavefronts + helpers) for ONE build of the
library (QOAUDIO_AMD_LIB=...): kernel ms by HIP events and -- finer -- the per-wavefront cycle counts of the recording
instantiation (EncodePlan.wave_stats(): p50 / min / max over the 512 wavefronts).  tools/k1lab/run_slack.sh alternates builds.
    python tools/k1lab/slack.py [frames=64] [kind=music]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind = sys.argv[2] if len(sys.argv) > 2 else "music"
n_clips, ch = 1024, 2
samples = frames * 5120
desc = gpu.QoaDesc(ch, 44100, samples)
clips, _, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda", kind=kind)
out = torch.zeros((n_clips, per_enc), dtype=torch.uint8, device="cuda")
plan = batch.EncodePlan(clips)
for _ in range(2):
    plan.run(pcm, out)
torch.cuda.synchronize()
ms = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(pcm, out); e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
wplan = batch.EncodePlan(clips, batch.PLAN_WAVE_STATS)
cps = []
for _ in range(3):
    wplan.run(pcm, out)
    cps.append(wplan.wave_stats()["cycles_per_slice"])
best = min(ms)
print(f"{os.path.basename(os.environ.get('QOAUDIO_AMD_LIB', 'product')):24s} {kind} {best:8.3f} ms (min of 5; mean {sum(ms) / 5:.3f})  "
      f"{best * 1e6 / (frames * 256):7.1f} ns/slice   cycles/slice p50 {min(c['p50'] for c in cps):7.1f}  min {min(c['min'] for c in cps):7.1f}  "
      f"max {min(c['max'] for c in cps):7.1f}", flush=True)
