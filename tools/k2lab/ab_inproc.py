#!/usr/bin/env python3
"""A/B of decode-kernel builds in ONE process on the SAME buffers (K2u's time depends on where its buffers are -- up to 10 %
between allocations, tools/k2lab/placement.py -- so builds timed in separate processes cannot be compared below that).
python tools/k2lab/ab_inproc.py lib_a.so lib_b.so ...   (paths relative to tools/k2lab/bin/)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import qoaudio_amd._lib as L  # noqa: E402
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

names = sys.argv[1:]
libs = {"product": L.lib()}
for n in names:
    L._lib = None
    L.LIB_PATH = os.path.join(ROOT, "tools", "k2lab", "bin", n)
    libs[n] = L.lib()
L._lib = libs["product"]

ch = int(os.environ.get("QOA_CH", "2"))
n_clips, frames = int(os.environ.get("QOA_CLIPS", 2048 // ch // 64 * 64)), 256
samples = frames * 5120
desc = gpu.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(64, samples, ch, device="cuda").repeat((n_clips // 64, 1, 1)).contiguous()
enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
batch.EncodePlan(clips).run(pcm, enc)
torch.cuda.synchronize()
del pcm
dplan = batch.DecodePlan.from_clips(clips)
outs = []
junk = []
for k in range(int(os.environ.get("N_OUT", "4"))):
    junk.append(torch.empty((k + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device="cuda"))
    outs.append(torch.empty((n_clips, samples, ch), dtype=torch.int16, device="cuda"))
ref = None
for o, out in enumerate(outs):
    row = []
    for rep in range(int(os.environ.get("REPS", "3"))):
        for name, lib in libs.items():
            L._lib = lib
            dplan.run(enc, out)
            torch.cuda.synchronize()
            r = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dplan.run(enc, out); e1.record(); torch.cuda.synchronize()
                r.append(e0.elapsed_time(e1))
            row.append(f"{name.replace('lib_', '').replace('.so', '')} {sum(r) / len(r):.3f}")
            if ref is None:
                ref = out.clone()
            elif rep == 0 and not torch.equal(ref, out):
                row[-1] += " (DIFFERENT PCM)"
    print(f"output {o}: " + " | ".join(row), flush=True)
