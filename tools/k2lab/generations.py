#!/usr/bin/env python3
"""K2u and the wave-quantisation tail: a wavefront decodes its frames start to end (~0.4 ms), so a launch lasts
ceil(wavefronts / resident slots) wave lifetimes.  Decode time and fraction of the HBM peak for batch sizes around whole
multiples of the chip's slots, optionally with fewer workgroups per CU (QOA_LAB_K2U_DYN_LDS on a -DQOA_K2U_LAB_DYNLDS build):
    python tools/k2lab/generations.py [channels] [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fpw = 64 // ch
samples = frames * 5120
sizes = [int(v) for v in os.environ.get("QOA_WAVES", "2048,3072,4096,6144,8192,9216,12288").split(",")]
base = synth.make_clips(32, samples, ch, 44100, device="cuda")
for waves in sizes:
    n_clips = waves * fpw // frames
    desc = gpu.QoaDesc(ch, 44100, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    pcm = base.repeat(((n_clips + 31) // 32, 1, 1))[:n_clips].contiguous()
    enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    batch.EncodePlan(clips).run(pcm, enc)
    torch.cuda.synchronize()
    del pcm
    out = torch.empty((n_clips, samples, ch), dtype=torch.int16, device="cuda")
    dplan = batch.DecodePlan.from_clips(clips)
    row = []
    for lds in os.environ.get("QOA_LDS_LIST", "0").split(","):
        os.environ["QOA_LAB_K2U_DYN_LDS"] = lds
        for _ in range(3):
            dplan.run(enc, out)
        torch.cuda.synchronize()
        r = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dplan.run(enc, out); e1.record(); torch.cuda.synchronize()
            r.append(e0.elapsed_time(e1))
        ms = sorted(r)[len(r) // 2]
        nbytes = enc.numel() + out.numel() * 2
        row.append(f"dyn LDS {lds:>6s}: {ms:.3f} ms frac {nbytes / ms / 1e9 / 8:.3f}")
    print(f"{ch} ch, {n_clips} clips x {frames} frames = {waves} wavefronts: " + " | ".join(row), flush=True)
    del enc, out
