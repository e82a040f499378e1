#!/usr/bin/env python3
"""K2u on full-size batches whose PCM is off the 128-byte grid (round 5: the absolute-line variant): 1024 stereo clips x FRAMES
frames, every clip's PCM shifted by DELTA bytes from the grid (clip stride padded to whole lines, so every frame of the batch
has the same delta).  Kernel ms by HIP events, best of 5 behind a warm-up; fraction of the 8 TB/s peak; PCM against delta = 0.
    python tools/k2lab/offgrid.py [frames=256] [deltas=0,16,48,8,40,120,4] [channels=2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import qoaudio_amd._lib as L  # noqa: E402
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

# QOA_LIBS=lib_a.so,lib_b.so (relative to tools/k2lab/bin/): builds alternating in THIS process on the same buffers
libs = {"product": L.lib()}
for n in filter(None, os.environ.get("QOA_LIBS", "").split(",")):
    L._lib = None
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", n)
    libs[n.replace("lib_", "").replace(".so", "")] = L.lib()
L._lib = libs["product"]

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
deltas = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,16,48,8,40,120,4").split(",")]
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n_clips = 2048 // ch
samples = frames * 5120
desc = gpu.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda")
enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
batch.EncodePlan(clips).run(pcm, enc)
torch.cuda.synchronize()
ref = None
for delta in deltas:
    shifted = [(po + delta // 2, oo, d) for po, oo, d in clips]
    out = torch.zeros(n_clips * per_pcm + 64, dtype=torch.int16, device="cuda")
    assert out.data_ptr() % 128 == 0
    plan = batch.DecodePlan.from_clips(shifted)
    byts = plan.in_bytes + plan.pcm_bytes
    for rep in range(2 if len(libs) > 1 else 1):
        for name, lib in libs.items():
            L._lib = lib
            out.zero_()
            for _ in range(6):
                plan.run(enc, out)
            torch.cuda.synchronize()
            ms = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); plan.run(enc, out); e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            got = out[delta // 2: delta // 2 + n_clips * per_pcm]
            if ref is None:
                ref = got.clone()
            print(f"delta {delta:3d} {name:12s}: {min(ms):.3f} ms (mean {sum(ms) / 5:.3f})  {byts / (min(ms) * 1e-3) / 8e12:.3f} of the peak  "
                  f"PCM equal to delta 0: {bool(torch.equal(ref, got))}  untouched in front: {bool((out[: delta // 2] == 0).all())}", flush=True)
    L._lib = libs["product"]
