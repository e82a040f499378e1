#!/usr/bin/env python3
"""K2u vs the RELATIVE placement of its input and output: both carved out of one big allocation, the output's offset swept.
python tools/k2lab/placement2.py [step_MB=256] [count=24]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

step = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else 256 << 20
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n_clips, frames, ch = 1024, 256, 2
samples = frames * 5120
desc = gpu.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(64, samples, ch, device="cuda").repeat((16, 1, 1)).contiguous()
enc0 = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
batch.EncodePlan(clips).run(pcm, enc0)
torch.cuda.synchronize()
del pcm
torch.cuda.empty_cache()
out_bytes = n_clips * samples * ch * 2
in_bytes = enc0.numel()
big = torch.empty(in_bytes + (2 << 20) + out_bytes + step * count + (4 << 20), dtype=torch.uint8, device="cuda")
base = big.data_ptr()
inp = big[:in_bytes].view(n_clips, per_enc)
inp.copy_(enc0)
del enc0
dplan = batch.DecodePlan.from_clips(clips)
o0 = (in_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
print(f"base {base:#x}  input at +0 ({in_bytes / 2**20:.1f} MB)  outputs from +{o0 / 2**20:.0f} MB in steps of {step / 2**20:.2f} MB")
for k in range(count):
    off = o0 + k * step
    out = big[off:off + out_bytes].view(torch.int16).view(n_clips, samples, ch)
    dplan.run(inp, out)
    torch.cuda.synchronize()
    r = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dplan.run(inp, out); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1))
    print(f"out at +{off / 2**20:9.2f} MB (addr {base + off:#x}): avg {sum(r) / len(r):.3f} best {min(r):.3f}")
