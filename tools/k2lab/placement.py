#!/usr/bin/env python3
"""Is K2u's run-to-run spread (1.38-1.6 ms for the same launch) a property of WHERE the buffers are?  One process, one
encoded batch; decode into several separately allocated output buffers (and from several copies of the input), interleaved.
python tools/k2lab/placement.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

pre = [torch.empty(int(float(g) * (1 << 30)), dtype=torch.uint8, device="cuda") for g in os.environ.get("PRE_GB", "").split(",") if g]
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
outs, ins = [], [enc]
junk = []
for k in range(int(os.environ.get('N_OUT', '5'))):
    junk.append(torch.empty((k + 1) * 37 * 1024 * 1024 + 4096 * k, dtype=torch.uint8, device="cuda"))  # shifts the next allocation
    outs.append(torch.empty((n_clips, samples, ch), dtype=torch.int16, device="cuda"))
for k in range(int(os.environ.get('N_IN', '2'))):
    junk.append(torch.empty((k + 1) * 53 * 1024 * 1024, dtype=torch.uint8, device="cuda"))
    ins.append(enc.clone())


def t(i, o, n=6):
    dplan.run(ins[i], outs[o])
    torch.cuda.synchronize()
    r = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dplan.run(ins[i], outs[o]); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1))
    return sum(r) / len(r), min(r)


for rep in range(int(os.environ.get('REPS', '3'))):
    for i in range(len(ins)):
        row = []
        for o in range(len(outs)):
            a, b = t(i, o)
            row.append(f"{a:.3f}/{b:.3f}")
        print(f"rep {rep} input {i} (ptr % 2MB = {ins[i].data_ptr() % (2 << 20)}): " + "  ".join(row))
print("inputs :", [hex(i.data_ptr()) for i in ins])
print("outputs:", [hex(o.data_ptr()) for o in outs])
