import sys, time, json
sys.path.insert(0, '.')
import torch, numpy as np
import qoaudio_amd as q
from qoaudio_amd import batch, synth
dev = torch.device('cuda:0')
n_clips, frames, ch = 1024, 256, 2
samples = frames * 5120
desc = q.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
pcm = synth.make_clips(64, samples, ch, device=dev).repeat((16, 1, 1)).contiguous()
out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
plan = batch.EncodePlan(clips); plan.run(pcm, out); torch.cuda.synchronize()
dplan = batch.DecodePlan.from_clips(clips)
back = torch.empty_like(pcm)
def dec(n):
    r = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dplan.run(out, back); e1.record(); torch.cuda.synchronize(); r.append(round(e0.elapsed_time(e1), 3))
    return r
print("right after encode:", dec(8))
time.sleep(1.0)
print("after 1 s idle:    ", dec(8))
for _ in range(5): plan.run(pcm, out)
torch.cuda.synchronize()
print("after 5 encodes:   ", dec(8))
time.sleep(0.2)
print("after 0.2 s idle:  ", dec(8))
# back-to-back without sync between
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dplan.run(out, back)
e1.record(); torch.cuda.synchronize()
print("20 back to back, avg:", e0.elapsed_time(e1) / 20)
