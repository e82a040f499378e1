#!/usr/bin/env python3
"""Which wavefront makes a hard class slow, and on which step forms does it spend its slices?  Needs a -DQOA_LAB_HWID build
(per compute wavefront: cycles, and wavefront-slices attempted per form):

    bash tools/k1lab/build_lab.sh hwid "" -DQOA_LAB_HWID
    QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/lib_hwid.so python tools/k1lab/per_wave.py [loud|white|music|fixture] [frames]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth, _lib  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "loud"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_clips, ch = 256, 2
samples = frames * 5120
if kind == "loud":
    pcm = (synth.make_clips(n_clips, samples, ch, 44100, device="cuda").to(torch.int32) * 3).clamp_(-32768, 32767).to(torch.int16)
elif kind == "fixture":  # PCM cut from the decoded reference fixture, as bench.py's signal_classes does
    from oracle import oracle as O
    fch, _, fpcm = O.decode_all(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden",
                                                  "julien_baker_sprained_ankle.qoa"), "rb").read())
    song = torch.from_numpy(fpcm.reshape(-1, fch))
    offs = (np.arange(n_clips) * 997 * 20) % (song.shape[0] - samples)
    pcm = torch.stack([song[o:o + samples] for o in offs]).to("cuda").contiguous()
else:
    pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda", kind=kind)
desc = gpu.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
plan = batch.EncodePlan(clips, batch.PLAN_COUNTERS | int(os.environ.get("QOA_FLAGS", "0")))
plan.run(pcm, out)
torch.cuda.synchronize()
N = 10 + 4096 + 2048 + 2048 * 8
buf = (ctypes.c_uint64 * N)()
_lib.lib().qoa_encode_plan_counters(plan._h, buf, N)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
waves, cpw, _ = plan.launch_info()
cyc = a[10 + 4096:10 + 4096 + waves] / (frames * 256)
forms = a[10 + 4096 + 2048:10 + 4096 + 2048 + waves * 8].reshape(waves, 8)
names = ["cheap", "exact", "pen", "wide", "full", "b32", "loop", "blocks"]
print(f"{kind}: {waves} compute wavefronts x {cpw} chains, {frames * 256} slices each; cycles per slice: min {cyc.min():.0f} "
      f"p50 {np.median(cyc):.0f} p90 {np.quantile(cyc, .9):.0f} p99 {np.quantile(cyc, .99):.0f} max {cyc.max():.0f}")
print("wavefront-slices attempted per form, all wavefronts: " + ", ".join(f"{n} {int(forms[:, i].sum())}" for i, n in enumerate(names)))
order = np.argsort(-cyc)
print("the slowest wavefronts:")
for w in order[:12]:
    print(f"  wave {w:4d}: {cyc[w]:6.0f} cycles/slice  " + "  ".join(f"{n} {int(forms[w, i])}" for i, n in enumerate(names)))
print("every 16th of the rest:")
for w in order[12::max(1, waves // 16)]:
    print(f"  wave {w:4d}: {cyc[w]:6.0f} cycles/slice  " + "  ".join(f"{n} {int(forms[w, i])}" for i, n in enumerate(names)))
# how well does a linear cost model explain the cycles?  (least squares: cycles = sum_k cost_k * attempts_k)
X = forms[:, :7].astype(np.float64) / (frames * 256)
coef, *_ = np.linalg.lstsq(X, cyc.astype(np.float64), rcond=None)
print("least-squares cycles per attempted wavefront-slice: " + ", ".join(f"{n} {c:.0f}" for n, c in zip(names, coef)))
