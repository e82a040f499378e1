#!/usr/bin/env python3
"""Where K1's slice time goes (lab build with in-kernel cycle stamps: tools/k1lab/build_lab.sh stamps stamps -DQOA_LAB_STAMPS).
Run on the GPU box: QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/lib_stamps.so python tools/k1lab/stamps.py [frames [clips]]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth, _lib  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_clips, ch = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 2
samples = frames * 5120
pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda")
per, enc = samples * ch, gpu.encoded_size(ch, samples)
clips = [(i * per, i * enc, gpu.QoaDesc(ch, 44100, samples)) for i in range(n_clips)]
plan = batch.EncodePlan(clips, batch.PLAN_COUNTERS | int(os.environ.get("QOA_EXTRA_FLAGS", "0")))
out = torch.zeros(plan.out_bytes, dtype=torch.uint8, device="cuda")
plan.run(pcm.view(-1), out)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 19)()
lib = _lib.lib()
lib.qoa_encode_plan_counters(plan._h, buf, 19)
c = list(buf)
c = c[:8] + c[10:]  # (the lab's stamps follow the product's 10 counters)
waves = plan.launch_info()[0]
ws = waves * frames * 256
print("counters", c[:8])
print(f"per wave-slice: steps {c[8] / ws:.1f} cycles, rest of the slice loop {c[9] / ws:.1f}, statement {c[10] / ws:.1f}, "
      f"kernel {c[11] / ws:.1f}  (each stamp costs its own latency: two per slice)")
wb = ws / 16
print(f"per wave-block: back edge + tile store {c[12] / wb:.0f}, flush + header {c[13] / wb:.0f}, prefetch issue {c[14] / wb:.0f}, "
      f"statements + dispatch {c[15] / wb:.0f} (statement alone {c[10] / wb:.0f})")
