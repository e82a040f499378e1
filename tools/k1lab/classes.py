#!/usr/bin/env python3
"""K1 on the signal classes of SURVEY 8(d) at config 3's chain count (CLIPS stereo clips x FRAMES frames): kernel time and the
plan counters per class and plan flag set.  Run on the GPU box:  python tools/k1lab/classes.py [frames [clips]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_clips, ch = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 2
samples = frames * 5120
per = samples * ch
enc = gpu.encoded_size(ch, samples)
clips = [(i * per, i * enc, gpu.QoaDesc(ch, 44100, samples)) for i in range(n_clips)]
kinds = (os.environ.get("QOA_CLASSES") or "music,white,silence,loud").split(",")
only_default = os.environ.get("QOA_DEFAULT_ONLY") == "1"
extra = int(os.environ.get("QOA_EXTRA_FLAGS", "0"))  # e.g. batch.PLAN_FORCE_HELPER = 8192, PLAN_NO_HELPER = 4096
for kind in kinds:
    if kind == "fixture":  # clips cut from the decoded reference fixture at offsets spread over the whole song
        import numpy as np
        from oracle import oracle as O
        _, _, fp = O.decode_all(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden",
                                                  "julien_baker_sprained_ankle.qoa"), "rb").read())
        song = torch.from_numpy(fp.reshape(-1, 2))
        offs = (np.arange(n_clips) * 997 * 20) % (song.shape[0] - samples)
        pcm = torch.stack([song[o:o + samples] for o in offs]).cuda().contiguous()
    elif kind == "periodic":  # period = one 16-slice block (for lab builds that reuse the first tile: -DQOA_LAB_NO_BLOCKTOP)
        import numpy as np
        t = np.arange(320)
        rng = np.random.default_rng(5)
        x = np.zeros((n_clips, 320, ch))
        for h in (1, 2, 3, 5, 8, 13):
            ph = rng.uniform(0, 2 * np.pi, (n_clips, 1, ch))
            x += (3000.0 / h) * np.sin(2 * np.pi * h * t[None, :, None] / 320 + ph)
        pcm = torch.from_numpy(np.rint(np.tile(x, (1, samples // 320, 1))).astype(np.int16)).cuda()
    elif kind == "loud":  # the music class amplified 3x into clipping
        pcm = (synth.make_clips(n_clips, samples, ch, 44100, device="cuda").to(torch.int32) * 3).clamp_(-32768, 32767).to(torch.int16)
    else:
        pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda", kind=kind)
    ref = None
    for name, fl in ((("default", 0),) if only_default else
                     (("default", 0), ("no penalty form", batch.PLAN_NO_PENALTY_FORM), ("exact quotient", batch.PLAN_EXACT_QUOTIENT),
                      ("no speculation", batch.PLAN_NO_SPECULATION))):
        plan = batch.EncodePlan(clips, fl | extra | batch.PLAN_COUNTERS)
        out = torch.zeros(plan.out_bytes, dtype=torch.uint8, device="cuda")
        plan.run(pcm.view(-1), out)
        torch.cuda.synchronize()
        s0 = plan.counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            plan.run(pcm.view(-1), out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(ref, out))
        slices = n_clips * ch * frames * 256
        print(f"{kind:8s} {name:16s} {ms:8.3f} ms  {ms * 1e6 / (frames * 256):7.1f} ns/slice  declined/unpacked/spec-miss/cheap-declined/marked "
              f"{s0} of {slices} slices  bytes equal: {same}", flush=True)
