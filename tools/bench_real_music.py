"""Config-3 shape (1024 stereo clips x 256 frames) cut from the reference fixture's real music instead of the synthetic
signal: kernel time of the step forms and the fast paths' statistics.  Run on the GPU box."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import qoaudio_amd as q
from qoaudio_amd import batch
from oracle import oracle as O
fx = open('/root/repo/tests/golden/julien_baker_sprained_ankle.qoa', 'rb').read()
d = q.decode_all(fx)
pcm_all = torch.from_numpy(d.samples.reshape(-1, 2))           # [2394122, 2]
n_clips, frames = 1024, 256
samples = frames * 5120
maxoff = pcm_all.shape[0] - samples
offs = (np.arange(n_clips) * 997 * 20) % maxoff                # slice-aligned starts spread over the song
pcm = torch.stack([pcm_all[o:o + samples] for o in offs]).cuda().contiguous()
desc = q.QoaDesc(2, 44100, samples)
clips, _, per_enc = batch.uniform_clips(n_clips, desc)
for name, fl in (("default", 0), ("exact quotient", batch.PLAN_EXACT_QUOTIENT), ("no speculation", batch.PLAN_NO_SPECULATION),
                 ("32-bit step", batch.PLAN_FORCE_32BIT_STEP)):
    plan = batch.EncodePlan(clips, fl | batch.PLAN_COUNTERS)
    out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device="cuda")
    plan.run(pcm, out); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(pcm, out); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    st = plan.counters()
    print(f"{name:16s} {min(ms):7.2f} ms  {plan.samples / (min(ms) * 1e-3) / 1e9:6.2f} G samples/s  per launch: (declined, unpacked, "
          f"speculation misses, cheap-quotient declines, wavefront-slices with a marked candidate) = {tuple(v // 4 for v in st)}")
    if fl == 0:
        ok = out[5].cpu().numpy().tobytes() == O.encode_all(pcm[5].cpu().numpy().reshape(-1), 2, 44100, samples)
        print("clip 5 bit-exact vs oracle:", ok, " slices per launch:", n_clips * 2 * samples // 20)
