"""Synthetic PCM clips for tests and bench.py ("music-like" primary class).

Per channel: 4 sinusoids (log-uniform 80-6000 Hz, amplitude 500-8000, a 0.5-4 Hz tremolo on the first)
plus Gaussian noise sigma=300, clamped to i16; channel c>0 re-uses 0.7x of channel 0's partials at a new
phase plus independent noise.  Parameters come from a CPU generator seeded with 0x51EDC0DE + clip index,
the sample arithmetic runs in float64 on whatever device is asked for (the GPU for the big bench batches;
float32 would put ~0.06 rad of phase noise on a 6 kHz partial 30 s into a clip).  The
same tensor is handed to the GPU path and -- copied to the host -- to the CPU oracle, so generator
float behaviour never enters a parity comparison.
"""
import math

import torch

SEED0 = 0x51EDC0DE


def _clip_params(clip_index: int, channels: int):
    g = torch.Generator(device="cpu")
    g.manual_seed(SEED0 + clip_index)
    u = torch.rand((channels, 4, 3), generator=g, dtype=torch.float64)
    freq = 80.0 * torch.exp(u[0, :, 0] * math.log(6000.0 / 80.0))  # shared partials
    amp = 500.0 + u[0, :, 1] * 7500.0
    phase = u[:, :, 2] * (2 * math.pi)  # per channel
    trem = 0.5 + torch.rand((1,), generator=g, dtype=torch.float64) * 3.5
    noise_seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g))
    return freq, amp, phase, float(trem), noise_seed


def make_clips(n_clips: int, samples: int, channels: int, sample_rate: int = 44100, device="cpu",
               first_index: int = 0, kind: str = "music", chunk: int = 32) -> torch.Tensor:
    """int16 tensor [n_clips, samples, channels] on `device`."""
    dev = torch.device(device)
    out = torch.empty((n_clips, samples, channels), dtype=torch.int16, device=dev)
    if kind == "silence":
        return out.zero_()
    if kind == "white":
        g = torch.Generator(device=dev)
        g.manual_seed(SEED0 + first_index)
        for i in range(0, n_clips, chunk):
            n = min(chunk, n_clips - i)
            out[i:i + n] = torch.randint(-32768, 32768, (n, samples, channels), generator=g, device=dev,
                                         dtype=torch.int32).to(torch.int16)
        return out
    assert kind == "music"
    fdt = torch.float64
    t = torch.arange(samples, device=dev, dtype=fdt) / float(sample_rate)
    for i in range(n_clips):
        freq, amp, phase, trem, nseed = _clip_params(first_index + i, channels)
        g = torch.Generator(device=dev)
        g.manual_seed(nseed)
        clip = torch.randn((samples, channels), generator=g, device=dev, dtype=fdt) * 300.0
        for c in range(channels):
            scale = 1.0 if c == 0 else 0.7
            for k in range(4):
                w = torch.sin(t * (2 * math.pi * float(freq[k])) + float(phase[c, k]))
                a = float(amp[k]) * scale
                if k == 0:
                    w = w * (0.75 + 0.25 * torch.sin(t * (2 * math.pi * trem)))
                clip[:, c] += a * w
        out[i] = clip.clamp_(-32768, 32767).round_().to(torch.int16)
    return out
