#!/usr/bin/env python3
"""Replays tests/test_gpu_fuzz.py::test_fuzz_block_loop_from_hostile_states[11] up to a given trial and examines that one.
python tools/k1lab/repro_fuzz11.py [trial=391]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rng_pcm  # noqa: E402

target = int(sys.argv[1]) if len(sys.argv) > 1 else 391
r = np.random.default_rng(11)
channels, n = 2, 640
for trial in range(target + 1):
    amp = int(r.choice([3000, 9000, 12000, 12288, 14000, 15900, 16300]))
    state = np.zeros((channels, 8), dtype=np.int32)
    for c in range(channels):
        w = r.integers(-amp, amp + 1, 4)
        if r.random() < 0.5:
            w = np.abs(w) * np.sign(r.integers(0, 2) * 2 - 1)
        h = r.integers(-32768, 32768, 4)
        if r.random() < 0.5:
            h = np.full(4, int(r.choice([-32768, 32767])))
        state[c] = list(h) + list(w)
    kind = str(r.choice(["rail", "flip", "white", "mix", "ramp"]))
    if kind == "rail":
        x = np.full(n * channels, int(r.choice([-32768, 32767])), dtype=np.int16)
    elif kind == "flip":
        x = rng_pcm(int(r.integers(1 << 30)), n * channels, "loud")
    elif kind == "ramp":
        x = np.clip(np.linspace(-40000, 40000, n * channels) * r.choice([-1, 1]), -32768, 32767).astype(np.int16)
    else:
        x = rng_pcm(int(r.integers(1 << 30)), n * channels, kind)
print("trial", target, kind, "x[0]", int(x[0]), "state", state.tolist())
np.save("/tmp/repro_state.npy", state)
np.save("/tmp/repro_x.npy", x)
from oracle import oracle as O  # noqa: E402
for c in range(channels):
    s, sf, end, totals = O.encode_slice(list(state[c][:4]), list(state[c][4:]), x[c::channels][:20])
    print("oracle ch", c, "sf", sf, "totals", [int(t) for t in totals])
    try:
        terms = O.slice_rank_terms(list(state[c][:4]), list(state[c][4:]), x[c::channels][:20])
        print("   terms", terms)
    except Exception as e:  # noqa: BLE001
        pass
if os.environ.get("NO_GPU") == "1":
    sys.exit(0)
import qoaudio_amd as gpu  # noqa: E402
enc = gpu.QoaEncoder(gpu.QoaDesc(channels, 44100, n))
enc.set_lms_state(state)
fl, data = enc.encode_frame(x)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _oracle_with_state  # noqa: E402
import oracle.oracle as OM  # noqa: E402
ref = _oracle_with_state(OM, channels, 44100, n, state)
want, _ = ref.encode_frame(x)
bad = [i for i in range(len(want)) if data[i] != want[i]]
print("device vs oracle: first differing bytes", bad[:8], "of", len(want))
if bad:
    k = (bad[0] - 40) // 8
    print("slice", k, "row", k // channels, "channel", k % channels, "device", data[40 + 8 * k:48 + 8 * k].hex(), "oracle", want[40 + 8 * k:48 + 8 * k].hex())
