#!/usr/bin/env python3
"""K1 on small launches (8 .. 1024 chains): ns per slice per plan packing (chains per wavefront) with and without helper
wavefronts, and -- with a -DQOA_LAB_HWID build of the library -- where the wavefronts landed (HW_ID / XCC_ID per
wavefront: how many SIMDs hold two of them).  Round 3 shipped a 2x regression for launches of <= 512 chains that only this
kind of sweep sees.  Run on the GPU box:

    python tools/k1lab/small_launch.py [frames]                       # timings, product library
    QOA_HWID=1 QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/lib_hwid.so python tools/k1lab/small_launch.py [frames]
"""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qoaudio_amd import batch, synth, _lib  # noqa: E402
import qoaudio_amd as gpu  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hwid = os.environ.get("QOA_HWID") == "1"
shapes = [(4, 2), (32, 2), (128, 2), (256, 2), (512, 2), (1, 8), (8, 8), (64, 8), (512, 1)]
if os.environ.get("QOA_SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["QOA_SHAPES"].split(",")]
B = batch
forms = [("default", 0), ("no helper", B.PLAN_NO_HELPER),
         ("cpw1 helper", B.PLAN_CPW_1 | B.PLAN_FORCE_HELPER), ("cpw1 classic", B.PLAN_CPW_1 | B.PLAN_NO_HELPER),
         ("cpw2 helper", B.PLAN_CPW_2 | B.PLAN_FORCE_HELPER), ("cpw2 classic", B.PLAN_CPW_2 | B.PLAN_NO_HELPER),
         ("cpw4 helper", B.PLAN_CPW_4 | B.PLAN_FORCE_HELPER), ("cpw4 classic", B.PLAN_CPW_4 | B.PLAN_NO_HELPER)]


def placement(plan, n_waves_max=4096):
    buf = (ctypes.c_uint64 * (10 + n_waves_max + 2048))()
    _lib.lib().qoa_encode_plan_counters(plan._h, buf, 10 + n_waves_max + 2048)
    simd = collections.defaultdict(list)
    cus = set()
    for w in range(n_waves_max):
        v = buf[10 + w]
        if not v >> 63:
            continue
        xcc, hw = (v >> 32) & 15, v & 0xffffffff
        cu = (xcc, (hw >> 8) & 0xff)
        cus.add(cu)
        simd[(cu, (hw >> 4) & 3)].append("h" if w & 1 else "c")
    kinds = collections.Counter("".join(sorted(v)) for v in simd.values())
    text = f"{len(cus)} CUs, {len(simd)} SIMDs; waves per SIMD (c = compute, h = helper): {dict(kinds)}"
    # compute wavefronts by what else sits on their SIMD: cycles per slice (the launch lasts as long as its slowest wavefront)
    by = collections.defaultdict(list)
    for w in range(0, n_waves_max, 2):
        v = buf[10 + w]
        if v >> 63:
            hw = v & 0xffffffff
            key = "".join(sorted(simd[(((v >> 32) & 15, (hw >> 8) & 0xff), (hw >> 4) & 3)]))
            by[key].append(buf[10 + 4096 + w // 2] / (frames * 256))
    text += "; compute cycles/slice by SIMD population: " + ", ".join(
        f"{k}: n {len(v)} mean {sum(v) / len(v):.0f} max {max(v):.0f}" for k, v in sorted(by.items()))
    allv = sorted(x for v in by.values() for x in v)
    if len(allv) >= 8:
        q = lambda f: allv[min(len(allv) - 1, int(f * len(allv)))]
        text += f"; all: min {allv[0]:.0f} p10 {q(.1):.0f} p50 {q(.5):.0f} p90 {q(.9):.0f} p99 {q(.99):.0f} max {allv[-1]:.0f}"
    if os.environ.get("QOA_HWID_DETAIL") == "1":  # per workgroup: XCC / CU / SIMD of the compute and helper wavefront, compute cycles
        rows = []
        for w in range(0, n_waves_max, 2):
            c, h = buf[10 + w], buf[10 + w + 1]
            if not c >> 63:
                continue
            f = lambda v: f"x{(v >> 32) & 15} se{(v >> 13) & 7} cu{(v >> 8) & 15} simd{(v >> 4) & 3}"
            rows.append(f"      wg {w // 2}: compute {f(c)}" + (f"  helper {f(h)}" if h >> 63 else "") +
                        f"  cycles/slice {buf[10 + 4096 + w // 2] / (frames * 256):.0f}")
        text += "\n" + "\n".join(rows[:int(os.environ.get("QOA_HWID_ROWS", "16"))])
    return text


for n_clips, ch in shapes:
    samples = frames * 5120
    pcm = synth.make_clips(n_clips, samples, ch, 44100, device="cuda")
    if os.environ.get("QOA_SAME") == "1":  # every chain gets the same samples (is a slow workgroup slow because of its data?)
        pcm = pcm[:1, :, :1].expand(n_clips, samples, ch).contiguous()
    elif os.environ.get("QOA_SAME") == "0":
        pcm = torch.zeros_like(pcm)
    per, enc = samples * ch, gpu.encoded_size(ch, samples)
    clips = [(i * per, i * enc, gpu.QoaDesc(ch, 44100, samples)) for i in range(n_clips)]
    ref = None
    print(f"--- {n_clips} clips x {ch} ch = {n_clips * ch} chains, {frames} frames", flush=True)
    for name, fl in forms:
        plan = batch.EncodePlan(clips, fl | (batch.PLAN_COUNTERS if hwid else 0))
        out = torch.zeros(plan.out_bytes, dtype=torch.uint8, device="cuda")
        plan.run(pcm.view(-1), out)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.run(pcm.view(-1), out)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        where = placement(plan) if hwid else ""  # (of the last launch)
        if ref is None:
            ref = out.clone()
        waves, cpw, _ = plan.launch_info()
        print(f"  {name:13s} cpw {cpw} waves {waves:4d}: {min(ms):8.3f} ms  {min(ms) * 1e6 / (frames * 256):7.1f} ns/slice  "
              f"bytes equal: {bool(torch.equal(ref, out))}  {where}", flush=True)
