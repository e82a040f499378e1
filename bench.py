"""bench.py -- encoded samples/s of the QOA encode hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: every rank encodes 1024 synthetic stereo 44.1 kHz
clips of 256 frames (BASELINE config 3's shape; 2,684,354,560 int16 samples, 5.37 GB of PCM resident in
HBM before the timed region) and, for N > 1, the encoded buffers (1.08 GB per rank) are gathered to
rank 0 with RCCL (config 4).  value = samples all ranks encoded / time of the K steps (max over ranks).

N > 1 runs one process per GPU.  Under `python -m torch.distributed.run` the ranks already exist (RANK /
WORLD_SIZE in the environment); started plainly as `python bench.py --gpus N` this process becomes a launcher
that never touches the GPU: it starts N rank processes of this same script and exits with their status.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


# everything that decides what a kernel executes: its text, the generated rows / slices, the generators, and the host code
# that picks step forms and splits the decode tables
KERNEL_SOURCES = {"encode": ("qoaudio_amd/csrc/qoa_kernels.hip", "qoaudio_amd/csrc/qoa_slice_asm.inc",
                             "qoaudio_amd/csrc/qoa_device.h", "tools/gen_slice_asm.py", "qoaudio_amd/csrc/qoa_capi.cpp"),
                  "decode": ("qoaudio_amd/csrc/qoa_decode.hip", "qoaudio_amd/csrc/qoa_decode_asm.inc",
                             "qoaudio_amd/csrc/qoa_device.h", "tools/gen_decode_asm.py", "qoaudio_amd/csrc/qoa_capi.cpp")}


def kernel_source_sha256(kernel: str) -> str:
    """sha256 over the source files a kernel is built from: ties a committed PMC profile to the build it was taken on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel]:
        with open(os.path.join(ROOT, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic(kernel: str, clips: int, channels: int, frames: int):
    """(HBM bytes per launch, note) from the committed rocprofv3 PMC profile (profiles/pmc_traffic.json: separate
    --pmc passes, FETCH_SIZE doubled per the gfx950 correction; tools/pmc.sh + tools/pmc_traffic.py write it).  PMC
    counters cannot be read from inside this process, so the number is only reported for the exact workload AND the
    exact kernel sources it was collected on; otherwise null with the reason."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if (clips, channels, frames) != (1024, 2, 256):
        return None, "PMC profile was collected on config 3 (1024 x 2 ch x 256 frames) only"
    if d[kernel].get("source_sha256") != kernel_source_sha256(kernel):
        return None, "traffic_stale: kernel sources changed since the PMC profile " + str(d.get("source"))
    return d[kernel]["traffic_bytes_per_launch"], None


def pmc_valu_per_slice(step_form: str):
    """VALU wave-instructions per slice and wavefront of a K1 step form as the PMC run counted them (SQ_INSTS_VALU /
    (SQ_WAVES x slices per chain), tools/pmc_traffic.py), or None when the committed profile is of other kernel sources or
    has no entry for the form."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    if d["encode"].get("source_sha256") != kernel_source_sha256("encode"):
        return None
    return d["encode"].get("valu_per_slice", {}).get(step_form)


def valu_issue(plan, samples: int, kernel_ms: float):
    """Secondary roofline for K1: VALU wave-instructions issued per second against the chip's issue peak (a SIMD issues at
    most one instruction per 4 cycles whatever the wavefront count, profiles/r02_ubench_gfx950_issue_classes.txt).  The
    instruction count per slice is PMC-measured for the step form the plan's launch takes (qoa_encode_plan_launch_info);
    when the committed profile does not cover it `achieved` and `frac` are null."""
    waves, cpw, lds, helper = plan.launch_shape()
    form = "lds_table" if lds else "cheap_quotient"
    per_slice = pmc_valu_per_slice(form)
    slices = (samples + 19) // 20
    peak = 1024 * 2.4e9 / 4.0
    achieved = per_slice * slices * waves / (kernel_ms * 1e-3) if per_slice else None
    return {"bound": "valu-issue", "achieved": achieved, "peak": peak, "unit": "wave-instructions/s",
            "frac": achieved / peak if achieved else None, "wavefronts": waves, "chains_per_wavefront": cpw,
            "step_form": form, "valu_per_slice": per_slice,
            "helper_wavefronts": helper}  # as the plan reports it (launches of up to 512 compute wavefronts)


def cpu_baseline_child(pcm_host: np.ndarray, rate: int, budget_s: float):
    """The CPU oracle's ref_style encoder (the reference's algorithm incl. its rank-sorted early exits) on a bounded
    sample of the same clips: one thread (generic x86-64, as the crate builds; then -march=native on a third of the
    budget), then all host cores.  Runs in child processes (oracle/cpu_bench.py) that never touch the GPU, so the worker
    pool forks from a clean process.  kind = "port": the Rust reference cannot be built here."""
    need = pcm_host.nbytes + (1 << 20)
    tmpdir = None
    if os.path.isdir("/dev/shm"):
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 2 * need:
            tmpdir = "/dev/shm"
    fd, path = tempfile.mkstemp(prefix="qoa_bench_", suffix=".npy", dir=tmpdir)
    os.close(fd)
    try:
        np.save(path, pcm_host)
        cmd = [sys.executable, "-m", "oracle.cpu_bench", path, str(rate)]
        out = subprocess.run(cmd + [str(budget_s)], cwd=ROOT, check=True, capture_output=True, text=True).stdout
        res = json.loads(out.strip().splitlines()[-1])
        try:
            env = dict(os.environ, QOA_ORACLE_LIB="liboracle_native.so")
            out = subprocess.run(cmd + [str(budget_s / 3), "--no-all-cores", "--rebuild"], cwd=ROOT, check=True,
                                 capture_output=True, text=True, env=env).stdout
            res["march_native"] = json.loads(out.strip().splitlines()[-1])
        except (subprocess.CalledProcessError, OSError, ValueError) as e:
            res["march_native"] = {"error": str(e)[:200]}
    finally:
        os.unlink(path)
    return res


def timed_launches(run, stream, reps=3, warm=1):
    """kernel ms of `reps` launches of run() on `stream`, each bracketed by its own HIP event pair, behind `warm` untimed ones"""
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); run(); e1.record(stream)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return ms


def small_launches(q, batch, synth, dev, stream, rate, frames=16):
    """K1 on launches far below the headline's size -- 8, 64 and 512 chains (stereo), and ONE 8-channel clip -- in ns per
    slice (kernel time / slices per chain).  Round 3 shipped a 2x regression for launches of <= 512 chains that no record in
    the bench line could see; this one does (default plan, as any caller gets it)."""
    from oracle import oracle as O
    samples = frames * 5120
    res = {"workload": f"stereo clips x {frames} frames, default plan; ns per slice = kernel time / {frames * 256} slices per chain"}
    for chains, ch in ((8, 2), (64, 2), (512, 2), (8, 8)):
        n_clips = chains // ch
        desc = q.QoaDesc(ch, rate, samples)
        clips, _, per_enc = batch.uniform_clips(n_clips, desc)
        pcm = synth.make_clips(n_clips, samples, ch, rate, device=dev, first_index=4096)
        out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
        plan = batch.EncodePlan(clips)
        ms = min(timed_launches(lambda: plan.run(pcm, out, stream), stream))
        waves, cpw, _, helper = plan.launch_shape()
        ok = out[n_clips - 1].cpu().numpy().tobytes() == O.encode_all(pcm[n_clips - 1].cpu().numpy().reshape(-1), ch, rate, samples)
        res[f"{chains}_chains" + ("_8ch" if ch == 8 else "")] = {
            "kernel_ms": ms, "ns_per_slice": ms * 1e6 / (frames * 256), "wavefronts": waves, "chains_per_wavefront": cpw,
            "helper_wavefronts": helper, "bit_exact_vs_oracle": bool(ok)}
    return res


def other_configs(q, batch, synth, dev, stream, pcm, rate, n_clips, frames):
    """BASELINE.json's configs 2 and 5 for the record (not part of `value`): config 2 = the headline's clips as mono (channel
    0), encode only; config 5 = the frame-at-a-time API on ONE 8-channel 48 kHz stream (200 calls of 5120 sample-times,
    host pointers in, host bytes out, LMS state carried in the handle), against the oracle's one-shot bytes."""
    from oracle import oracle as O
    res = {}
    samples = frames * 5120
    mdesc = q.QoaDesc(1, rate, samples)
    mclips, _, mper_enc = batch.uniform_clips(n_clips, mdesc)
    mpcm = pcm[:, :, 0].contiguous().view(n_clips, samples, 1)
    menc = torch.empty((n_clips, mper_enc), dtype=torch.uint8, device=dev)
    mplan = batch.EncodePlan(mclips)
    ms = float(np.mean(timed_launches(lambda: mplan.run(mpcm, menc, stream), stream, reps=2)))
    k = n_clips // 2
    ok = menc[k].cpu().numpy().tobytes() == O.encode_all(mpcm[k].cpu().numpy().reshape(-1), 1, rate, samples)
    res["cfg2_mono_encode"] = {"workload": f"{n_clips} clips x 1 ch x {frames} frames, encode", "kernel_ms": ms,
                               "ns_per_slice": ms * 1e6 / (frames * 256), "samples_per_s": mplan.samples / (ms * 1e-3),
                               "bit_exact_vs_oracle": bool(ok)}
    del mplan, menc, mpcm
    ch5, rate5, calls = 8, 48000, 200
    x = synth.make_clips(1, calls * 5120, ch5, rate5, device=dev, first_index=8192)[0].cpu().numpy().reshape(-1)
    enc = q.QoaEncoder(q.QoaDesc(ch5, rate5, calls * 5120))
    got = bytearray()
    enc.write_header(got)
    enc.encode_frame(x[:5120 * ch5])  # warm-up call on a throw-away handle state
    enc = q.QoaEncoder(q.QoaDesc(ch5, rate5, calls * 5120))
    t0 = time.perf_counter()
    for f in range(calls):
        enc.encode_frame(x[f * 5120 * ch5:(f + 1) * 5120 * ch5], got)
    dt = time.perf_counter() - t0
    want = O.encode_all(x, ch5, rate5, calls * 5120)
    res["cfg5_stream"] = {"workload": f"1 stream x {ch5} ch x {rate5} Hz, {calls} encode_frame calls of 5120 sample-times "
                                      f"(81,920 B in / 16,520 B out per call, host memory both sides)",
                          "ms_per_call": dt / calls * 1e3, "x_realtime": calls * 5120 / rate5 / dt,
                          "bit_exact_vs_oracle": bytes(got) == want}
    return res


def decode_over_placements(q, batch, dplan, enc, pcm_like, stream, pairs=4, warm=8, reps=10):
    """The decode launch on `pairs` separately allocated (input, output) buffer pairs: K2u's time depends on where its buffers
    lie (a stable property of the pair of allocations, LABLOG.md K2u item 6), so one pair reports the allocator's luck; the
    figure to quote is the median over pairs.  Returns (per-pair mean kernel ms, the last pair's output)."""
    per_pair, back = [], None
    keep = []  # (all pairs stay allocated while the others are timed: they must be distinct placements)
    for i in range(pairs):
        e = enc if i == 0 else enc.clone()
        b = torch.empty_like(pcm_like)
        keep.append((e, b))
    for e, b in keep:
        for _ in range(warm):
            dplan.run(e, b, stream)
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); dplan.run(e, b, stream); e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        per_pair.append(float(np.mean([a.elapsed_time(c) for a, c in evs])))
        back = b
    return per_pair, back


def decode_record(dplan, per_pair, kernel, traffic=None, traffic_note=None):
    d_bytes = dplan.in_bytes + dplan.pcm_bytes
    med = float(np.median(per_pair))
    rec = {"value": dplan.samples / (med * 1e-3), "unit": "samples/s", "kernel_ms_avg": med,
           "kernel_ms_per_placement": [round(v, 4) for v in per_pair], "kernel_ms_min": min(per_pair), "kernel_ms_max": max(per_pair),
           "roofline": {"bound": "hbm", "achieved": d_bytes / (med * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": d_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "frac_best_placement": d_bytes / (min(per_pair) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "frac_worst_placement": d_bytes / (max(per_pair) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "kernel": kernel, "algorithmic_bytes_per_launch": d_bytes},
           "note": f"median over {len(per_pair)} separately allocated (input, output) buffer pairs, each the mean of back-to-back "
                   "launches behind a warm-up: the same launch differs by several per cent between allocations (LABLOG.md K2u item 6)"}
    if traffic_note:
        rec["roofline"]["traffic_note"] = traffic_note
    return rec


def general_decode(q, batch, synth, dev, stream, rate):
    """The decode launches that are not whole wavefronts of full 1 / 2 / 4 / 8-channel frames: 6 and 3 channels (682 clips x 64
    frames: channel counts that do not divide a wavefront; and the same at config 3's amount of work), and a stereo batch whose
    clips end in a short frame like the reference's fixture (63 full frames + 3,082 sample-times)."""
    from oracle import oracle as O
    res = {}
    for name, n_clips, ch, samples, pad in (("6ch", 682, 6, 64 * 5120, 0), ("3ch", 682, 3, 64 * 5120, 0),
                                            # the same kernels on config 3's amount of work (5.37 GB of PCM, ~8300 wavefronts: the 682-clip
                                            # batches above are ONE generation of wavefronts, bound by a wavefront's lifetime)
                                            ("3ch_full_chip", 2731, 3, 64 * 5120, 0), ("6ch_full_chip", 1366, 6, 64 * 5120, 0),
                                            ("stereo_short_last_frame", 1024, 2, 63 * 5120 + 3082, 2),
                                            ("stereo_short_last_frame_packed", 1024, 2, 63 * 5120 + 3082, 0),
                                            ("stereo_short_last_frame_line_aligned", 1024, 2, 63 * 5120 + 3082, 22)):
        # pad: extra sample-times per clip in the PCM tensor so that every clip starts on a 16-byte boundary (325,642 stereo
        # sample-times are 8 bytes short of one); packed = clips back to back: every second clip's PCM is 8 bytes off the
        # 16-byte grid; line_aligned = every clip's PCM starts on a 128-byte line (the layout INTEGRATION.md recommends to callers
        # who own it: the full frames then take the streamed-store path of K2u)
        desc = q.QoaDesc(ch, rate, samples)
        clips, _, per_enc = batch.uniform_clips(n_clips, desc)
        clips = [(k * (samples + pad) * ch, oo, dd) for k, (po, oo, dd) in enumerate(clips)]
        pcm = synth.make_clips(n_clips, samples + pad, ch, rate, device=dev, first_index=12288)
        enc = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
        batch.EncodePlan(clips).run(pcm, enc, stream)
        dplan = batch.DecodePlan.from_clips(clips)
        per_pair, back = decode_over_placements(q, batch, dplan, enc, pcm, stream, pairs=2, warm=6, reps=8)
        k = n_clips - 1
        ok = np.array_equal(back[k, :samples].cpu().numpy().reshape(-1), O.decode_all(enc[k].cpu().numpy().tobytes())[2])
        rec = decode_record(dplan, per_pair, "qoa_decode_uniform_kernel" + (" + qoa_decode_frames_kernel" if name.startswith("stereo") else ""))
        rec.update({"workload": f"{n_clips} clips x {ch} ch x {samples} sample-times, decode", "header_mismatches": dplan.mismatches(),
                    "bit_exact_vs_oracle": bool(ok)})
        res[name] = rec
        del dplan, enc, pcm, back
    return res


def signal_classes(q, batch, synth, dev, stream, ch, rate, n_clips=256, frames=64):
    """K1 on the secondary signal classes of SURVEY 8(d) -- full-scale white noise (the CPU's worst realistic case),
    digital silence (its best), PCM cut from the decoded reference fixture (the reference's own bench input,
    benches/qoa_benchmarks.rs:17-30) -- beside the music-like class of the headline, all at 256 clips x 64 frames: kernel
    time, bytes against the oracle on one clip, and the plan counters (which step forms the material reached)."""
    from oracle import oracle as O
    samples = frames * 5120
    desc = q.QoaDesc(ch, rate, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
    res = {"workload": f"{n_clips} clips x {ch} ch x {frames} frames per class"}
    fixture = os.path.join(ROOT, "tests", "golden", "julien_baker_sprained_ankle.qoa")
    for kind in ("music", "white", "silence", "fixture", "loud"):
        if kind == "loud":  # the music class amplified 3x into hard clipping: the worst material found (tools/k1lab/classes.py)
            pcm = (synth.make_clips(n_clips, samples, ch, rate, device=dev).to(torch.int32) * 3).clamp_(-32768, 32767).to(torch.int16)
        elif kind == "fixture":
            if ch != 2 or not os.path.exists(fixture):
                continue
            fch, _, fpcm = O.decode_all(open(fixture, "rb").read())
            song = torch.from_numpy(fpcm.reshape(-1, fch))
            offs = (np.arange(n_clips) * 997 * 20) % (song.shape[0] - samples)  # slice-aligned starts spread over the song
            pcm = torch.stack([song[o:o + samples] for o in offs]).to(dev).contiguous()
        else:
            pcm = synth.make_clips(n_clips, samples, ch, rate, device=dev, kind=kind)
        plan = batch.EncodePlan(clips)
        plan.run(pcm, out, stream)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); plan.run(pcm, out, stream); e1.record(stream)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        k = n_clips // 3
        ok = out[k].cpu().numpy().tobytes() == O.encode_all(pcm[k].cpu().numpy().reshape(-1), ch, rate, samples)
        cplan = batch.EncodePlan(clips, batch.PLAN_COUNTERS)
        cplan.run(pcm, out, stream)
        c = cplan.counters()
        res[kind] = {"kernel_ms": float(np.mean(ms)), "ns_per_slice": float(np.mean(ms)) * 1e6 / (frames * 256),
                     "samples_per_s": plan.samples / (float(np.mean(ms)) * 1e-3),
                     "bit_exact_vs_oracle": bool(ok),
                     "counters": {"declined_slices": c[0], "unpacked_slices": c[1], "speculation_misses": c[2],
                                  "cheap_quotient_declines": c[3], "cheap_quotient_marked_candidates": c[4],
                                  "wavefront_slices_on_32bit_step": c[5], "penalty_form_declines": c[7],
                                  "wide_form_declines": c[8], "wavefront_slices_on_wide_form": c[9],
                                  "slices_total": n_clips * ch * frames * 256}}
        # what the class cost wavefront by wavefront (the recording instantiation of K1: one record per wavefront at its exit,
        # no atomics): measured cycles per slice over the wavefronts, the slowest one's form mix, and the bound it is held
        # against -- every slice on the 32-bit block statement, 842 issue slots x 4.06 cycles x 1.10 of loop overhead (include/qoaudio_amd.h)
        wplan = batch.EncodePlan(clips, batch.PLAN_WAVE_STATS)
        wplan.run(pcm, out, stream)
        ws = wplan.wave_stats()
        res[kind]["per_wavefront"] = {
            "cycles_per_slice_p50": round(ws["cycles_per_slice"]["p50"], 1), "cycles_per_slice_max": round(ws["cycles_per_slice"]["max"], 1),
            "bound_cycles_per_slice": round(ws["bound_cycles_per_slice"], 1),
            "max_over_bound": round(ws["cycles_per_slice"]["max"] / ws["bound_cycles_per_slice"], 3),
            "slowest_wavefront_form_mix": {k: v for k, v in ws["slowest_wavefront"]["form_mix"].items() if v},
            "slowest_predicted_cycles_per_slice": round(ws["slowest_wavefront"]["predicted_cycles_per_slice"], 1),
            "wavefront_slices_per_form": {k: v for k, v in ws["slices"].items() if v}}
        del plan, cplan, wplan, pcm
    if "music" in res:
        for kind in ("white", "silence", "fixture", "loud"):
            if kind in res:
                res[kind]["vs_music"] = res[kind]["kernel_ms"] / res["music"]["kernel_ms"]
    return res


def e2e_record(q, batch, dev, stream, pcm, out, clips, chunk_frames=8, reps=2):
    """Config 3 END TO END through the host-buffer boundary: pinned host PCM -> HBM -> K1 -> pinned host
    bytes with ONE call of qoa_host_pipeline_encode, and the way back with ONE call of qoa_host_pipeline_decode (time-sliced:
    the copy engines move slice j+1 / j-1 while slice j runs).  Wall time of the synchronous call; PCIe-inclusive, so never
    bench.py's `value`.  Bytes are compared with what the device-resident plans produced from the same PCM."""
    n_clips = pcm.shape[0]
    host_pcm = torch.empty(pcm.shape, dtype=torch.int16, pin_memory=True)
    host_pcm.copy_(pcm)
    host_out = torch.empty(out.shape, dtype=torch.uint8, pin_memory=True)
    host_back = torch.empty(pcm.shape, dtype=torch.int16, pin_memory=True)
    torch.cuda.synchronize()
    pipe = batch.HostPipeline()
    rec = {"workload": f"{n_clips} clips x {pcm.shape[2]} ch x {pcm.shape[1]} sample-times from / to pinned host memory, "
                       f"{chunk_frames}-frame time slices, one call per direction"}
    samples = pcm.numel()
    pipe.encode(host_pcm.view(-1), clips, host_out.view(-1), chunk_frames)  # warm-up: staging buffers, chain tables
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); pipe.encode(host_pcm.view(-1), clips, host_out.view(-1), chunk_frames); ts.append(time.perf_counter() - t0)
    same = bool(torch.equal(host_out, out.cpu()))
    rec["encode"] = {"wall_ms": min(ts) * 1e3, "samples_per_s": samples / min(ts),
                     "host_to_device_GBps": pcm.numel() * 2 / min(ts) / 1e9, "bytes_equal_device_path": same}
    pipe.decode(host_out.view(-1), clips, host_back.view(-1), chunk_frames)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); pipe.decode(host_out.view(-1), clips, host_back.view(-1), chunk_frames); ts.append(time.perf_counter() - t0)
    back = torch.empty_like(pcm)
    batch.DecodePlan.from_clips(clips).run(out, back, stream)
    torch.cuda.synchronize()
    k = list(range(0, n_clips, max(1, n_clips // 16)))
    same = bool(torch.equal(host_back[k], back[k].cpu()))
    rec["decode"] = {"wall_ms": min(ts) * 1e3, "samples_per_s": samples / min(ts),
                     "device_to_host_GBps": pcm.numel() * 2 / min(ts) / 1e9, "pcm_equal_device_path": same}
    rec["samples_per_s"] = rec["encode"]["samples_per_s"]
    del pipe, back
    return rec


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def final_line(full: dict) -> dict:
    """The contract's JSON line from the full record: every contract key, `roofline` and `cpu_baseline` as FLAT dicts (the
    decode kernel's fraction and the CPU decode / all-cores figures as scalars beside the encode kernel's), and compact
    `decode`, `configs`, `e2e` records last.  Everything dropped here is on the `bench_extras` line printed before it."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "rccl_ranks", "dist_backend", "gathered_bytes", "gathered_shards_bit_exact_vs_oracle",
            "preflight", "bit_exact_vs_oracle", "speedup_vs_cpu_1_thread", "speedup_vs_cpu_all_cores", "speedup_decode_vs_cpu_1_thread")
    out = {k: full[k] for k in keep if k in full}
    rf = {k: v for k, v in full["roofline"].items() if not isinstance(v, dict)}
    vi = full["roofline"].get("valu_issue") or {}
    rf.update({"valu_issue_frac": _r(vi.get("frac"), 4), "valu_per_slice": vi.get("valu_per_slice"), "step_form": vi.get("step_form"),
               "wavefronts": vi.get("wavefronts"), "helper_wavefronts": vi.get("helper_wavefronts")})
    dec = full.get("decode")
    if dec:
        dr = dec["roofline"]
        rf.update({"decode_kernel": dr["kernel"], "decode_kernel_ms": _r(dec["kernel_ms_avg"]), "decode_achieved": _r(dr["achieved"], 1),
                   "decode_frac": _r(dr["frac"]), "decode_frac_best_placement": _r(dr["frac_best_placement"]),
                   "decode_frac_worst_placement": _r(dr["frac_worst_placement"]), "decode_traffic": dr.get("traffic"),
                   "decode_algorithmic_bytes_per_launch": dr["algorithmic_bytes_per_launch"]})
    k2 = full.get("k2_extras") or {}
    if "mono" in k2:
        rf["decode_frac_mono"] = _r(k2["mono"]["roofline"]["frac"])
    for name, rec in (k2.get("general") or {}).items():
        rf["decode_frac_" + name] = _r(rec["roofline"]["frac"])
    out["roofline"] = rf
    k1 = full.get("k1_extras") or {}
    sc = k1.get("signal_classes") or {}
    if "music" in sc:
        cls = {"workload": sc.get("workload"), "music_ns_per_slice": _r(sc["music"]["ns_per_slice"], 1)}
        for kind in ("white", "silence", "fixture", "loud"):
            if kind in sc:
                cls[kind + "_vs_music"] = _r(sc[kind]["vs_music"])
        if "per_wavefront" in sc.get("loud", {}):
            cls["loud_per_wavefront"] = sc["loud"]["per_wavefront"]
        cls["all_bit_exact_vs_oracle"] = all(sc[k]["bit_exact_vs_oracle"] for k in sc if isinstance(sc[k], dict))
        out["signal_classes"] = cls
    sl = k1.get("small_launch") or {}
    if sl:
        out["small_launch_ns_per_slice"] = {k: _r(v["ns_per_slice"], 1) for k, v in sl.items() if isinstance(v, dict)}
    if "full_chip" in k1:
        out["full_chip"] = {"workload": k1["full_chip"]["workload"], "samples_per_s": _r(k1["full_chip"]["samples_per_s"], 0),
                            "valu_issue_frac": _r((k1["full_chip"]["valu_issue"] or {}).get("frac"), 4),
                            "bit_exact_vs_oracle": k1["full_chip"]["bit_exact_vs_oracle"]}
    if k2:
        shapes = dict(k2.get("general") or {}, **({"mono": k2["mono"]} if "mono" in k2 else {}))
        out["k2_shapes"] = {"kernel_ms": {k: _r(v["kernel_ms_avg"]) for k, v in shapes.items()},
                            "all_bit_exact_vs_oracle": all(v["bit_exact_vs_oracle"] for v in shapes.values()),
                            "header_mismatches": sum(v["header_mismatches"] for v in shapes.values())}
    if "packed_path" in full:
        out["packed_path"] = full["packed_path"]
    if dec:
        dr = dec["roofline"]
        out["decode"] = {"workload": full["config"]["workload"].replace("encode_all per clip", "decode of the streams the encode leg wrote"),
                         "value": dec["value"], "unit": dec["unit"], "kernel_ms_avg": _r(dec["kernel_ms_avg"]),
                         "kernel_ms_per_placement": dec["kernel_ms_per_placement"], "header_mismatches": dec.get("header_mismatches"),
                         "round_trip_rms": _r(dec.get("round_trip_rms"), 2),
                         "roofline": {"bound": "hbm", "achieved": _r(dr["achieved"], 1), "peak": dr["peak"], "unit": "GB/s",
                                      "frac": _r(dr["frac"]), "traffic": dr.get("traffic"), "kernel": dr["kernel"],
                                      "algorithmic_bytes_per_launch": dr["algorithmic_bytes_per_launch"]}}
    cf = full.get("configs") or {}
    if cf:
        out["configs"] = {k: {kk: _r(vv) for kk, vv in v.items() if kk != "workload"} for k, v in cf.items()}
    if "e2e" in full:
        e = full["e2e"]
        out["e2e"] = e if "error" in e else {"workload": e["workload"], "samples_per_s": _r(e["samples_per_s"], 0),
                                             "encode": {k: _r(v) for k, v in e["encode"].items()},
                                             "decode": {k: _r(v) for k, v in e["decode"].items()}}
    if "cpu_baseline" in full:
        out["cpu_baseline"] = {k: _r(v, 1) if k.endswith("value") else _r(v) for k, v in full["cpu_baseline"].items()}
    return out


def preflight(qdist, rank, world, local_rank, dev):
    """Before the timed loop of an N > 1 run: what the first run on a multi-GPU box should establish before anything is
    measured -- how many devices every rank sees, the peer-access matrix (row = this rank's
    device, as hipDeviceCanAccessPeer reports it), and a 1 MB gather to rank 0 through the very function the timed loop
    uses (RCCL when the backend is nccl), checked byte for byte and timed (second call: the first builds the communicator's
    channels).  Failures are reported in the record, they do not end the run."""
    rec = {"backend": dist.get_backend(), "world_size": world}
    try:
        ndev = torch.cuda.device_count()
        row = "".join("1" if (j == local_rank or torch.cuda.can_device_access_peer(local_rank, j)) else "0" for j in range(ndev))
        rows = [None] * world
        dist.all_gather_object(rows, (rank, local_rank, ndev, torch.cuda.get_device_name(local_rank), row))
        rec["device_count_per_rank"] = [r[2] for r in rows]
        rec["devices"] = sorted({r[3] for r in rows})
        rec["peer_access_rows"] = {f"rank{r[0]}:dev{r[1]}": r[4] for r in rows}
        rec["all_peers_accessible"] = all("0" not in r[4] for r in rows)
        x = torch.full((1, 1 << 20), rank + 1, dtype=torch.uint8, device=dev)
        out = torch.zeros((world, 1 << 20), dtype=torch.uint8, device=dev) if rank == 0 else None
        ms = []
        for _ in range(2):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            work, _ = qdist.gather_encoded_async(x, world, dst=0, out=out)
            work.wait()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        rec["gather_1mb_ms"] = [round(v, 3) for v in ms]
        if rank == 0:
            want = torch.arange(1, world + 1, dtype=torch.uint8, device=dev).view(world, 1).expand(world, 1 << 20)
            rec["gather_1mb_ok"] = bool(torch.equal(out, want))
    except Exception as e:
        rec["error"] = f"{type(e).__name__}: {e}"[:300]
    return rec


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (fresh interpreters, one
    per GPU) and wait.  This process makes no GPU call before or after (torch.cuda.device_count() does not initialise
    the device).  Rank 0's JSON line goes straight to our stdout; any failing rank ends the others and fails the run."""
    have = torch.cuda.device_count()
    if os.environ.get("QOA_BENCH_SHARE_GPU") == "1":  # control-flow smoke test on a 1-GPU box (with QOA_DIST_BACKEND=gloo)
        have = n
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr)
                    for q in pending:  # exact PIDs we started
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main_abi_multi(args):
    """`--abi-multi`: the N-GPU step through include/qoaudio_amd.h alone -- ONE process; qoa_multi_encode shards the clips
    over the devices (device i holds its 1024 clips' PCM in its own HBM), every device runs its time-sliced pipeline, and
    the encoded streams arrive in device 0's buffer by peer copies that overlap the following slice's kernel.  Same JSON
    keys as the torch.distributed form; `ms_per_step` is the wall time of the synchronous call."""
    import qoaudio_amd as q
    from qoaudio_amd import batch, synth
    world = args.gpus
    dup = bool(args.duplicate_devices)  # all shards on device 0 (the C ABI's test mode): the N-shard control flow on a 1-GPU box
    if torch.cuda.device_count() < world and not dup:
        print(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        sys.exit(2)
    dev_of = (lambda i: 0) if dup else (lambda i: i)
    n_clips, ch = args.clips, args.channels
    samples = args.frames * 5120
    desc = q.QoaDesc(ch, args.rate, samples)
    per_pcm, per_enc = samples * ch, q.encoded_size(ch, samples)
    pcms = [synth.make_clips(n_clips, samples, ch, args.rate, device=f"cuda:{dev_of(i)}", first_index=i * n_clips) for i in range(world)]
    clips = [(k * per_pcm, (i * n_clips + k) * per_enc, desc) for i in range(world) for k in range(n_clips)]
    out = torch.empty((world * n_clips, per_enc), dtype=torch.uint8, device="cuda:0")
    multi = batch.MultiGpu([dev_of(i) for i in range(world)], allow_duplicate_devices=dup)
    flat = [p.view(-1) for p in pcms]
    for _ in range(args.warmup):
        multi.encode(flat, clips, out.view(-1), args.chunk_frames)
    for i in range(world):
        torch.cuda.synchronize(dev_of(i))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        multi.encode(flat, clips, out.view(-1), args.chunk_frames)
    for i in range(world):
        torch.cuda.synchronize(dev_of(i))
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    total_samples = world * n_clips * samples * ch
    ms = elapsed_ms / max(args.steps, 1)
    algo_bytes = n_clips * (per_pcm * 2 + per_enc)  # per device
    from oracle import oracle as O
    ok = True
    for r in (0, world - 1):  # first clip of the first and of the last shard, as they arrived in device 0's buffer
        want = O.encode_all(pcms[r][0].cpu().numpy().reshape(-1), ch, args.rate, samples)
        ok = ok and out[r * n_clips].cpu().numpy().tobytes() == want
    result = {
        "metric": "encoded samples/sec (stereo 44.1 kHz)", "value": total_samples * args.steps / (elapsed_ms * 1e-3),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"{n_clips} clips/GPU x {ch} ch x {args.rate} Hz x {args.frames} frames ({samples} sample-times), "
                               f"qoa_multi_encode: one process, {args.chunk_frames}-frame time slices, encoded streams gathered "
                               f"on device 0 by peer copies",
                   "clips_per_gpu": n_clips, "channels": ch, "sample_rate": args.rate, "frames_per_clip": args.frames,
                   "samples_per_step": total_samples, "pcm_bytes_per_gpu": n_clips * per_pcm * 2,
                   "encoded_bytes_per_gpu": n_clips * per_enc, "parallelism": f"clips sharded over {world} GPU(s), C ABI"},
        "roofline": {"bound": "hbm", "achieved": algo_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": algo_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "qoa_encode_kernel",
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "per device, over the wall time of the whole pipelined call (kernels of all time slices + "
                             "staging copies), not a kernel duration"},
        "multi_gpu_path": "c-abi" + (" (ALL SHARDS ON DEVICE 0: control-flow test mode, not a scaling figure)" if dup else ""), "gathered_bytes": int(out.numel()), "gathered_shards_bit_exact_vs_oracle": bool(ok),
        "bit_exact_vs_oracle": bool(ok),
    }
    print(json.dumps(result))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=1024, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=256, help="frames (5120 sample-times) per clip")
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the 1-thread CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the K1 records outside the headline shape")
    ap.add_argument("--preflight", action="store_true", help="N > 1: device count, peer-access matrix and a 1 MB gather "
                                                             "before the timed loop (the default for N > 1; see --no-preflight)")
    ap.add_argument("--no-preflight", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (pinned host memory) record")
    ap.add_argument("--extras-file", default="", help="also write the full record (the bench_extras line) to this file")
    ap.add_argument("--abi-multi", action="store_true",
                    help="N > 1 through the C ABI's one-process form (qoa_multi_*: one host thread + pipeline per device, "
                         "peer copies to device 0) instead of one torch.distributed rank per GPU")
    ap.add_argument("--chunk-frames", type=int, default=8, help="--abi-multi: frames per time slice of the device pipelines")
    ap.add_argument("--duplicate-devices", action="store_true",
                    help="--abi-multi: put every shard on device 0 (QOA_MULTI_ALLOW_DUPLICATE_DEVICES): times the N-shard control "
                         "flow, staging and copy path on a 1-GPU box; NOT a scaling measurement")
    args = ap.parse_args()

    if args.abi_multi:
        return main_abi_multi(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import qoaudio_amd as q
    from qoaudio_amd import batch, synth
    from qoaudio_amd import dist as qdist

    rank, world, local_rank = qdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("QOA_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pre = preflight(qdist, rank, world, local_rank, dev) if world > 1 and not args.no_preflight else None

    n_clips, ch = args.clips, args.channels
    samples = args.frames * 5120
    desc = q.QoaDesc(ch, args.rate, samples)
    clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
    # synthetic PCM generated once, resident in HBM before anything is timed
    pcm = synth.make_clips(n_clips, samples, ch, args.rate, device=dev, first_index=rank * n_clips)
    # two output buffers: for N > 1 the gather of step i runs on the communication stream while step i+1 encodes
    outs = [torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    out = outs[0]
    plan = batch.EncodePlan(clips)
    stream = torch.cuda.current_stream()
    gather_dst = [torch.empty((n_clips * world, per_enc), dtype=torch.uint8, device=dev) if (world > 1 and rank == 0)
                  else None for _ in outs]
    pending = [None] * len(outs)

    def step(i, events=None):
        b = i % len(outs)
        if pending[b] is not None:  # the gather that last read outs[b] must be done before it is overwritten
            pending[b].wait()
            pending[b] = None
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        plan.run(pcm, outs[b], stream)
        if events is not None:
            e1.record(stream)
            events.append((e0, e1))
        if world > 1:
            pending[b], _ = qdist.gather_encoded_async(outs[b], n_clips * world, dst=0, out=gather_dst[b])

    def drain():
        for b in range(len(outs)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    events = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    elapsed_ms = qdist.barrier_max_ms(elapsed_ms, dev)
    gathered = gather_dst[(args.steps - 1) % len(outs)] if args.steps else None
    kernel_ms = [a.elapsed_time(b) for a, b in events]

    total_samples = plan.samples * world
    value = total_samples * args.steps / (elapsed_ms * 1e-3)
    k_avg_ms = float(np.mean(kernel_ms))
    algo_bytes = plan.pcm_bytes + plan.out_bytes  # per launch: PCM read + QOA written (2.4039 B/sample)
    achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
    enc_traffic, enc_traffic_note = pmc_traffic("encode", n_clips, ch, args.frames)

    result = {
        "metric": "encoded samples/sec (stereo 44.1 kHz)", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_ms / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"{n_clips} clips/GPU x {ch} ch x {args.rate} Hz x {args.frames} frames "
                               f"({samples} sample-times), encode_all per clip"
                               + (", RCCL gather of encoded buffers to rank 0" if world > 1 else ""),
                   "clips_per_gpu": n_clips, "channels": ch, "sample_rate": args.rate, "frames_per_clip": args.frames,
                   "samples_per_step": total_samples, "pcm_bytes_per_gpu": plan.pcm_bytes,
                   "encoded_bytes_per_gpu": plan.out_bytes, "parallelism": f"clips sharded over {world} GPU(s)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": enc_traffic, "kernel": "qoa_encode_kernel",
                     "kernel_ms_avg": k_avg_ms, "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "integer-VALU-issue bound (serial LMS chain per channel), see DESIGN.md",
                     # the bound that actually governs K1 (profiles/r02_pmc_summary.txt): 528 VALU wave-instructions
                     # per slice per wavefront, one per 4 cycles per SIMD at 2.4 GHz on 1024 SIMDs
                     "valu_issue": valu_issue(plan, samples, k_avg_ms)},
    }
    if enc_traffic_note:
        result["roofline"]["traffic_note"] = enc_traffic_note
    if pre is not None:
        result["preflight"] = pre
    if world > 1:
        result["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
        result["dist_backend"] = dist.get_backend()

    if rank == 0:
        # bit-exactness of this very run against the oracle, on a few clips (checker only, untimed)
        from oracle import oracle as O
        hp = pcm[:2].cpu().numpy()
        he = out[:2].cpu().numpy()
        ok = all(he[i].tobytes() == O.encode_all(hp[i].reshape(-1), ch, args.rate, samples) for i in range(2))
        result["bit_exact_vs_oracle"] = bool(ok)
        # K1's fast paths: slices handed to a slower form.  The timed plan carries no counters (no atomics in its kernel);
        # the same batch runs once more, untimed, through a plan that counts.
        cplan = batch.EncodePlan(clips, batch.PLAN_COUNTERS)
        cplan.run(pcm, out, stream)
        declined, unpacked, misses, cheap_declined, marked = cplan.counters()[:5]
        result["packed_path"] = {"declined_slices": declined, "unpacked_slices": unpacked, "speculation_misses": misses,
                                 "cheap_quotient_declines": cheap_declined, "cheap_quotient_marked_candidates": marked,
                                 "slices_total": n_clips * ch * ((samples + 19) // 20), "launches": 1}
        del cplan
        if world > 1 and gathered is not None:
            # the gather itself: the first clip of the first and of the last rank's shard, as they arrived on rank 0,
            # against the oracle on that clip's PCM (regenerated here: clip indices are global)
            result["gathered_bytes"] = int(gathered.numel())
            g_ok = True
            for r in (0, world - 1):
                one = synth.make_clips(1, samples, ch, args.rate, device=dev, first_index=r * n_clips)
                want = O.encode_all(one[0].cpu().numpy().reshape(-1), ch, args.rate, samples)
                g_ok = g_ok and gathered[r * n_clips].cpu().numpy().tobytes() == want
            result["gathered_shards_bit_exact_vs_oracle"] = bool(g_ok)

    if not args.no_decode:
        dplan = batch.DecodePlan.from_clips(clips)
        # 4 separately allocated (input, output) pairs, each: untimed warm-up (the clocks keep rising for ~40 ms after a
        # host-side pause), then launches enqueued back to back like the encode steps, one HIP event pair per launch
        per_pair, back = decode_over_placements(q, batch, dplan, out, pcm, stream, pairs=4 if world == 1 else 1,
                                                warm=24 if world > 1 else 12, reps=max(10, args.steps))
        dec_traffic, dec_traffic_note = pmc_traffic("decode", n_clips, ch, args.frames)
        # K2u takes the full frames of every channel count 1 .. 8 (round 4)
        result["decode"] = decode_record(dplan, per_pair, "qoa_decode_uniform_kernel" if 1 <= ch <= 8 else "qoa_decode_kernel",
                                         dec_traffic, dec_traffic_note)
        result["decode"].update({"header_mismatches": dplan.mismatches(),
                                 "round_trip_rms": float((back[:8].float() - pcm[:8].float()).pow(2).mean().sqrt())})
        del back

    if rank == 0 and world == 1 and not args.no_extras and args.frames % 8 == 0:
        # K1 outside the headline shape (lab records, not part of `value`): the same 2048 chains at 2 chains per
        # wavefront (1024 wavefronts, one per SIMD), and a batch with a full wavefront for every SIMD several times over
        # (8192 clips x 32 frames: the same 5.4 GB of PCM, 16384 chains = 4096 wavefronts, 4 per SIMD)
        extras = {}
        p2 = batch.EncodePlan(clips, batch.PLAN_CPW_2)
        p2.run(pcm, out, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); p2.run(pcm, out, stream); e1.record(stream)
        torch.cuda.synchronize()
        extras["config3_at_2_chains_per_wavefront_kernel_ms"] = e0.elapsed_time(e1)
        del p2
        p3 = batch.EncodePlan(clips, batch.PLAN_NO_HELPER)  # the classic kernel: every wavefront does its own block tops
        p3.run(pcm, out, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); p3.run(pcm, out, stream); e1.record(stream)
        torch.cuda.synchronize()
        extras["config3_without_helper_wavefronts_kernel_ms"] = e0.elapsed_time(e1)
        del p3
        plan.run(pcm, out, stream)  # `out` again holds the default plan's bytes (identical by construction; decode leg reads it)
        fc_clips, fc_frames = 8 * n_clips, max(1, args.frames // 8)
        fc_samples = fc_frames * 5120
        fdesc = q.QoaDesc(ch, args.rate, fc_samples)
        fclips, _, fper_enc = batch.uniform_clips(fc_clips, fdesc)
        fpcm = pcm.view(fc_clips, fc_samples, ch)  # the same PCM bytes cut into 8x as many, 8x shorter clips
        fout = torch.empty((fc_clips, fper_enc), dtype=torch.uint8, device=dev)
        fplan = batch.EncodePlan(fclips)
        fplan.run(fpcm, fout, stream)
        torch.cuda.synchronize()
        fms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); fplan.run(fpcm, fout, stream); e1.record(stream)
            torch.cuda.synchronize()
            fms.append(e0.elapsed_time(e1))
        f_ms = float(np.mean(fms))
        from oracle import oracle as O
        f_ok = fout[fc_clips - 1].cpu().numpy().tobytes() == O.encode_all(fpcm[fc_clips - 1].cpu().numpy().reshape(-1), ch,
                                                                          args.rate, fc_samples)
        extras["full_chip"] = {"workload": f"{fc_clips} clips x {ch} ch x {fc_frames} frames (4 wavefronts per SIMD)",
                               "kernel_ms_avg": f_ms, "samples_per_s": fplan.samples / (f_ms * 1e-3),
                               "valu_issue": valu_issue(fplan, fc_samples, f_ms), "bit_exact_vs_oracle": bool(f_ok)}
        del fplan, fout
        torch.cuda.synchronize()
        extras["signal_classes"] = signal_classes(q, batch, synth, dev, stream, ch, args.rate)
        extras["small_launch"] = small_launches(q, batch, synth, dev, stream, args.rate)
        result["k1_extras"] = extras
        result["configs"] = other_configs(q, batch, synth, dev, stream, pcm, args.rate, n_clips, args.frames)
        if not args.no_decode:
            # K2u on config 2's shape (1024 mono clips x 256 frames): half the samples of config 3, the same launch geometry
            mdesc = q.QoaDesc(1, args.rate, samples)
            mclips, _, mper_enc = batch.uniform_clips(n_clips, mdesc)
            mpcm = pcm[:, :, 0].contiguous().view(n_clips, samples, 1)
            menc = torch.empty((n_clips, mper_enc), dtype=torch.uint8, device=dev)
            batch.EncodePlan(mclips).run(mpcm, menc, stream)
            mplan = batch.DecodePlan.from_clips(mclips)
            m_pairs, mback = decode_over_placements(q, batch, mplan, menc, mpcm, stream, pairs=4, warm=12, reps=10)
            from oracle import oracle as O
            m_ok = np.array_equal(mback[7].cpu().numpy().reshape(-1), O.decode_all(menc[7].cpu().numpy().tobytes())[2])
            mono = decode_record(mplan, m_pairs, "qoa_decode_uniform_kernel")
            mono.update({"workload": f"{n_clips} clips x 1 ch x {args.frames} frames, decode", "header_mismatches": mplan.mismatches(),
                         "bit_exact_vs_oracle": bool(m_ok)})
            result["k2_extras"] = {"mono": mono}
            del mplan, mback, menc, mpcm
            result["k2_extras"]["general"] = general_decode(q, batch, synth, dev, stream, args.rate)

    if rank == 0 and world == 1 and not args.no_e2e and not args.no_extras:
        try:
            result["e2e"] = e2e_record(q, batch, dev, stream, pcm, out, clips)
        except Exception as e:  # (not enough pinnable host memory, ...): the GPU record above must survive
            result["e2e"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            # only the clips the budget can reach (a server core encodes 15-30 M samples/s), at least 4
            sample_clips = min(n_clips, max(4, int(args.cpu_seconds * 30e6 / (samples * ch)) + 2))
            host = pcm[:sample_clips].cpu().numpy()
            cpu = cpu_baseline_child(host, args.rate, args.cpu_seconds)
            one, n1 = cpu["one_thread"], cpu["one_thread"]["clips"]
            he = out[:n1].cpu().numpy()
            same = all(hashlib.sha256(he[i].tobytes()).hexdigest() == cpu["sha256"][i] for i in range(n1))
            cpus = cpu.get("cpus", {})
            result["cpu_baseline"] = {"value": one["value"], "unit": "samples/s", "cores": 1, "kind": "port",
                                      "sample": f"{n1} of the same {n_clips} clips, {one['seconds']:.1f} s, oracle ref_style "
                                                f"encoder, gcc -O3 -flto generic x86-64, 1 thread",
                                      "sample_clips": n1, "sample_samples": one["samples"], "sample_seconds": one["seconds"],
                                      "gpu_bytes_equal_cpu_bytes": bool(same),
                                      "cores_visible": cpus.get("visible"), "cores_in_affinity_mask": cpus.get("affinity"),
                                      "cgroup_cpu_quota": cpus.get("cgroup_quota"), "cores_usable": cpus.get("usable"),
                                      "threads_per_core": cpus.get("threads_per_core")}
            cb = result["cpu_baseline"]
            if "decode_one_thread" in cpu and not args.no_decode:
                # the CPU's decode_all beside K2u's: the same streams (the encoder legs' bytes are equal), PCM sha256-compared
                dd, nd = cpu["decode_one_thread"], cpu["decode_one_thread"]["clips"]
                back = torch.empty((nd,) + tuple(pcm.shape[1:]), dtype=torch.int16, device=dev)
                batch.DecodePlan.from_clips(clips[:nd]).run(out, back, stream)
                torch.cuda.synchronize()
                hb = back.cpu().numpy()
                same_pcm = all(hashlib.sha256(hb[i].tobytes()).hexdigest() == cpu["decode_sha256"][i] for i in range(nd))
                cb.update({"decode_value": dd["value"], "decode_unit": "samples/s", "decode_cores": 1,
                           "decode_sample_clips": nd, "decode_sample_seconds": dd["seconds"],
                           "decode_sample": f"oracle decode_all (the reference's item-by-item iterator drain) on {nd} of the "
                                            f"streams above, {dd['seconds']:.1f} s, 1 thread",
                           "gpu_pcm_equals_cpu_pcm": bool(same_pcm)})
                if "decode" in result:
                    result["speedup_decode_vs_cpu_1_thread"] = result["decode"]["value"] / dd["value"]
                del back
            mn = cpu.get("march_native", {})
            if "one_thread" in mn:
                cb["march_native_value"] = mn["one_thread"]["value"]
                cb["march_native_bytes_equal_generic_build"] = mn["sha256"] == cpu["sha256"][:len(mn["sha256"])]
                if "decode_one_thread" in mn:
                    cb["march_native_decode_value"] = mn["decode_one_thread"]["value"]
            elif mn:
                cb["march_native_error"] = str(mn.get("error"))[:120]
            if "all_cores" in cpu:
                ac = cpu["all_cores"]
                cb.update({"all_cores_value": ac["value"], "all_cores_workers": ac["cores"], "all_cores_clips": ac["clips"],
                           "all_cores_seconds": ac["seconds"], "scaling_efficiency": ac["scaling_efficiency"],
                           "scaling_efficiency_note": "all_cores_value / (value x all_cores_workers); workers = CPUs in the "
                                                      "affinity mask capped by the cgroup quota, one process each, every one "
                                                      "encoding whole clips for the same >= 2 s window"})
                result["speedup_vs_cpu_all_cores"] = value / ac["value"]
            result["speedup_vs_cpu_1_thread"] = value / one["value"]
        except Exception as e:  # the GPU record above must not be lost to a failing CPU leg (full /dev/shm, ...)
            result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        # Two lines.  The lab records (per-class counters, per-placement times, small launches, ...) go out FIRST, on a line of
        # their own; the LAST line is the contract's record, kept under 6 KB so that whoever keeps only the tail of this
        # process's output still holds config 3's decode roofline, the other configs, the end-to-end figures and the CPU legs.
        print(json.dumps({"bench_extras": result}))
        final = final_line(result)
        if args.extras_file:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.extras_file)), exist_ok=True)
                with open(args.extras_file, "w") as f:
                    json.dump(result, f)
            except OSError:
                pass
        print(json.dumps(final))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
