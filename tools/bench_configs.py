"""The BASELINE.json configs that are not bench.py's headline line (1, 2, 5), measured for the record.
Run on the GPU box: python tools/bench_configs.py > gpurun_out/configs.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import qoaudio_amd as q
from qoaudio_amd import batch, synth
from oracle import oracle as O

out = {}
dev = torch.device("cuda:0")

# config 1: fixture -> PCM -> QOA through the one-clip host-pointer API (PCIe + launch + sync included)
fx = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "julien_baker_sprained_ankle.qoa"), "rb").read()
q.decode_all(fx)  # warm up
t = time.perf_counter(); d = q.decode_all(fx); t_dec = time.perf_counter() - t
desc = q.QoaDesc(2, 44100, d.samples.size // 2)
q.encode_all(d.samples, desc)
t = time.perf_counter(); enc = q.encode_all(d.samples, desc); t_enc = time.perf_counter() - t
t = time.perf_counter(); ref = O.encode_all(d.samples, 2, 44100, d.samples.size // 2); t_cpu = time.perf_counter() - t
t = time.perf_counter(); O.decode_all(fx); t_cpu_dec = time.perf_counter() - t
out["config1_fixture"] = {"samples": int(d.samples.size), "gpu_encode_all_ms": t_enc * 1e3, "gpu_decode_all_ms": t_dec * 1e3,
                          "cpu_oracle_encode_ms": t_cpu * 1e3, "cpu_oracle_decode_ms": t_cpu_dec * 1e3,
                          "bit_exact": enc == ref, "fixture_prefix_bytes_equal": enc[:1073768] == fx[:1073768],
                          "note": "2 serial chains only: a single clip cannot use the GPU (SURVEY 7, hard part 6)"}

# config 2: 1024 mono clips x 256 frames, encode only
n_clips, frames = 1024, 256
samples = frames * 5120
pcm = synth.make_clips(n_clips, samples, 1, device=dev)
desc = q.QoaDesc(1, 44100, samples)
clips, _, per_enc = batch.uniform_clips(n_clips, desc)
plan = batch.EncodePlan(clips)
o = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
plan.run(pcm, o); torch.cuda.synchronize()
ms = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(pcm, o); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
hp = pcm[3].cpu().numpy().reshape(-1)
out["config2_mono_1024x256"] = {"samples": plan.samples, "kernel_ms": float(np.mean(ms)), "samples_per_s": plan.samples / (np.mean(ms) * 1e-3),
                                "algorithmic_GBps": (plan.pcm_bytes + plan.out_bytes) / (np.mean(ms) * 1e-3) / 1e9,
                                "bit_exact_clip3": o[3].cpu().numpy().tobytes() == O.encode_all(hp, 1, 44100, samples)}
del pcm, o, plan

# config 5: 8 channels, 48 kHz, 600 s, frame-at-a-time streaming API (host pointers in, host bytes out, every call)
ch, rate, frames = 8, 48000, 5625
x = synth.make_clips(1, frames * 5120, ch, rate, device=dev)[0].cpu().numpy().reshape(-1)
enc = q.QoaEncoder(q.QoaDesc(ch, rate, frames * 5120))
stream = bytearray(); enc.write_header(stream)
t = time.perf_counter()
for f in range(frames):
    enc.encode_frame(x[f * 5120 * ch:(f + 1) * 5120 * ch], stream)
dt = time.perf_counter() - t
e = O.OracleEncoder(ch, rate, frames * 5120)
ref = bytearray(e.write_header())
t = time.perf_counter()
for f in range(200):
    ref += e.encode_frame(x[f * 5120 * ch:(f + 1) * 5120 * ch])[0]
dt_cpu = (time.perf_counter() - t) * frames / 200
out["config5_streaming_8ch_48k_600s"] = {"samples": frames * 5120 * ch, "calls": frames, "wall_s": dt, "samples_per_s": frames * 5120 * ch / dt,
                                         "ms_per_encode_frame_call": dt / frames * 1e3, "x_realtime": 600.0 / dt,
                                         "cpu_oracle_1_thread_s_extrapolated": dt_cpu,
                                         "first_200_frames_bit_exact": bytes(stream[:len(ref)]) == bytes(ref), "bytes": len(stream)}
# config 3 end to end from pinned host memory (what a host-buffer boundary would pay; never bench.py's `value`)
n_clips, frames, ch = 1024, 256, 2
samples = frames * 5120
desc = q.QoaDesc(ch, 44100, samples)
clips, per_pcm, per_enc = batch.uniform_clips(n_clips, desc)
host_pcm = torch.empty((n_clips, samples, ch), dtype=torch.int16).pin_memory()
host_pcm.copy_(synth.make_clips(64, samples, ch, device=dev).repeat((16, 1, 1)).cpu())
host_out = torch.empty((n_clips, per_enc), dtype=torch.uint8).pin_memory()
d_pcm = torch.empty_like(host_pcm, device=dev); d_out = torch.empty((n_clips, per_enc), dtype=torch.uint8, device=dev)
plan = batch.EncodePlan(clips)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_pcm.copy_(host_pcm, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    plan.run(d_pcm, d_out); torch.cuda.synchronize(); t2 = time.perf_counter()
    host_out.copy_(d_out, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
out["config3_pcie_inclusive_serial"] = {"h2d_ms": (t1 - t0) * 1e3, "h2d_GBps": plan.pcm_bytes / (t1 - t0) / 1e9, "kernel_ms": (t2 - t1) * 1e3,
                                        "d2h_ms": (t3 - t2) * 1e3, "d2h_GBps": plan.out_bytes / (t3 - t2) / 1e9,
                                        "samples_per_s_end_to_end": plan.samples / (t3 - t0)}
# signal classes: the GPU kernel's cost is content-independent, the CPU's early exits are not
cls = {}
n_c, fr = 256, 64
for kind in ("music", "white", "silence", "fixture"):
    if kind == "fixture":
        base = torch.from_numpy(d.samples[: fr * 5120 * 2].reshape(1, fr * 5120, 2).copy()).to(dev)
        x = base.repeat((n_c, 1, 1)).contiguous()
    else:
        x = synth.make_clips(16, fr * 5120, 2, device=dev, kind=kind).repeat((n_c // 16, 1, 1)).contiguous()
    dsc = q.QoaDesc(2, 44100, fr * 5120)
    cl, _, pe = batch.uniform_clips(n_c, dsc)
    pl = batch.EncodePlan(cl, batch.PLAN_CPW_4)
    oo = torch.empty((n_c, pe), dtype=torch.uint8, device=dev)
    pl.run(x, oo); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); pl.run(x, oo); e1.record(); torch.cuda.synchronize()
    hx = x[0].cpu().numpy().reshape(-1)
    enc1 = O.OracleEncoder(2, 44100, fr * 5120)
    t0 = time.perf_counter(); ref = enc1.encode(hx); dtc = time.perf_counter() - t0
    cls[kind] = {"gpu_kernel_ms_256clips_x_64frames": e0.elapsed_time(e1), "cpu_1thread_samples_per_s": hx.size / dtc,
                 "cpu_candidate_steps_per_slice": enc1.candidate_steps / enc1.slices,
                 "bit_exact": oo[0].cpu().numpy().tobytes() == ref}
out["signal_classes"] = cls

# the same from host memory through the time-sliced pipeline (qoa_encode_batch_host)
host_out.zero_()
for cf in (16, 8):
    batch.encode_batch_host(host_pcm.view(-1), clips, host_out.view(-1), chunk_frames=cf)
    t0 = time.perf_counter(); batch.encode_batch_host(host_pcm.view(-1), clips, host_out.view(-1), chunk_frames=cf); dt = time.perf_counter() - t0
    out[f"config3_host_pipeline_chunk{cf}"] = {"wall_ms": dt * 1e3, "samples_per_s_end_to_end": plan.samples / dt,
                                               "bytes_equal_device_path": bool(torch.equal(host_out, d_out.cpu()))}
# persistent handle (qoa_host_pipeline_*): repeated batches without the stream / event / hipMalloc set-up, and the decode
# half (PCIe-bound by construction: 2 bytes per sample back to the host)
pipe = batch.HostPipeline()
host_back = torch.empty((n_clips, samples, ch), dtype=torch.int16).pin_memory()
pipe.encode(host_pcm.view(-1), clips, host_out.view(-1), chunk_frames=8)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); pipe.encode(host_pcm.view(-1), clips, host_out.view(-1), chunk_frames=8); ts.append(time.perf_counter() - t0)
out["config3_host_pipeline_persistent_encode"] = {"wall_ms": [t * 1e3 for t in ts], "samples_per_s_end_to_end": plan.samples / min(ts),
                                                  "bytes_equal_device_path": bool(torch.equal(host_out, d_out.cpu()))}
pipe.decode(host_out.view(-1), clips, host_back.view(-1), chunk_frames=8)
ts = []
for cf in (8, 8, 16):
    t0 = time.perf_counter(); pipe.decode(host_out.view(-1), clips, host_back.view(-1), chunk_frames=cf); ts.append(time.perf_counter() - t0)
dplan = batch.DecodePlan.from_clips(clips)
d_back = torch.empty_like(d_pcm)
dplan.run(d_out, d_back); torch.cuda.synchronize()
t0 = time.perf_counter(); host_back2 = d_back.cpu(); t_d2h = time.perf_counter() - t0
out["config3_host_pipeline_decode"] = {"wall_ms_chunk8_8_16": [t * 1e3 for t in ts], "samples_per_s_end_to_end": plan.samples / min(ts),
                                       "pcm_GBps_to_host": plan.pcm_bytes / min(ts) / 1e9,
                                       "equals_device_path": bool(torch.equal(host_back, host_back2)),
                                       "plain_d2h_copy_ms_pageable_dst": t_d2h * 1e3}
print(json.dumps(out, indent=1))
