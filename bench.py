"""bench.py -- encoded samples/s of the QOA encode hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: every rank encodes 1024 synthetic stereo 44.1 kHz
clips of 256 frames (BASELINE config 3's shape; 2,684,354,560 int16 samples, 5.37 GB of PCM resident in
HBM before the timed region) and, for N > 1, the encoded buffers (1.08 GB per rank) are gathered to
rank 0 with RCCL (config 4).  value = samples all ranks encoded / time of the K steps (max over ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(kernel: str, clips: int, channels: int, frames: int):
    """HBM bytes per launch from the committed rocprofv3 PMC profile (profiles/pmc_traffic.json: separate --pmc
    passes, FETCH_SIZE doubled per the gfx950 correction).  PMC counters cannot be read from inside this process,
    so the number is only reported for the exact workload it was collected on; otherwise null."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    if (clips, channels, frames) != (1024, 2, 256):
        return None
    return d[kernel]["traffic_bytes_per_launch"]


def valu_issue(n_chains: int, samples: int, kernel_ms: float):
    """Secondary roofline for K1: VALU wave-instructions issued per second against the chip's issue peak.
    Instruction count per slice is the PMC-measured 660 (SQ_INSTS_VALU / slices / waves, config 3, CPW = 4,
    profiles/r01i_pmc_summary.txt); a lone wavefront per SIMD can issue at most one instruction per 4 cycles."""
    waves = (n_chains + 3) // 4
    slices = (samples + 19) // 20
    achieved = 660.0 * slices * waves / (kernel_ms * 1e-3)
    peak = 1024 * 2.4e9 / 4.0
    return {"bound": "valu-issue", "achieved": achieved, "peak": peak, "unit": "wave-instructions/s",
            "frac": achieved / peak, "wavefronts": waves}


def cpu_baseline(pcm_host: np.ndarray, channels: int, rate: int, budget_s: float):
    """The CPU oracle's ref_style encoder (the reference's algorithm incl. its rank-sorted early exits),
    one thread, on a bounded sample of the same clips.  kind = "port": the Rust reference cannot be built here."""
    from oracle import oracle as O
    O.lib()
    n = 0
    samples = pcm_host.shape[1]
    t0 = time.perf_counter()
    outs = []
    for i in range(pcm_host.shape[0]):
        outs.append(O.encode_all(pcm_host[i].reshape(-1), channels, rate, samples, O.REF_STYLE))
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return n * samples * channels / dt, n, dt, outs


def _cpu_worker(args):
    pcm, channels, rate = args
    from oracle import oracle as O
    return len(O.encode_all(pcm.reshape(-1), channels, rate, pcm.shape[0], O.REF_STYLE))


def cpu_all_cores(pcm_host: np.ndarray, channels: int, rate: int):
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    clips = [pcm_host[i % pcm_host.shape[0]] for i in range(max(cores, min(4 * cores, 64)))]
    with mp.get_context("fork").Pool(cores) as pool:
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(c, channels, rate) for c in clips])
        dt = time.perf_counter() - t0
    return len(clips) * pcm_host.shape[1] * channels / dt, cores, len(clips), dt


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
    args = ap.parse_args()

    import qoaudio_amd as q
    from qoaudio_amd import batch, synth
    from qoaudio_amd import dist as qdist

    rank, world, local_rank = qdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

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
                     "traffic": pmc_traffic("encode", n_clips, ch, args.frames), "kernel": "qoa_encode_kernel",
                     "kernel_ms_avg": k_avg_ms, "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "integer-VALU-issue bound (serial LMS chain per channel), see DESIGN.md",
                     # the bound that actually governs K1 (profiles/r01i_pmc_summary.txt): 660 VALU wave-instructions
                     # per slice per wavefront, one per 4 cycles per SIMD at 2.4 GHz on 1024 SIMDs
                     "valu_issue": valu_issue(n_clips * ch, samples, k_avg_ms)},
    }

    if rank == 0:
        # bit-exactness of this very run against the oracle, on a few clips (checker only, untimed)
        from oracle import oracle as O
        hp = pcm[:2].cpu().numpy()
        he = out[:2].cpu().numpy()
        ok = all(he[i].tobytes() == O.encode_all(hp[i].reshape(-1), ch, args.rate, samples) for i in range(2))
        result["bit_exact_vs_oracle"] = bool(ok)
        declined, unpacked, misses = plan.stats()  # K1's fast paths: slices handed to a slower form, all launches
        result["packed_path"] = {"declined_slices": declined, "unpacked_slices": unpacked, "speculation_misses": misses,
                                 "slices_total": (args.warmup + args.steps) * n_clips * ch * ((samples + 19) // 20)}
        if world > 1 and gathered is not None:
            result["gathered_bytes"] = int(gathered.numel())

    if not args.no_decode:
        dplan = batch.DecodePlan.from_clips(clips)
        back = torch.empty_like(pcm)
        dplan.run(out, back, stream)
        torch.cuda.synchronize()
        dev_ms = []
        for _ in range(max(3, args.steps)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dplan.run(out, back, stream)
            e1.record(stream)
            torch.cuda.synchronize()
            dev_ms.append(e0.elapsed_time(e1))
        d_ms = float(np.mean(dev_ms))
        d_bytes = dplan.in_bytes + dplan.pcm_bytes
        result["decode"] = {"value": dplan.samples / (d_ms * 1e-3), "unit": "samples/s", "kernel_ms_avg": d_ms,
                            "roofline": {"bound": "hbm", "achieved": d_bytes / (d_ms * 1e-3) / 1e9,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": d_bytes / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "traffic": pmc_traffic("decode", n_clips, ch, args.frames),
                                         "kernel": "qoa_decode_frames_kernel",
                                         "algorithmic_bytes_per_launch": d_bytes},
                            "header_mismatches": dplan.mismatches(),
                            "round_trip_rms": float((back[:8].float() - pcm[:8].float()).pow(2).mean().sqrt())}

    if rank == 0 and world == 1 and not args.no_cpu:
        sample_clips = min(n_clips, 96)
        host = pcm[:sample_clips].cpu().numpy()
        v1, n1, dt1, outs = cpu_baseline(host, ch, args.rate, args.cpu_seconds)
        he = out[:n1].cpu().numpy()
        same = all(he[i].tobytes() == outs[i] for i in range(n1))
        result["cpu_baseline"] = {"value": v1, "unit": "samples/s", "cores": 1, "kind": "port",
                                  "sample": f"{n1} of the same {n_clips} clips ({n1 * samples * ch} samples, "
                                            f"{dt1:.1f} s), oracle ref_style encoder, gcc -O3 -flto, 1 thread",
                                  "gpu_bytes_equal_cpu_bytes": bool(same)}
        va, cores, nc, dta = cpu_all_cores(host, ch, args.rate)
        result["cpu_baseline"]["all_cores"] = {"value": va, "cores": cores,
                                               "sample": f"{nc} clips, one clip per process, {dta:.1f} s"}
        result["speedup_vs_cpu_1_thread"] = value / v1

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
