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


KERNEL_SOURCES = {"encode": ("qoa_kernels.hip", "qoa_slice_asm.inc", "qoa_device.h"),
                  "decode": ("qoa_decode.hip", "qoa_device.h")}


def kernel_source_sha256(kernel: str) -> str:
    """sha256 over the source files a kernel is built from: ties a committed PMC profile to the build it was taken on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel]:
        with open(os.path.join(ROOT, "qoaudio_amd", "csrc", name), "rb") as f:
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

K1_VALU_PER_SLICE = 528.0  # PMC: SQ_INSTS_VALU / (wavefronts x slices) on config 3, profiles/r02_pmc_summary.txt


def valu_issue(n_chains: int, samples: int, kernel_ms: float, per_slice: float = K1_VALU_PER_SLICE):
    """Secondary roofline for K1: VALU wave-instructions issued per second against the chip's issue peak.
    Instruction count per slice is PMC-measured (SQ_INSTS_VALU / slices / waves): 528 for the cheap-quotient register form
    (config 3, profiles/r02_pmc_summary.txt; 517 in the block loop, the rest are the block tops; 660 at the start of round
    2); the LDS-table form that launches with >= 8 wavefronts per SIMD use has 451 (+ 26 LDS) in its loop.  A SIMD issues at
    most one instruction per 4 cycles whatever the wavefront count (profiles/r02_ubench_gfx950_issue_classes.txt)."""
    waves = (n_chains + 3) // 4
    slices = (samples + 19) // 20
    achieved = per_slice * slices * waves / (kernel_ms * 1e-3)
    peak = 1024 * 2.4e9 / 4.0
    return {"bound": "valu-issue", "achieved": achieved, "peak": peak, "unit": "wave-instructions/s",
            "frac": achieved / peak, "wavefronts": waves, "valu_per_slice": per_slice}


def cpu_baseline_child(pcm_host: np.ndarray, rate: int, budget_s: float):
    """The CPU oracle's ref_style encoder (the reference's algorithm incl. its rank-sorted early exits) on a bounded
    sample of the same clips: one thread, then all host cores.  Runs in a child process (oracle/cpu_bench.py) that
    never touches the GPU, so its worker pool forks from a clean process.  kind = "port": the Rust reference cannot
    be built here."""
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd, path = tempfile.mkstemp(prefix="qoa_bench_", suffix=".npy", dir=shm)
    os.close(fd)
    try:
        np.save(path, pcm_host)
        out = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", path, str(rate), str(budget_s)], cwd=ROOT,
                             check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(path)
    return json.loads(out.strip().splitlines()[-1])


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
    args = ap.parse_args()

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
                     "valu_issue": valu_issue(n_clips * ch, samples, k_avg_ms)},
    }
    if enc_traffic_note:
        result["roofline"]["traffic_note"] = enc_traffic_note
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
        declined, unpacked, misses, cheap_declined, marked = plan.counters()  # K1's fast paths: slices handed to a slower form, all launches
        result["packed_path"] = {"declined_slices": declined, "unpacked_slices": unpacked, "speculation_misses": misses, "cheap_quotient_declines": cheap_declined, "cheap_quotient_marked_wave_slices": marked,
                                 "slices_total": (args.warmup + args.steps) * n_clips * ch * ((samples + 19) // 20)}
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
        back = torch.empty_like(pcm)
        # untimed warm-up: after the host-side pause above the clocks keep rising for ~30 launches (40 ms) of this 1.5-ms
        # kernel (kernel_ms_runs showed the 10 launches behind an 8-launch warm-up still falling by 5 %)
        for _ in range(24):
            dplan.run(out, back, stream)
        # the timed launches are enqueued back to back like the encode steps (a host sync after every launch lets the
        # clocks sag between them: +5 % on this 1.6-ms kernel), one HIP event pair per launch
        d_events = []
        for _ in range(max(20, args.steps)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dplan.run(out, back, stream)
            e1.record(stream)
            d_events.append((e0, e1))
        torch.cuda.synchronize()
        dev_ms = [a.elapsed_time(b) for a, b in d_events]
        d_ms = float(np.mean(dev_ms))
        d_bytes = dplan.in_bytes + dplan.pcm_bytes
        dec_traffic, dec_traffic_note = pmc_traffic("decode", n_clips, ch, args.frames)
        result["decode"] = {"value": dplan.samples / (d_ms * 1e-3), "unit": "samples/s", "kernel_ms_avg": d_ms,
                            "kernel_ms_runs": [round(v, 4) for v in dev_ms],
                            "roofline": {"bound": "hbm", "achieved": d_bytes / (d_ms * 1e-3) / 1e9,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": d_bytes / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "traffic": dec_traffic,
                                         "kernel": "qoa_decode_uniform_kernel" if ch <= 2 else "qoa_decode_frames_kernel",
                                         "algorithmic_bytes_per_launch": d_bytes},
                            "header_mismatches": dplan.mismatches(),
                            "round_trip_rms": float((back[:8].float() - pcm[:8].float()).pow(2).mean().sqrt())}
        if dec_traffic_note:
            result["decode"]["roofline"]["traffic_note"] = dec_traffic_note

    if rank == 0 and world == 1 and not args.no_extras and args.frames % 8 == 0:
        # K1 outside the headline shape (records for DESIGN.md, not part of `value`): the same 2048 chains at 2 chains per
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
                               "step_form": "LDS-table (451 VALU + 26 LDS per slice)" if fc_clips * ch // 4 >= 8192
                               else f"cheap-quotient register form ({K1_VALU_PER_SLICE:.0f} VALU per slice)",
                               "valu_issue": valu_issue(fc_clips * ch, fc_samples, f_ms,
                                                        451.0 if fc_clips * ch // 4 >= 8192 else K1_VALU_PER_SLICE),
                               "bit_exact_vs_oracle": bool(f_ok)}
        result["k1_extras"] = extras
        del fplan, fout
        torch.cuda.synchronize()
        if not args.no_decode:
            # K2u on config 2's shape (1024 mono clips x 256 frames): half the samples of config 3, the same launch geometry
            mdesc = q.QoaDesc(1, args.rate, samples)
            mclips, _, mper_enc = batch.uniform_clips(n_clips, mdesc)
            mpcm = pcm[:, :, 0].contiguous().view(n_clips, samples, 1)
            menc = torch.empty((n_clips, mper_enc), dtype=torch.uint8, device=dev)
            batch.EncodePlan(mclips).run(mpcm, menc, stream)
            mplan = batch.DecodePlan.from_clips(mclips)
            mback = torch.empty_like(mpcm)
            for _ in range(24):
                mplan.run(menc, mback, stream)
            m_events = []
            for _ in range(max(20, args.steps)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream); mplan.run(menc, mback, stream); e1.record(stream)
                m_events.append((e0, e1))
            torch.cuda.synchronize()
            m_ms = float(np.mean([a.elapsed_time(b) for a, b in m_events]))
            m_bytes = mplan.in_bytes + mplan.pcm_bytes
            from oracle import oracle as O
            m_ok = np.array_equal(mback[7].cpu().numpy().reshape(-1), O.decode_all(menc[7].cpu().numpy().tobytes())[2])
            result["k2_extras"] = {"mono": {"workload": f"{n_clips} clips x 1 ch x {args.frames} frames, decode",
                                            "kernel_ms_avg": m_ms, "samples_per_s": mplan.samples / (m_ms * 1e-3),
                                            "roofline": {"bound": "hbm", "achieved": m_bytes / (m_ms * 1e-3) / 1e9,
                                                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                         "frac": m_bytes / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                         "algorithmic_bytes_per_launch": m_bytes},
                                            "header_mismatches": mplan.mismatches(), "bit_exact_vs_oracle": bool(m_ok)}}
            del mplan, mback, menc, mpcm

    if rank == 0 and world == 1 and not args.no_cpu:
        sample_clips = min(n_clips, 96)
        host = pcm[:sample_clips].cpu().numpy()
        cpu = cpu_baseline_child(host, args.rate, args.cpu_seconds)
        one, n1 = cpu["one_thread"], cpu["one_thread"]["clips"]
        he = out[:n1].cpu().numpy()
        same = all(hashlib.sha256(he[i].tobytes()).hexdigest() == cpu["sha256"][i] for i in range(n1))
        result["cpu_baseline"] = {"value": one["value"], "unit": "samples/s", "cores": 1, "kind": "port",
                                  "sample": f"{n1} of the same {n_clips} clips ({one['samples']} samples, "
                                            f"{one['seconds']:.1f} s), oracle ref_style encoder, gcc -O3 -flto, 1 thread",
                                  "gpu_bytes_equal_cpu_bytes": bool(same)}
        if "all_cores" in cpu:
            ac = cpu["all_cores"]
            result["cpu_baseline"]["all_cores"] = {"value": ac["value"], "cores": ac["cores"],
                                                   "sample": f"{ac['clips']} clips, one clip per process, "
                                                             f"{ac['seconds']:.1f} s"}
        result["speedup_vs_cpu_1_thread"] = value / one["value"]

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
