"""BASELINE config 4 (clips sharded one block per GPU, RCCL gather of the encoded buffers) through the HIP path on real
devices.  Needs >= 2 GPUs on the box and skips otherwise (the round's GPU test box has one; the N > 1 code path is
covered on CPU by tests/test_dist_gloo.py).  Runs bench.py exactly as a user would -- `python bench.py --gpus 2` -- so
the launcher (one fresh rank process per GPU, started before anything touches a device) is part of what is tested."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rng_pcm

pytestmark = pytest.mark.gpu


def test_two_rank_shard_encode_and_rccl_gather(gpu, oracle):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--clips", "64", "--frames", "8", "--no-cpu"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dist_backend"] == "nccl"
    assert line["bit_exact_vs_oracle"] is True
    assert line["gathered_shards_bit_exact_vs_oracle"] is True
    assert line["gathered_bytes"] == 2 * 64 * gpu.encoded_size(2, 8 * 5120)
    assert line["config"]["samples_per_step"] == 2 * 64 * 8 * 5120 * 2


def test_two_rank_control_flow_on_one_gpu(gpu, oracle):
    """The same `python bench.py --gpus 2` on a ONE-GPU box: both ranks share device 0 and the gather is staged through
    host memory over gloo (QOA_BENCH_SHARE_GPU=1 QOA_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).  Not a
    measurement -- it keeps the launcher, the double-buffered gather loop, the max-over-ranks timing and the gathered-shard
    check of the N > 1 path exercised where no second GPU exists."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QOA_BENCH_SHARE_GPU="1", QOA_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--clips", "64", "--frames", "8", "--no-cpu"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 2 and lines[0].startswith('{"bench_extras"')  # the lab records, then the contract's line (last, <= 6 KB)
    assert len(lines[-1]) <= 6144
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0
    pre = line["preflight"]  # (N > 1 default: device count, peer-access rows, a checked 1 MB gather before the timed loop)
    assert "error" not in pre and pre["gather_1mb_ok"] is True and len(pre["peer_access_rows"]) == 2 and pre["world_size"] == 2
    assert line["bit_exact_vs_oracle"] is True and line["gathered_shards_bit_exact_vs_oracle"] is True
    assert line["gathered_bytes"] == 2 * 64 * gpu.encoded_size(2, 8 * 5120)
    assert line["config"]["samples_per_step"] == 2 * 64 * 8 * 5120 * 2
    assert line["decode"]["header_mismatches"] == 0


def test_the_drivers_command_line_for_n_gpus(gpu, oracle):
    """The command the driver runs for N > 1, verbatim -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with N = 2 on this one-GPU box (both
    ranks on device 0, gather over gloo): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher's environment, rank 0
    alone prints, and its LAST line is the contract's record with the preflight in it."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QOA_BENCH_SHARE_GPU="1", QOA_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--clips", "64", "--frames", "8"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 2, out.stdout[-2000:]     # rank 0's two lines, nothing from rank 1
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["bit_exact_vs_oracle"] is True and line["gathered_shards_bit_exact_vs_oracle"] is True
    assert line["gathered_bytes"] == 2 * 64 * gpu.encoded_size(2, 8 * 5120)
    assert line["preflight"]["gather_1mb_ok"] is True and "error" not in line["preflight"]
    assert "cpu_baseline" not in line and "e2e" not in line   # (N = 1 only)


def test_two_rank_control_flow_at_config4_per_rank_shape(gpu, oracle):
    """The same two-rank run at config 4's REAL per-rank shape -- 1024 stereo clips x 256 frames per rank: 5.37 GB of PCM,
    1.08 GB encoded shards, two output buffers per rank and two 2.17 GB gather destinations on rank 0, rotated by the
    double-buffered gather loop.  Not a measurement (both ranks share device 0, the gather is staged through host memory
    over gloo); it proves the allocation sizes, the buffer rotation and the launcher at full size before an 8-GPU box
    ever sees them."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QOA_BENCH_SHARE_GPU="1", QOA_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--clips", "1024", "--frames", "256", "--no-cpu", "--no-extras", "--no-decode"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    per_enc = gpu.encoded_size(2, 256 * 5120)
    assert line["n_gpus"] == 2 and line["dist_backend"] == "gloo"
    assert line["config"]["encoded_bytes_per_gpu"] == 1024 * per_enc == 1084235776
    assert line["gathered_bytes"] == 2 * 1024 * per_enc
    assert line["bit_exact_vs_oracle"] is True and line["gathered_shards_bit_exact_vs_oracle"] is True
    assert line["config"]["samples_per_step"] == 2 * 1024 * 256 * 5120 * 2


def test_bench_abi_multi_one_device(gpu):
    """`bench.py --abi-multi --gpus 1`: the C ABI's one-process multi-GPU form with the only device count this box has;
    same JSON keys as the torch.distributed form."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--abi-multi", "--gpus", "1", "--steps", "2",
                          "--warmup", "1", "--clips", "64", "--frames", "16"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["multi_gpu_path"] == "c-abi" and line["unit"] == "samples/s"
    assert line["bit_exact_vs_oracle"] is True and line["gathered_shards_bit_exact_vs_oracle"] is True
    assert line["gathered_bytes"] == 64 * gpu.encoded_size(2, 16 * 5120)
    for key in ("metric", "value", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline"):
        assert key in line


# ---- multi-GPU behind the C ABI (qoa_multi_*, include/qoaudio_amd.h): one process, one pipeline per device ----

def _ragged_batch(gpu):
    descs = [gpu.QoaDesc(2, 44100, 3 * 5120 + 77), gpu.QoaDesc(1, 8000, 5120), gpu.QoaDesc(2, 44100, 19),
             gpu.QoaDesc(8, 48000, 2 * 5120 + 1), gpu.QoaDesc(3, 22050, 4000), gpu.QoaDesc(2, 44100, 2 * 5120)]
    clips, po, oo = [], 0, 0
    for d in descs:
        clips.append((po, oo, d))
        po += d.samples * d.channels
        oo += (gpu.encoded_size(d.channels, d.samples) + 7) // 8 * 8
    return clips, po, oo


@pytest.mark.parametrize("where", ["device", "host"])
def test_abi_multi_one_device_vs_oracle(gpu, oracle, where):
    """qoa_multi_encode / qoa_multi_decode with n_devices = 1 (what this box can run): a ragged 1..8-channel batch and a
    uniform stereo batch, PCM and output on the device or in host memory, every clip against the oracle."""
    from qoaudio_amd import batch
    m = batch.MultiGpu([0])
    dev = "cuda:0" if where == "device" else "cpu"
    for uniform in (False, True):
        if uniform:
            d = gpu.QoaDesc(2, 44100, 9 * 5120 + 333)
            clips, per_pcm, per_enc = batch.uniform_clips(24, d)
            n_pcm, n_enc = 24 * per_pcm, 24 * per_enc
        else:
            clips, n_pcm, n_enc = _ragged_batch(gpu)
        host = rng_pcm(77 + uniform, n_pcm, "mix")
        pcm = torch.from_numpy(host).to(dev)
        out = torch.zeros(n_enc, dtype=torch.uint8, device=dev)
        m.encode([pcm], clips, out, chunk_frames=2)
        he = out.cpu().numpy()
        for po, oo, d in clips:
            want = oracle.encode_all(host[po:po + d.samples * d.channels], d.channels, d.sample_rate, d.samples)
            assert he[oo:oo + len(want)].tobytes() == want
        back = torch.zeros(n_pcm, dtype=torch.int16, device=dev)
        m.decode(out, clips, [back], chunk_frames=2)
        hb = back.cpu().numpy()
        for po, oo, d in clips:
            n = d.samples * d.channels
            _, _, want = oracle.decode_all(he[oo:oo + gpu.encoded_size(d.channels, d.samples)].tobytes())
            assert np.array_equal(hb[po:po + n], want)


def test_abi_multi_errors_never_abort(gpu):
    from qoaudio_amd import batch, _lib
    n = torch.cuda.device_count()
    for devs in ([n], [-1], [0, 0], []):
        with pytest.raises(gpu.EncodeError) as e:
            batch.MultiGpu(devs)
        assert e.value.kind == "IoError"
        assert _lib.lib().qoa_last_error_detail()
    m = batch.MultiGpu([0])
    bad = [(0, 0, gpu.QoaDesc(9, 44100, 100))]
    with pytest.raises(gpu.EncodeError) as e:
        m.encode([torch.zeros(900, dtype=torch.int16)], bad, torch.zeros(4096, dtype=torch.uint8))
    assert e.value.kind == "InvalidChannels"


def test_abi_multi_two_devices(gpu, oracle):
    """The real thing: two shards on two devices, the encoded streams gathered in device 0's buffer by peer copies."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    from qoaudio_amd import batch
    d = gpu.QoaDesc(2, 44100, 6 * 5120 + 5)
    clips, per_pcm, per_enc = batch.uniform_clips(37, d)
    host = rng_pcm(5, 37 * per_pcm, "mix")
    m = batch.MultiGpu([0, 1])
    shards, bases = [], []
    for i in range(2):
        first, count = batch.multi_shard(37, 2, i)
        shards.append(torch.from_numpy(host[first * per_pcm:(first + count) * per_pcm]).to(f"cuda:{i}"))
        bases.append(first)
    # a shard's pcm_offsets count from the shard's own base
    n0 = batch.multi_shard(37, 2, 0)[1]
    rel = [(po - (0 if k < n0 else bases[1]) * per_pcm, oo, dd) for k, (po, oo, dd) in enumerate(clips)]
    out = torch.zeros(37 * per_enc, dtype=torch.uint8, device="cuda:0")
    m.encode(shards, rel, out, chunk_frames=2)
    he = out.cpu().numpy()
    for k, (po, oo, dd) in enumerate(clips):
        assert he[oo:oo + per_enc].tobytes() == oracle.encode_all(host[po:po + per_pcm], 2, 44100, d.samples)


@pytest.mark.parametrize("where", ["device", "host"])
@pytest.mark.parametrize("n_shards", [2, 3])
def test_abi_multi_duplicate_devices_vs_oracle(gpu, oracle, where, n_shards):
    """qoa_multi_* with n_devices >= 2 on a ONE-GPU box: QOA_MULTI_ALLOW_DUPLICATE_DEVICES lists device 0 several times, so
    the per-device host threads, the shard offsets relative to pcm[i], and -- for every shard but the first -- the staging
    buffers and copies into a shared buffer that is treated as a peer's (the NON-LOCAL `out` / `enc` path of a second GPU)
    all execute: ragged and uniform batches, device and host memory, every clip against the oracle, twice (the second call
    runs on the handle's cached chain / frame tables) and once more with other data in the same layout."""
    from qoaudio_amd import batch
    m = batch.MultiGpu([0] * n_shards, allow_duplicate_devices=True)
    dev = "cuda:0" if where == "device" else "cpu"
    for uniform in (False, True):
        if uniform:
            d = gpu.QoaDesc(2, 44100, 5 * 5120 + 333)
            clips, per_pcm, per_enc = batch.uniform_clips(13, d)
            n_enc = 13 * per_enc
        else:
            clips, _, n_enc = _ragged_batch(gpu)
        n = len(clips)
        shard_of, base = [], []
        for i in range(n_shards):
            first, count = batch.multi_shard(n, n_shards, i)
            shard_of += [i] * count
            base.append(clips[first][0] if count else 0)
        rel = [(po - base[shard_of[k]], oo, dd) for k, (po, oo, dd) in enumerate(clips)]  # offsets count from the shard's base
        n_pcm = clips[-1][0] + clips[-1][2].samples * clips[-1][2].channels
        for seed in (31 + uniform, 31 + uniform, 90):
            host = rng_pcm(seed, n_pcm, "mix")
            shards = []
            for i in range(n_shards):
                first, count = batch.multi_shard(n, n_shards, i)
                end = clips[first + count - 1][0] + clips[first + count - 1][2].samples * clips[first + count - 1][2].channels if count else 0
                shards.append(torch.from_numpy(host[base[i]:end].copy() if count else np.zeros(1, np.int16)).to(dev))
            out = torch.zeros(n_enc, dtype=torch.uint8, device=dev)
            m.encode(shards, rel, out, chunk_frames=2)
            he = out.cpu().numpy()
            for po, oo, d in clips:
                want = oracle.encode_all(host[po:po + d.samples * d.channels], d.channels, d.sample_rate, d.samples)
                assert he[oo:oo + len(want)].tobytes() == want
            backs = [torch.zeros_like(t) for t in shards]
            m.decode(out, rel, backs, chunk_frames=2)
            for k, (po, oo, d) in enumerate(rel):
                _, _, want = oracle.decode_all(he[oo:oo + gpu.encoded_size(d.channels, d.samples)].tobytes())
                assert np.array_equal(backs[shard_of[k]].cpu().numpy()[po:po + d.samples * d.channels], want)


def test_abi_multi_orders_itself_after_the_producer_stream(gpu, oracle):
    """Device buffers handed to qoa_multi_encode straight from the kernels that fill them (no synchronize in between, a
    side stream): the call waits for the device's outstanding work before its own streams read PCM or overwrite `out`
    (advisor finding, round 3: the in-place path ran unordered against the producer)."""
    from qoaudio_amd import batch
    d = gpu.QoaDesc(2, 44100, 3 * 5120)
    clips, per_pcm, per_enc = batch.uniform_clips(64, d)
    host = rng_pcm(11, 64 * per_pcm, "mix")
    src = torch.from_numpy(host).pin_memory()
    m = batch.MultiGpu([0])
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            big = torch.empty(64 << 20, dtype=torch.int16, device="cuda:0")
            for _ in range(8):
                big.add_(1)                                     # keeps the side stream busy ahead of the producer
            pcm = torch.empty(64 * per_pcm, dtype=torch.int16, device="cuda:0")
            pcm.copy_(src, non_blocking=True)
            pcm.add_(0)
            out = torch.empty(64 * per_enc, dtype=torch.uint8, device="cuda:0")
            out.fill_(0xEE)                                     # lands AFTER the encode if the call does not order itself
        m.encode([pcm], clips, out)
        he = out.cpu().numpy()
        for k in (0, 31, 63):
            po, oo, _ = clips[k]
            assert he[oo:oo + per_enc].tobytes() == oracle.encode_all(host[po:po + per_pcm], 2, 44100, d.samples)


def test_abi_multi_unaligned_device_output_takes_the_staging_path(gpu, oracle):
    """An encoded buffer in device memory whose base is not 8-byte aligned is accepted like the same layout in host memory
    (round 3 returned IoError for it): the shard goes through the staging buffers."""
    from qoaudio_amd import batch
    d = gpu.QoaDesc(2, 44100, 2 * 5120 + 40)
    clips, per_pcm, per_enc = batch.uniform_clips(5, d)
    host = rng_pcm(3, 5 * per_pcm, "mix")
    pcm = torch.from_numpy(host).to("cuda:0")
    raw = torch.zeros(5 * per_enc + 16, dtype=torch.uint8, device="cuda:0")
    out = raw[4:4 + 5 * per_enc]                              # base % 8 == 4
    assert out.data_ptr() % 8 == 4
    m = batch.MultiGpu([0])
    m.encode([pcm], clips, out)
    he = out.cpu().numpy()
    for po, oo, _ in clips:
        assert he[oo:oo + per_enc].tobytes() == oracle.encode_all(host[po:po + per_pcm], 2, 44100, d.samples)
    back = torch.zeros_like(pcm)
    m.decode(out, clips, [back])
    for po, oo, _ in clips:
        _, _, want = oracle.decode_all(he[oo:oo + per_enc].tobytes())
        assert np.array_equal(back.cpu().numpy()[po:po + per_pcm], want)
