"""The N>1 path on CPU: world_size-2 gloo processes shard a batch of clips, 'encode' their shard (the
CPU oracle stands in for the GPU kernel here -- it is the checker, and this test is about the sharding
and the gather, not the codec) and gather the fixed-size encoded buffers to rank 0."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, rng_pcm


def test_shard_range_partitions_exactly():
    from qoaudio_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1024, 8192, 8195):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(8192, r, 8) for r in range(8)] == [(1024 * r, 1024 * (r + 1)) for r in range(8)]


def _worker(rank, world, port, n_clips, samples, channels, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import oracle as O
    from qoaudio_amd import dist as qdist
    r, w, _ = qdist.init_from_env("gloo")
    a, b = qdist.shard_range(n_clips, r, w)
    per_enc = O.encoded_size(channels, samples)
    local = torch.empty((b - a, per_enc), dtype=torch.uint8)
    for i in range(a, b):
        x = rng_pcm(1000 + i, samples * channels)
        local[i - a] = torch.frombuffer(bytearray(O.encode_all(x, channels, 44100, samples)), dtype=torch.uint8)
    out = qdist.gather_encoded(local, n_clips, dst=0)
    # the double-buffered asynchronous form bench.py uses for N > 1 (round 5: uneven partitions too -- the padded path)
    bufs = [local.clone(), local.clone()]
    dsts = [torch.zeros((n_clips, per_enc), dtype=torch.uint8) if rank == 0 else None for _ in bufs]
    works = [qdist.gather_encoded_async(bufs[b], n_clips, dst=0, out=dsts[b])[0] for b in range(2)]
    for wk in works:
        wk.wait()
    if rank == 0:
        assert torch.equal(dsts[0], out) and torch.equal(dsts[1], out)
    ms = qdist.barrier_max_ms(float(rank + 1), torch.device("cpu"))
    if rank == 0:
        q.put((out.numpy().copy(), ms))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [6, 7])
def test_two_rank_shard_and_gather(oracle, n_clips):
    samples, channels, world = 333, 2, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, samples, channels, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, ms = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ms == 2.0  # MAX over ranks
    assert out.shape == (n_clips, oracle.encoded_size(channels, samples))
    for i in range(n_clips):
        x = rng_pcm(1000 + i, samples * channels)
        assert out[i].tobytes() == oracle.encode_all(x, channels, 44100, samples)


@pytest.mark.parametrize("n_clips", [8192, 8195])
def test_eight_rank_shard_and_gather_at_config4_clip_count(oracle, n_clips):
    """Config 4's rank count and clip count on CPU (BASELINE.json configs[3]: 8192 clips over 8 GPUs): world size 8 over gloo,
    8,192 tiny clips (1,024 per rank: the even partition, every shard received straight into its row block) and 8,195 (the
    first three ranks hold one clip more: the padded path, synchronous and asynchronous).  The clips are one 20-sample mono
    slice each (40 encoded bytes): what is under test is shard_range + gather_encoded + gather_encoded_async at the shape
    the first 8-GPU run will have, not the codec.  No scaling claim: still unmeasured on hardware."""
    samples, channels, world = 20, 1, 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, samples, channels, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, ms = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ms == 8.0  # MAX over ranks
    per_enc = oracle.encoded_size(channels, samples)
    assert out.shape == (n_clips, per_enc) and per_enc == 40
    from qoaudio_amd.dist import shard_range
    firsts = [shard_range(n_clips, r, world)[0] for r in range(world)]
    for i in sorted(set(firsts + [f - 1 for f in firsts if f] + [n_clips - 1] + list(range(0, n_clips, 97)))):
        x = rng_pcm(1000 + i, samples * channels)
        assert out[i].tobytes() == oracle.encode_all(x, channels, 44100, samples), i


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` without a launcher starts its own rank processes -- but only after checking, without
    initialising any device, that N GPUs exist.  On this CPU-only box that check must fail fast and loudly."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--clips", "4", "--frames", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "only" in r.stderr and "GPU" in r.stderr
    assert r.stdout.strip() == ""
