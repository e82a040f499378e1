"""Clip sharding across the GPUs of one node (BASELINE config 4).

Clips are fully independent units, so a batch shards with NO data-path collective: rank r encodes a
contiguous block of clips on its own GPU.  The one exchange step the config names is the gather of the
fixed-size encoded buffers (encoded size is a pure function of the desc, src/lib.rs:374-378, so every
offset is known a priori): rank 0 receives each peer's shard over that peer's own xGMI link
(torch.distributed `gather` = grouped RCCL send/recv; backend "nccl" is RCCL on ROCm).  The same code
runs over gloo on CPU tensors, which is how tests/test_dist_gloo.py covers the N>1 path.
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init_from_env(backend: Optional[str] = None):
    """One process per GPU, launched by torch.distributed.run.  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # QOA_DIST_BACKEND=gloo: smoke-test the N > 1 control flow on a 1-GPU box (RCCL refuses two ranks on one
            # device; gloo gathers are staged through host memory below).  Never set in a measured run.
            backend = os.environ.get("QOA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


class _DoneWork:
    """Stand-in for a finished collective (the host-staged gloo debug path is synchronous)."""

    def wait(self):
        return True


class _PaddedWork:
    """An asynchronous gather of an UNEVEN partition: every rank sent n_max rows (its shard padded with zero rows), rank dst
    received them into scratch blocks; wait() waits for the collective and then trims the blocks into `out` (on the
    current stream for device tensors: the work's wait() orders that stream behind the collective)."""

    def __init__(self, work, parts, counts, out, send=None):
        self._work, self._parts, self._counts, self._out = work, parts, counts, out
        self._send = send  # (the padded copy this rank sends must outlive the collective: held until wait())

    def wait(self):
        ok = self._work.wait()
        self._send = None
        if self._parts is not None:
            for (a, b), p in zip(self._counts, self._parts):
                self._out[a:b].copy_(p[: b - a])
            self._parts = None
        return ok


def gather_encoded_async(local: torch.Tensor, n_total: int, dst: int = 0, group=None,
                         out: Optional[torch.Tensor] = None):
    """Gather that returns (work, out): the collective is enqueued (async_op=True) and the caller waits on `work`
    before it reuses `local` or reads `out`.  Lets step i's gather run on the communication stream while step i+1
    encodes into a second output buffer (bench.py, N > 1).  With an even partition every peer's shard lands straight in
    its row block of `out`; an uneven one (n_total % world != 0) takes the padded path: equal-sized sends, scratch
    blocks on dst, trimmed into `out` by work.wait()."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(n_total, r, world) for r in range(world)]
    assert local.shape[0] == counts[rank][1] - counts[rank][0], "shard does not match shard_range()"
    n_max = max(b - a for a, b in counts)
    even = all(b - a == n_max for a, b in counts)
    send = local if local.is_contiguous() else local.contiguous()
    if not even:
        if send.shape[0] != n_max:  # (a fresh buffer per call: the caller may overwrite `local` only after wait())
            padded = torch.zeros((n_max, local.shape[1]), dtype=local.dtype, device=local.device)
            padded[: local.shape[0]] = send
            send = padded
        if send.is_cuda and dist.get_backend(group) == "gloo":
            send = send.cpu()
        if rank != dst:
            return _PaddedWork(dist.gather(send, gather_list=None, dst=dst, group=group, async_op=True), None, counts, None, send), None
        if out is None:
            out = torch.empty((n_total, local.shape[1]), dtype=local.dtype, device=local.device)
        parts = [torch.empty((n_max, local.shape[1]), dtype=send.dtype, device=send.device) for _ in range(world)]
        return _PaddedWork(dist.gather(send, gather_list=parts, dst=dst, group=group, async_op=True), parts, counts, out, send), out
    if send.is_cuda and dist.get_backend(group) == "gloo":  # debug path only (see init_from_env): host-staged, synchronous
        host = send.cpu()
        parts = [torch.empty_like(host) for _ in counts] if rank == dst else None
        dist.gather(host, gather_list=parts, dst=dst, group=group)
        if rank == dst:
            if out is None:
                out = torch.empty((n_total, local.shape[1]), dtype=local.dtype, device=local.device)
            for (a, b), p in zip(counts, parts):
                out[a:b].copy_(p)
        return _DoneWork(), (out if rank == dst else None)
    if rank != dst:
        return dist.gather(send, gather_list=None, dst=dst, group=group, async_op=True), None
    if out is None:
        out = torch.empty((n_total, local.shape[1]), dtype=local.dtype, device=local.device)
    parts = [out[a:b] for a, b in counts]
    return dist.gather(send, gather_list=parts, dst=dst, group=group, async_op=True), out


def gather_encoded(local: torch.Tensor, n_total: int, dst: int = 0, group=None,
                   out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Gather every rank's [n_local, per_enc] uint8 shard (block partition of n_total clips, see shard_range)
    into [n_total, per_enc] on rank `dst`; other ranks return None.  `out` may be a preallocated destination
    (rank dst only); with an even partition every peer's shard is received straight into its slice of it."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per_enc = local.shape[1]
    counts = [shard_range(n_total, r, world) for r in range(world)]
    assert local.shape[0] == counts[rank][1] - counts[rank][0], "shard does not match shard_range()"
    n_max = max(b - a for a, b in counts)
    even = all(b - a == n_max for a, b in counts)
    send = local.contiguous()
    if not even and local.shape[0] != n_max:  # uneven tail: pad to the common size, trimmed again on dst
        send = torch.zeros((n_max, per_enc), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if rank != dst:
        dist.gather(send, gather_list=None, dst=dst, group=group)
        return None
    if out is None:
        out = torch.empty((n_total, per_enc), dtype=local.dtype, device=local.device)
    if even:
        parts = [out[a:b] for a, b in counts]  # contiguous row blocks of `out`: no staging copy
        dist.gather(send, gather_list=parts, dst=dst, group=group)
        return out
    parts = [torch.empty((n_max, per_enc), dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.gather(send, gather_list=parts, dst=dst, group=group)
    for r, (a, b) in enumerate(counts):
        out[a:b] = parts[r][: b - a]
    return out


def barrier_max_ms(elapsed_ms: float, device) -> float:
    """MAX over ranks of a per-rank elapsed time (bench.py's timing rule)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_ms
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([elapsed_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
