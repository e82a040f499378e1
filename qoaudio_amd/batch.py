"""Batch encode/decode on device-resident buffers: the data-parallel hot path.

torch is plumbing here (device memory, streams); the work is qoa_encode_plan_run /
qoa_decode_plan_run in libqoaudio_amd.so.  A batch is semantically `encode_all` per clip
(src/lib.rs:760-763): clips are independent, every (clip, channel) is one serial chain.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .api import DecodeError, EncodeError, QoaDesc, encoded_size

PLAN_DEFAULT = 0
PLAN_FORCE_GENERIC_PATH = 1
PLAN_FORCE_32BIT_STEP = 2
PLAN_NO_SPECULATION = 4
PLAN_LDS_TABLES = 8
PLAN_NO_LDS_TABLES = 128
PLAN_EXACT_QUOTIENT = 256
PLAN_NO_PENALTY_FORM = 1024
PLAN_NO_BLOCK32 = 2048
PLAN_NO_HELPER = 4096
PLAN_FORCE_HELPER = 8192
PLAN_NO_WIDE_FORM = 16384
PLAN_WAVE_STATS = 32768  # per-wavefront form counts and cycles, written once at each wavefront's exit (EncodePlan.wave_stats)
PLAN_COUNTERS = 512  # count the fast paths' fall-backs (EncodePlan.stats / .counters); off by default: no atomics in the kernel
PLAN_CPW_1, PLAN_CPW_2, PLAN_CPW_4 = 1 << 4, 2 << 4, 4 << 4
MULTI_ALLOW_DUPLICATE_DEVICES = 1


def _clips_array(clips: Sequence[Tuple[int, int, QoaDesc]]):
    arr = (_lib.Clip * len(clips))()
    for i, (pcm_off, out_off, d) in enumerate(clips):
        arr[i].pcm_offset = pcm_off
        arr[i].out_offset = out_off
        arr[i].desc = d._c()
    return arr


def _stream_ptr(stream: Optional[torch.cuda.Stream]):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _require_cuda(t: torch.Tensor, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (qoaudio_amd has no CPU path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous {dtype}")


class EncodePlan:
    """Chain table for a batch of clips, resident on the current device."""

    def __init__(self, clips: Sequence[Tuple[int, int, QoaDesc]], flags: int = PLAN_DEFAULT):
        self._h = C.c_void_p(None)
        arr = _clips_array(clips)
        rc = _lib.lib().qoa_encode_plan_create(arr, len(clips), flags, C.byref(self._h))
        if rc:
            self._h = None
            raise EncodeError(rc)
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _lib.lib().qoa_encode_plan_bytes(self._h, C.byref(a), C.byref(b), C.byref(c))
        self.pcm_bytes, self.out_bytes, self.samples = a.value, b.value, c.value

    def __del__(self):
        try:  # module globals may already be torn down at interpreter exit
            if getattr(self, "_h", None):
                _lib.lib().qoa_encode_plan_free(self._h)
                self._h = None
        except Exception:
            pass

    def run(self, pcm: torch.Tensor, out: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Enqueue the batch encode (asynchronous).  pcm: int16 device tensor, out: uint8 device tensor."""
        _require_cuda(pcm, torch.int16, "pcm")
        _require_cuda(out, torch.uint8, "out")
        rc = _lib.lib().qoa_encode_plan_run(self._h, C.c_void_p(pcm.data_ptr()), C.c_void_p(out.data_ptr()),
                                            _stream_ptr(stream))
        if rc:
            raise EncodeError(rc)

    def launch_info(self) -> Tuple[int, int, bool]:
        """(wavefronts over all launches, chains per wavefront of the largest launch, LDS-table step form?)."""
        return self.launch_shape()[:3]

    def launch_shape(self) -> Tuple[int, int, bool, bool]:
        """launch_info() + whether the largest launch runs helper wavefronts (qoa_encode_plan_launch_shape)."""
        sh = _lib.LaunchShape()
        sh.struct_bytes = C.sizeof(sh)
        rc = _lib.lib().qoa_encode_plan_launch_shape(self._h, C.byref(sh))
        if rc:
            raise EncodeError(rc)
        return sh.wavefronts, sh.chains_per_wavefront, bool(sh.lds_table_form), bool(sh.helper_wavefronts)

    def wave_stats(self) -> dict:
        """What the LAST run cost wavefront by wavefront (plans created with PLAN_WAVE_STATS; qoa_encode_plan_wave_stats):
        wavefront-slices per step form, measured cycles per slice (min / p50 / max over wavefronts), the slowest wavefront's
        form mix with the cycles that mix predicts, and the bound no input can exceed."""
        ws = _lib.WaveStats()
        ws.struct_bytes = C.sizeof(ws)
        rc = _lib.lib().qoa_encode_plan_wave_stats(self._h, C.byref(ws))
        if rc:
            raise EncodeError(rc)
        names = _lib.WAVE_FORM_NAMES
        return {"wavefronts": ws.wavefronts, "slices": {n: int(v) for n, v in zip(names, ws.slices)},
                "issue_slots": {n: int(v) for n, v in zip(names, ws.issue_slots)}, "cycles_per_issue": ws.cycles_per_issue,
                "loop_overhead": ws.loop_overhead, "bound_cycles_per_slice": ws.bound_cycles_per_slice,
                "cycles_per_slice": {"min": ws.cycles_per_slice_min, "p50": ws.cycles_per_slice_p50, "max": ws.cycles_per_slice_max},
                "slowest_wavefront": {"index": ws.slowest_wavefront, "channels": ws.slowest_channels,
                                      "slices_of_chain": ws.slowest_slices_of_chain,
                                      "form_mix": {n: int(v) for n, v in zip(names, ws.slowest_slices)},
                                      "predicted_cycles_per_slice": ws.slowest_predicted_cycles_per_slice}}

    def stats(self) -> Tuple[int, int, int]:
        """(declined, unpacked, speculation_misses): slices the kernel's fast paths handed to a slower form, summed over
        the plan's runs so far (see qoa_encode_plan_stats).  Diagnostics only (synchronises the device)."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = _lib.lib().qoa_encode_plan_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        if rc:
            raise EncodeError(rc)
        return a.value, b.value, c.value

    def counters(self) -> Tuple[int, ...]:
        """stats() + (slices the cheap-quotient step form declined, (candidate, slice) pairs it marked,
        wavefront-slices on the 32-bit / tail path, wavefront-slices on the one-slice packed statement, slices the
        penalty form declined, slices the wide form declined, wavefront-slices the wide form settled)
        (qoa_encode_plan_counters)."""
        out = (C.c_uint64 * 10)()
        rc = _lib.lib().qoa_encode_plan_counters(self._h, out, 10)
        if rc:
            raise EncodeError(rc)
        return tuple(int(v) for v in out)


class DecodePlan:
    """Frame table for a batch of encoded streams, resident on the current device."""

    def __init__(self):
        self._h = C.c_void_p(None)
        self.status: List[int] = []
        self.samples_out: List[int] = []

    @classmethod
    def from_clips(cls, clips: Sequence[Tuple[int, int, QoaDesc]]):
        """Streams laid out as EncodePlan wrote them: (pcm_offset = PCM destination, out_offset = encoded source)."""
        self = cls()
        arr = _clips_array(clips)
        rc = _lib.lib().qoa_decode_plan_from_clips(arr, len(clips), C.byref(self._h))
        if rc:
            self._h = None
            raise DecodeError(rc)
        self._fill_bytes()
        return self

    @classmethod
    def from_streams(cls, streams: Sequence[bytes], in_offsets: Sequence[int], out_offsets: Sequence[int]):
        """Parse host copies of arbitrary QOA streams the way the reference decoder walks them."""
        self = cls()
        n = len(streams)
        bufs = [np.frombuffer(bytes(s), dtype=np.uint8) for s in streams]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data_as(C.c_void_p).value or 0 for b in bufs])
        lens = (C.c_size_t * n)(*[b.size for b in bufs])
        ino = (C.c_uint64 * n)(*in_offsets)
        outo = (C.c_uint64 * n)(*out_offsets)
        status = (C.c_int * n)()
        nout = (C.c_uint64 * n)()
        rc = _lib.lib().qoa_decode_plan_from_streams(ptrs, lens, ino, outo, n, status, nout, C.byref(self._h))
        if rc:
            self._h = None
            raise DecodeError(rc)
        self.status = list(status)
        self.samples_out = list(nout)
        self._fill_bytes()
        return self

    def _fill_bytes(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _lib.lib().qoa_decode_plan_bytes(self._h, C.byref(a), C.byref(b), C.byref(c))
        self.in_bytes, self.pcm_bytes, self.samples = a.value, b.value, c.value

    def __del__(self):
        try:  # module globals may already be torn down at interpreter exit
            if getattr(self, "_h", None):
                _lib.lib().qoa_decode_plan_free(self._h)
                self._h = None
        except Exception:
            pass

    def run(self, enc: torch.Tensor, pcm: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        _require_cuda(enc, torch.uint8, "enc")
        _require_cuda(pcm, torch.int16, "pcm")
        rc = _lib.lib().qoa_decode_plan_run(self._h, C.c_void_p(enc.data_ptr()), C.c_void_p(pcm.data_ptr()),
                                            _stream_ptr(stream))
        if rc:
            raise DecodeError(rc)

    def mismatches(self) -> int:
        n = C.c_uint32(0)
        rc = _lib.lib().qoa_decode_plan_mismatches(self._h, C.byref(n))
        if rc:
            raise DecodeError(rc)
        return n.value


def uniform_clips(n_clips: int, desc: QoaDesc):
    """Contiguous layout of n identical-shape clips: PCM [n, samples, channels], encoded [n, encoded_size]."""
    per_pcm = desc.samples * desc.channels
    per_enc = encoded_size(desc.channels, desc.samples)
    return [(i * per_pcm, i * per_enc, desc) for i in range(n_clips)], per_pcm, per_enc


def encode_batch(pcm: torch.Tensor, sample_rate: int, flags: int = PLAN_DEFAULT) -> torch.Tensor:
    """pcm: int16 device tensor [n_clips, samples, channels] -> uint8 device tensor [n_clips, encoded_size]."""
    n, samples, channels = pcm.shape
    desc = QoaDesc(channels, sample_rate, samples)
    clips, _, per_enc = uniform_clips(n, desc)
    plan = EncodePlan(clips, flags)
    out = torch.empty((n, per_enc), dtype=torch.uint8, device=pcm.device)
    plan.run(pcm, out)
    return out


def encode_batch_host(pcm: torch.Tensor, clips: Sequence[Tuple[int, int, QoaDesc]], out: torch.Tensor,
                      chunk_frames: int = 0) -> None:
    """Host-memory batch encode through the time-sliced pipeline (qoa_encode_batch_host).  pcm: int16 CPU tensor
    (pinned for asynchronous copies), out: uint8 CPU tensor; offsets in `clips` index into them."""
    if pcm.is_cuda or out.is_cuda:
        raise ValueError("encode_batch_host takes host tensors; use EncodePlan for device tensors")
    if pcm.dtype != torch.int16 or out.dtype != torch.uint8 or not pcm.is_contiguous() or not out.is_contiguous():
        raise ValueError("pcm must be contiguous int16 and out contiguous uint8")
    arr = _clips_array(clips)
    rc = _lib.lib().qoa_encode_batch_host(arr, len(clips), C.c_void_p(pcm.data_ptr()), C.c_void_p(out.data_ptr()),
                                          chunk_frames)
    if rc:
        raise EncodeError(rc)


def _host_args(pcm: torch.Tensor, enc: torch.Tensor):
    if pcm.is_cuda or enc.is_cuda:
        raise ValueError("the host pipeline takes host tensors; use EncodePlan / DecodePlan for device tensors")
    if pcm.dtype != torch.int16 or enc.dtype != torch.uint8 or not pcm.is_contiguous() or not enc.is_contiguous():
        raise ValueError("pcm must be contiguous int16 and the encoded buffer contiguous uint8")


def decode_batch_host(enc: torch.Tensor, clips: Sequence[Tuple[int, int, QoaDesc]], pcm: torch.Tensor,
                      chunk_frames: int = 0) -> None:
    """Host-memory batch decode through the frame-sliced pipeline (qoa_decode_batch_host): the inverse of
    encode_batch_host with the same clip list (out_offset locates a clip's QOA stream in `enc`, pcm_offset its PCM)."""
    _host_args(pcm, enc)
    arr = _clips_array(clips)
    rc = _lib.lib().qoa_decode_batch_host(arr, len(clips), C.c_void_p(enc.data_ptr()), C.c_void_p(pcm.data_ptr()),
                                          chunk_frames)
    if rc:
        raise DecodeError(rc)


class HostPipeline:
    """Persistent form of encode_batch_host / decode_batch_host: streams, events and device staging buffers are created
    once and reused by every call (qoa_host_pipeline_*)."""

    def __init__(self):
        self._h = C.c_void_p(None)
        rc = _lib.lib().qoa_host_pipeline_create(C.byref(self._h))
        if rc:
            self._h = None
            raise EncodeError(rc)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().qoa_host_pipeline_free(self._h)
                self._h = None
        except Exception:
            pass

    def encode(self, pcm: torch.Tensor, clips, out: torch.Tensor, chunk_frames: int = 0) -> None:
        _host_args(pcm, out)
        arr = _clips_array(clips)
        rc = _lib.lib().qoa_host_pipeline_encode(self._h, arr, len(clips), C.c_void_p(pcm.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), chunk_frames)
        if rc:
            raise EncodeError(rc)

    def decode(self, enc: torch.Tensor, clips, pcm: torch.Tensor, chunk_frames: int = 0) -> None:
        _host_args(pcm, enc)
        arr = _clips_array(clips)
        rc = _lib.lib().qoa_host_pipeline_decode(self._h, arr, len(clips), C.c_void_p(enc.data_ptr()),
                                                 C.c_void_p(pcm.data_ptr()), chunk_frames)
        if rc:
            raise DecodeError(rc)


def multi_shard(n_clips: int, n_devices: int, index: int) -> Tuple[int, int]:
    """(first, count) of shard `index` of the block partition qoa_multi_* uses (= dist.shard_range)."""
    a, b = C.c_size_t(0), C.c_size_t(0)
    if _lib.lib().qoa_multi_shard(n_clips, n_devices, index, C.byref(a), C.byref(b)):
        raise ValueError("bad shard index")
    return a.value, b.value


class MultiGpu:
    """Several GPUs driven from ONE process through the C ABI (qoa_multi_*): device i encodes / decodes shard i of the
    clip list with its own host thread and time-sliced pipeline; with `out` on devices[0] the stage-out of every slice is
    the peer copy that gathers the encoded streams on the root (include/qoaudio_amd.h)."""

    def __init__(self, devices: Sequence[int], allow_duplicate_devices: bool = False):
        """allow_duplicate_devices: test mode for boxes with fewer GPUs than shards (QOA_MULTI_ALLOW_DUPLICATE_DEVICES): a
        device may be listed several times; every shard but the first then takes the staging + copy path of a peer GPU."""
        self.devices = list(devices)
        self._h = C.c_void_p(None)
        self._marshalled = None  # (clips object, ctypes array): a repeated batch is marshalled once
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = _lib.lib().qoa_multi_create_flags(arr, len(self.devices), MULTI_ALLOW_DUPLICATE_DEVICES if allow_duplicate_devices else 0,
                                               C.byref(self._h))
        if rc:
            self._h = None
            raise EncodeError(rc)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().qoa_multi_free(self._h)
                self._h = None
        except Exception:
            pass

    def _clips(self, clips):
        """ctypes image of the clip list; a list with the same contents handed in again (a loop over one batch) is not
        marshalled again.  The key is a digest of EVERY entry (offsets and desc), not the list's identity: a caller that
        edits an entry of the same list in place gets a fresh image (advisor, round 4: the C side would have launched
        kernels at the stale offsets)."""
        sig = hash(tuple((po, oo, d.channels, d.sample_rate, d.samples) for po, oo, d in clips))
        if self._marshalled is None or self._marshalled[0] != (len(clips), sig):
            self._marshalled = ((len(clips), sig), _clips_array(clips))
        return self._marshalled[1]

    def _bases(self, tensors):
        if len(tensors) != len(self.devices):
            raise ValueError("one PCM tensor per device")
        for t in tensors:
            if t.dtype != torch.int16 or not t.is_contiguous():
                raise ValueError("PCM tensors must be contiguous int16")
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def encode(self, pcm: Sequence[torch.Tensor], clips, out: torch.Tensor, chunk_frames: int = 0) -> None:
        """pcm[i]: shard i's PCM (host tensor, or device tensor on devices[i]; the shard's pcm_offsets count from its start);
        out: ONE uint8 tensor for the whole batch (host, or on devices[0])."""
        if out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError("out must be contiguous uint8")
        arr = self._clips(clips)
        rc = _lib.lib().qoa_multi_encode(self._h, arr, len(clips), self._bases(pcm), C.c_void_p(out.data_ptr()), chunk_frames)
        if rc:
            raise EncodeError(rc)

    def decode(self, enc: torch.Tensor, clips, pcm: Sequence[torch.Tensor], chunk_frames: int = 0) -> None:
        if enc.dtype != torch.uint8 or not enc.is_contiguous():
            raise ValueError("enc must be contiguous uint8")
        arr = self._clips(clips)
        rc = _lib.lib().qoa_multi_decode(self._h, arr, len(clips), C.c_void_p(enc.data_ptr()), self._bases(pcm), chunk_frames)
        if rc:
            raise DecodeError(rc)


def decode_batch(enc: torch.Tensor, desc: QoaDesc) -> torch.Tensor:
    """enc: uint8 device tensor [n_clips, encoded_size] as encode_batch wrote it -> int16 [n, samples, channels]."""
    n = enc.shape[0]
    clips, _, _ = uniform_clips(n, desc)
    plan = DecodePlan.from_clips(clips)
    pcm = torch.empty((n, desc.samples, desc.channels), dtype=torch.int16, device=enc.device)
    plan.run(enc, pcm)
    bad = plan.mismatches()  # synchronises
    if bad:
        raise DecodeError(3)
    return pcm
