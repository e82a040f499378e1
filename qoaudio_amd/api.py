"""Python mirror of rafaelcaricio/qoaudio's public API over the C ABI (include/qoaudio_amd.h).

Same names, argument meaning and error behaviour as the crate (src/lib.rs), so the parity tests in
tests/ read like the reference's own tests.  Every compute call lands in the gfx950 kernels through
libqoaudio_amd.so; nothing here computes samples.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Iterator, Optional, Union

import numpy as np

from . import _lib

# src/lib.rs:12-19
QOA_SLICE_LEN = 20
QOA_LMS_LEN = 4
QOA_HEADER_SIZE = 8
QOA_MAGIC = 0x716F6166
MAX_SLICES_PER_CHANNEL_PER_FRAME = 256
QOA_SLICES_PER_FRAME = 256
QOA_FRAME_LEN = QOA_SLICES_PER_FRAME * QOA_SLICE_LEN
QOA_MAX_CHANNELS = 8


class EncodeError(Exception):
    """EncodeError, src/lib.rs:106-115 (Display strings :897-906)."""
    KINDS = {1: "InvalidChannels", 2: "InvalidSampleRate", 3: "InvalidSamples", 4: "IoError"}

    def __init__(self, code: int):
        L = _lib.lib()
        self.code = code
        self.kind = self.KINDS.get(code, "Unknown")
        msg = L.qoa_encode_strerror(code).decode()
        if code == 4:
            msg += L.qoa_last_error_detail().decode()
        super().__init__(msg)


class DecodeError(Exception):
    """DecodeError, src/lib.rs:866-873 (Display strings :877-887)."""
    KINDS = {1: "NotQoaFile", 2: "NoSamples", 3: "InvalidFrameHeader", 4: "IncompatibleFrame", 5: "IoError"}

    def __init__(self, code: int):
        L = _lib.lib()
        self.code = code
        self.kind = self.KINDS.get(code, "Unknown")
        msg = L.qoa_decode_strerror(code).decode()
        if code == 5:
            msg += L.qoa_last_error_detail().decode()
        super().__init__(msg)


@dataclass
class QoaDesc:
    """src/lib.rs:95-102"""
    channels: int
    sample_rate: int
    samples: int

    def _c(self):
        if not (0 <= self.channels <= 255 and 0 <= self.sample_rate < 2 ** 32 and 0 <= self.samples < 2 ** 32):
            raise OverflowError("QoaDesc field out of range (u8 / u32 / u32)")
        return _lib.Desc(self.channels, self.sample_rate, self.samples)


@dataclass(frozen=True)
class FrameHeader:
    """src/lib.rs:774-781"""
    num_channels: int = 0
    sample_rate: int = 0
    num_samples_per_channel: int = 0


@dataclass(frozen=True)
class FixedSamples:
    """ProcessingMode::FixedSamples, src/lib.rs:34-41"""
    channels: int
    sample_rate: int
    samples: int


@dataclass(frozen=True)
class Streaming:
    """ProcessingMode::Streaming, src/lib.rs:42-44"""


ProcessingMode = Union[FixedSamples, Streaming]


@dataclass(frozen=True)
class Sample:
    """QoaItem::Sample, src/lib.rs:657"""
    value: int


QoaItem = Union[Sample, FrameHeader]


@dataclass
class DecodedQoa:
    """src/lib.rs:696-704"""
    num_channels: int
    sample_rate: int
    samples: np.ndarray


def _pcm_arg(sample_data):
    a = np.ascontiguousarray(sample_data, dtype=np.int16)
    return a, a.ctypes.data_as(C.c_void_p)


def encoded_size(channels: int, samples: int) -> int:
    """Exact output size of encode (src/lib.rs:374-378)."""
    return _lib.lib().qoa_encoded_size(channels, samples)


class QoaEncoder:
    """QoaEncoder, src/lib.rs:85-91, :333-439."""

    def __init__(self, desc: QoaDesc):
        self._h = C.c_void_p(None)
        self._desc = desc
        d = desc._c()
        rc = _lib.lib().qoa_encoder_new(C.byref(d), C.byref(self._h))
        if rc:
            self._h = None
            raise EncodeError(rc)

    new = classmethod(lambda cls, desc: cls(desc))

    def __del__(self):
        try:  # module globals may already be torn down at interpreter exit
            if getattr(self, "_h", None):
                _lib.lib().qoa_encoder_free(self._h)
                self._h = None
        except Exception:
            pass

    def encode(self, sample_data) -> bytes:
        """src/lib.rs:367-398"""
        a, p = _pcm_arg(sample_data)
        cap = encoded_size(self._desc.channels, self._desc.samples)
        out = np.empty(cap, dtype=np.uint8)
        w = C.c_size_t(0)
        rc = _lib.lib().qoa_encoder_encode(self._h, p, a.size, out.ctypes.data_as(C.c_void_p), cap, C.byref(w))
        if rc:
            raise EncodeError(rc)
        return out[: w.value].tobytes()

    def write_header(self, writer=None) -> bytes:
        """src/lib.rs:404-408.  Appends to `writer` (bytearray or file-like) when given."""
        buf = (C.c_uint8 * 8)()
        rc = _lib.lib().qoa_encoder_write_header(self._h, buf)
        if rc:
            raise EncodeError(rc)
        return _emit(writer, bytes(buf))

    def encode_frame(self, sample_data, writer=None):
        """src/lib.rs:418-439.  Returns sample-times encoded (and the frame bytes when no writer is given)."""
        a, p = _pcm_arg(sample_data)
        cap = 8 + 16 * QOA_MAX_CHANNELS + 8 * QOA_SLICES_PER_FRAME * QOA_MAX_CHANNELS
        out = np.empty(cap, dtype=np.uint8)
        w, fl = C.c_size_t(0), C.c_size_t(0)
        rc = _lib.lib().qoa_encoder_encode_frame(self._h, p, a.size, out.ctypes.data_as(C.c_void_p), cap,
                                                 C.byref(w), C.byref(fl))
        if rc:
            raise EncodeError(rc)
        data = out[: w.value].tobytes()
        if writer is None:
            return fl.value, data
        _emit(writer, data)
        return fl.value

    def lms_state(self) -> np.ndarray:
        """Carried LMS state, [channels, 8] = history[4] | weights[4] (checkpoint hook)."""
        s = np.empty((self._desc.channels, 8), dtype=np.int32)
        rc = _lib.lib().qoa_encoder_get_lms(self._h, s.ctypes.data_as(C.c_void_p))
        if rc:
            raise EncodeError(rc)
        return s

    def debug_set_plan_flags(self, flags: int) -> None:
        """Test hook (qoa_debug_encoder_set_plan_flags): later launches of this handle take a plan's step-form switches
        (batch.PLAN_*); the bytes do not depend on them."""
        rc = _lib.lib().qoa_debug_encoder_set_plan_flags(self._h, flags)
        if rc:
            raise EncodeError(rc)

    def set_lms_state(self, state) -> None:
        s = np.ascontiguousarray(state, dtype=np.int32).reshape(self._desc.channels, 8)
        rc = _lib.lib().qoa_encoder_set_lms(self._h, s.ctypes.data_as(C.c_void_p))
        if rc:
            raise EncodeError(rc)


def _emit(writer, data: bytes):
    if writer is None:
        return data
    if isinstance(writer, bytearray):
        writer.extend(data)
    else:
        writer.write(data)
    return data


def encode_all(sample_data, desc: QoaDesc) -> bytes:
    """src/lib.rs:760-763"""
    return QoaEncoder(desc).encode(sample_data)


class QoaDecoder:
    """QoaDecoder<R>, src/lib.rs:59-68, :117-201, :619-693.  `reader` is bytes-like, or a file-like object that is pulled
    LAZILY, one frame at a time, as the iterator advances (qoa_decoder_new_reader): the GPU decodes whole frames, the
    iterator drains them in the reference's item order, and into_inner() hands the reader back where the last pulled
    frame ended."""

    def __init__(self, reader=None, *, _streaming=False):
        self._h = C.c_void_p(None)
        self._reader = None
        self._owned = None        # a file this object opened itself (QoaDecoder.open): closed with the decoder
        # what the reader raised (any exception): an I/O error of the stream, src/lib.rs:889-893.  A one-element holder that
        # the callback closes over instead of `self`: self -> _cb -> _pull -> self would be a reference cycle, the decoder
        # reclaimed only by the cyclic collector and a file opened by QoaDecoder.open() closed whenever that runs
        self._read_error_box = [None]
        L = _lib.lib()
        if _streaming:
            rc = L.qoa_decoder_new_streaming(C.byref(self._h))
        elif hasattr(reader, "read"):
            self._reader = reader
            box = self._read_error_box

            def _pull(_ctx, buf, n):  # qoa_read_fn: bytes stored, 0 at the end, negative for an I/O error
                try:
                    chunk = reader.read(n)
                    if chunk is None:  # a non-blocking reader with nothing to give: io::ErrorKind::WouldBlock, not the end
                        raise BlockingIOError("reader returned None (no data available)")
                    if len(chunk) > n:
                        raise ValueError("reader returned more bytes than asked for")
                except Exception as e:  # ctypes would swallow it and report 0 = a clean end of stream
                    box[0] = e
                    return -1
                if not chunk:
                    return 0
                C.memmove(buf, bytes(chunk), len(chunk))
                return len(chunk)

            self._cb = _lib.READ_FN(_pull)  # must outlive the handle
            rc = L.qoa_decoder_new_reader(self._cb, None, C.byref(self._h))
        else:
            buf = np.frombuffer(bytes(reader), dtype=np.uint8)
            rc = L.qoa_decoder_new(buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size, C.byref(self._h))
        if rc:
            self._h = None
            raise self._error(rc)
        self._done = False

    def _error(self, rc):
        """DecodeError(rc), chained to the exception the reader raised when that is what it reports."""
        err = DecodeError(rc)
        cause, self._read_error_box[0] = self._read_error_box[0], None
        if cause is not None and err.kind == "IoError":
            err.__cause__ = cause
        return err

    new = classmethod(lambda cls, reader: cls(reader))

    @classmethod
    def open(cls, path):
        """src/lib.rs:619-625"""
        try:
            f = open(path, "rb")  # io::BufReader<File>: read lazily, src/lib.rs:619-625
        except OSError as e:
            err = DecodeError.__new__(DecodeError)
            Exception.__init__(err, f"IO error: {e}")
            err.code, err.kind = 5, "IoError"
            raise err from e
        try:
            dec = cls(f)
        except BaseException:
            f.close()
            raise
        dec._owned = f  # closed by close() / when the decoder is dropped; into_inner() hands it over instead
        return dec

    @classmethod
    def new_streaming(cls):
        """src/lib.rs:632-638"""
        return cls(_streaming=True)

    def close(self) -> None:
        """Drop the handle; a file QoaDecoder.open() opened is closed with it."""
        if getattr(self, "_h", None):
            _lib.lib().qoa_decoder_free(self._h)
            self._h = None
        f, self._owned = getattr(self, "_owned", None), None
        if f is not None:
            f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:  # module globals may already be torn down at interpreter exit
            self.close()
        except Exception:
            pass

    def into_inner(self):
        """src/lib.rs:172-178: the reader, positioned behind the last frame the decoder pulled from it."""
        r, self._reader = self._reader, None
        self._owned = None  # the caller's now
        if getattr(self, "_h", None):
            _lib.lib().qoa_decoder_free(self._h)
            self._h = None
        return r

    def mode(self) -> ProcessingMode:
        m = _lib.Mode()
        _lib.lib().qoa_decoder_mode(self._h, C.byref(m))
        return Streaming() if m.kind == 1 else FixedSamples(m.channels, m.sample_rate, m.samples)

    def current_frame_header(self) -> FrameHeader:
        h = _lib.FrameHeader()
        _lib.lib().qoa_decoder_current_frame_header(self._h, C.byref(h))
        return FrameHeader(h.num_channels, h.sample_rate, h.num_samples_per_channel)

    def total_duration(self) -> Optional[float]:
        d = _lib.lib().qoa_decoder_total_duration(self._h)
        return None if d < 0 else d

    def next(self):
        """Iterator::next, src/lib.rs:668-692: a QoaItem, None at the end, or raises DecodeError (Some(Err))."""
        it = _lib.Item()
        _lib.lib().qoa_decoder_next(self._h, C.byref(it))
        if it.kind == 0:
            return Sample(it.sample)
        if it.kind == 1:
            return FrameHeader(it.header.num_channels, it.header.sample_rate, it.header.num_samples_per_channel)
        if it.kind == 2:
            return None
        raise self._error(it.error)

    def __iter__(self) -> Iterator[QoaItem]:
        while True:
            it = self.next()
            if it is None:
                return
            yield it

    def read_samples(self, max_samples: int) -> np.ndarray:
        """Bulk form of next(): consecutive Sample items up to the next FrameHeader / end."""
        out = np.empty(max_samples, dtype=np.int16)
        n = _lib.lib().qoa_decoder_read_samples(self._h, out.ctypes.data_as(C.c_void_p), max_samples)
        return out[:n]

    def decode_frame(self, frame_data) -> np.ndarray:
        """src/lib.rs:641-651"""
        buf = np.frombuffer(bytes(frame_data), dtype=np.uint8)
        n = C.c_size_t(0)
        L = _lib.lib()
        rc = L.qoa_decoder_decode_frame(self._h, buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size,
                                        None, 0, C.byref(n))
        if n.value == 0:
            if rc:
                raise DecodeError(rc)
            return np.empty(0, dtype=np.int16)
        out = np.empty(n.value, dtype=np.int16)
        rc = L.qoa_decoder_decode_frame(self._h, None, 0, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
        if rc:
            raise DecodeError(rc)
        return out


def decode_all(reader) -> DecodedQoa:
    """src/lib.rs:713-747 -- the same walk as qoa_decode_all in the C ABI, driven through the decoder handle so the
    stream is decoded on the GPU once and drained in bulk.  decode_all consumes its reader (`R: io::Read` by value, :713), so
    a file-like reader is read to its end first and the whole stream goes to the GPU in one piece (the lazy, frame-at-a-time
    pull is QoaDecoder's: one upload + launch + download per frame is the wrong shape for a bulk drain)."""
    if hasattr(reader, "read"):
        try:
            reader = reader.read()
            if reader is None:
                raise BlockingIOError("reader returned None (no data available)")
        except Exception as e:
            err = DecodeError.__new__(DecodeError)
            Exception.__init__(err, f"IO error: {e}")
            err.code, err.kind = 5, "IoError"
            raise err from e
    decoder = QoaDecoder(reader)
    first = decoder.next()  # :724-731: always the first FrameHeader
    assert isinstance(first, FrameHeader)
    chunks = []
    while True:
        item = decoder.next()
        if item is None:
            break
        if isinstance(item, Sample):
            chunks.append(np.array([item.value], dtype=np.int16))
            rest = decoder.read_samples(QOA_FRAME_LEN * 13 * 255)
            if rest.size:
                chunks.append(rest.copy())
        elif item.num_channels != first.num_channels or item.sample_rate != first.sample_rate:  # :735-738
            raise DecodeError(4)
    samples = np.concatenate(chunks) if chunks else np.empty(0, dtype=np.int16)
    return DecodedQoa(first.num_channels, first.sample_rate, samples)


def decode_all_capi(data) -> DecodedQoa:
    """decode_all through the single C entry point qoa_decode_all (two passes: size query, then fill)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    p = buf.ctypes.data_as(C.c_void_p) if buf.size else None
    L = _lib.lib()
    ch, sr, n = C.c_uint8(0), C.c_uint32(0), C.c_size_t(0)
    rc = L.qoa_decode_all(p, buf.size, C.byref(ch), C.byref(sr), None, 0, C.byref(n))
    if rc:
        raise DecodeError(rc)
    out = np.empty(n.value, dtype=np.int16)
    if n.value:
        rc = L.qoa_decode_all(p, buf.size, C.byref(ch), C.byref(sr), out.ctypes.data_as(C.c_void_p), out.size,
                              C.byref(n))
        if rc:
            raise DecodeError(rc)
    return DecodedQoa(ch.value, sr.value, out)


def open_and_decode_all(path) -> DecodedQoa:
    """src/lib.rs:750-754"""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError as e:
        err = DecodeError.__new__(DecodeError)
        Exception.__init__(err, f"IO error: {e}")
        err.code, err.kind = 5, "IoError"
        raise err from e
    return decode_all(data)
