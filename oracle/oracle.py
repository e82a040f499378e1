"""ctypes binding of the CPU oracle (oracle/qoa_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under qoaudio_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

REF_STYLE = 0
EXHAUSTIVE = 1

ENC_ERRORS = {1: "InvalidChannels", 2: "InvalidSampleRate", 3: "InvalidSamples", 4: "IoError"}
DEC_ERRORS = {1: "NotQoaFile", 2: "NoSamples", 3: "InvalidFrameHeader", 4: "IncompatibleFrame", 5: "IoError"}

ITEM_SAMPLE, ITEM_FRAME_HEADER, ITEM_END, ITEM_ERROR = 0, 1, 2, 3


class OracleError(Exception):
    def __init__(self, kind, code):
        super().__init__(kind)
        self.kind = kind
        self.code = code


class Desc(C.Structure):
    _fields_ = [("channels", C.c_uint8), ("sample_rate", C.c_uint32), ("samples", C.c_uint32)]


class Lms(C.Structure):
    _fields_ = [("history", C.c_int32 * 4), ("weights", C.c_int32 * 4)]


class Encoder(C.Structure):
    _fields_ = [("desc", Desc), ("lms", Lms * 8), ("search", C.c_int),
                ("candidate_steps", C.c_uint64), ("slices", C.c_uint64)]


class FrameHeader(C.Structure):
    _fields_ = [("num_channels", C.c_uint8), ("sample_rate", C.c_uint32),
                ("num_samples_per_channel", C.c_uint16)]


class Item(C.Structure):
    _fields_ = [("kind", C.c_int), ("sample", C.c_int16), ("header", FrameHeader), ("error", C.c_int)]


def build(san=False):
    target = "liboracle_san.so" if san else "liboracle.so"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, target)


_lib = None


def lib():
    global _lib
    if _lib is None:
        # QOA_ORACLE_LIB=liboracle_native.so selects the -march=native build (oracle/cpu_bench.py's second leg)
        name = os.environ.get("QOA_ORACLE_LIB", "liboracle.so")
        path = os.path.join(_HERE, name)
        src = os.path.join(_HERE, "qoa_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, name])
        L = C.CDLL(path)
        L.oq_encoded_size.restype = C.c_size_t
        L.oq_encoded_size.argtypes = [C.c_uint32, C.c_uint32]
        L.oq_frame_size.restype = C.c_uint16
        L.oq_div.restype = C.c_int32
        L.oq_div.argtypes = [C.c_int32, C.c_int]
        L.oq_encode_slice.restype = C.c_uint64
        L.oq_encode_slice.argtypes = [C.POINTER(Lms), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.c_int, C.POINTER(C.c_int), C.POINTER(Lms),
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oq_encode_all.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Desc), C.c_int, C.c_void_p,
                                    C.c_size_t, C.POINTER(C.c_size_t)]
        L.oq_encoder_init.argtypes = [C.POINTER(Encoder), C.POINTER(Desc), C.c_int]
        L.oq_encoder_encode.argtypes = [C.POINTER(Encoder), C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_size_t, C.POINTER(C.c_size_t)]
        L.oq_encoder_write_header.argtypes = [C.POINTER(Encoder), C.c_void_p]
        L.oq_encoder_encode_frame.argtypes = [C.POINTER(Encoder), C.c_void_p, C.c_size_t, C.c_void_p,
                                              C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.oq_decoder_new.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.oq_decoder_new_streaming.argtypes = [C.POINTER(C.c_void_p)]
        L.oq_decoder_free.argtypes = [C.c_void_p]
        L.oq_decoder_mode.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]
        L.oq_decoder_current_frame_header.argtypes = [C.c_void_p, C.POINTER(FrameHeader)]
        L.oq_decoder_next.argtypes = [C.c_void_p, C.POINTER(Item)]
        L.oq_decoder_lms.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Lms)]
        L.oq_decode_slice.argtypes = [C.POINTER(Lms), C.c_uint64, C.c_void_p, C.c_size_t]
        L.oq_decode_slice.restype = None
        L.oq_decoder_decode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                              C.c_size_t, C.POINTER(C.c_size_t)]
        L.oq_decode_all.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def _pcm(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a, a.ctypes.data_as(C.c_void_p)


def encoded_size(channels, samples):
    return lib().oq_encoded_size(channels, samples)


def encode_all(pcm, channels, sample_rate, samples, search=REF_STYLE):
    """encode_all(sample_data, &QoaDesc{..}) -> bytes  (src/lib.rs:760-763)."""
    a, p = _pcm(pcm)
    d = Desc(channels, sample_rate, samples)
    cap = max(encoded_size(max(channels, 1), samples), 16) if 0 < channels <= 8 else 16
    out = np.empty(cap, dtype=np.uint8)
    w = C.c_size_t(0)
    rc = lib().oq_encode_all(p, a.size, C.byref(d), search, out.ctypes.data_as(C.c_void_p), cap, C.byref(w))
    if rc:
        raise OracleError(ENC_ERRORS[rc], rc)
    return out[: w.value].tobytes()


class OracleEncoder:
    """QoaEncoder (src/lib.rs:333-439) over the C oracle."""

    def __init__(self, channels, sample_rate, samples, search=REF_STYLE):
        self.e = Encoder()
        d = Desc(channels, sample_rate, samples)
        rc = lib().oq_encoder_init(C.byref(self.e), C.byref(d), search)
        if rc:
            raise OracleError(ENC_ERRORS[rc], rc)

    def encode(self, pcm):
        a, p = _pcm(pcm)
        cap = encoded_size(self.e.desc.channels, self.e.desc.samples)
        out = np.empty(cap, dtype=np.uint8)
        w = C.c_size_t(0)
        rc = lib().oq_encoder_encode(C.byref(self.e), p, a.size, out.ctypes.data_as(C.c_void_p), cap, C.byref(w))
        if rc:
            raise OracleError(ENC_ERRORS[rc], rc)
        return out[: w.value].tobytes()

    def write_header(self):
        out = (C.c_uint8 * 8)()
        lib().oq_encoder_write_header(C.byref(self.e), out)
        return bytes(out)

    def encode_frame(self, pcm):
        a, p = _pcm(pcm)
        cap = 8 + 16 * 8 + 256 * 8 * 8
        out = np.empty(cap, dtype=np.uint8)
        w, fl = C.c_size_t(0), C.c_size_t(0)
        rc = lib().oq_encoder_encode_frame(C.byref(self.e), p, a.size, out.ctypes.data_as(C.c_void_p), cap,
                                           C.byref(w), C.byref(fl))
        if rc:
            raise OracleError(ENC_ERRORS[rc], rc)
        return out[: w.value].tobytes(), fl.value

    @property
    def candidate_steps(self):
        return self.e.candidate_steps

    @property
    def slices(self):
        return self.e.slices

    def lms_state(self):
        return [(list(self.e.lms[c].history), list(self.e.lms[c].weights)) for c in range(self.e.desc.channels)]


def encode_slice(history, weights, pcm, search=EXHAUSTIVE):
    """One slice from an explicit entry state; returns (slice_u64, sf, (hist, weights), totals[16])."""
    entry = Lms((C.c_int32 * 4)(*history), (C.c_int32 * 4)(*weights))
    a, p = _pcm(pcm)
    sf = C.c_int(0)
    best = Lms()
    totals = (C.c_uint64 * 16)()
    steps = C.c_uint64(0)
    s = lib().oq_encode_slice(C.byref(entry), p, 0, a.size, 1, search, C.byref(sf), C.byref(best), totals,
                              C.byref(steps))
    return s, sf.value, (list(best.history), list(best.weights)), list(totals)


def slice_rank_terms(history, weights, pcm):
    """All 16 candidates of one slice run to the end: (err_sq[16], pen_sq[16], first_err_sq[16], first_pen_sq[16]);
    a candidate's rank is err_sq + pen_sq (src/lib.rs:576)."""
    entry = Lms((C.c_int32 * 4)(*history), (C.c_int32 * 4)(*weights))
    a, p = _pcm(pcm)
    arrs = [(C.c_uint64 * 16)() for _ in range(4)]
    L = lib()
    L.oq_slice_rank_terms.argtypes = [C.POINTER(Lms), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t] + [C.c_void_p] * 4
    L.oq_slice_rank_terms.restype = None
    L.oq_slice_rank_terms(C.byref(entry), p, 0, a.size, 1, *arrs)
    return tuple(list(x) for x in arrs)


def slice_candidates(history, weights, pcm):
    """The slice word each of the 16 scalefactor candidates produces when run to the end (no tail shift)."""
    entry = Lms((C.c_int32 * 4)(*history), (C.c_int32 * 4)(*weights))
    a, p = _pcm(pcm)
    words = (C.c_uint64 * 16)()
    L = lib()
    L.oq_slice_candidates.argtypes = [C.POINTER(Lms), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    L.oq_slice_candidates.restype = None
    L.oq_slice_candidates(C.byref(entry), p, 0, a.size, 1, words)
    return list(words)


def decode_slice(history, weights, slice_u64):
    """One channel's slice from an explicit state (src/lib.rs:300-318): (int16[20], (history, weights) after it)."""
    st = Lms((C.c_int32 * 4)(*history), (C.c_int32 * 4)(*weights))
    out = np.empty(20, dtype=np.int16)
    lib().oq_decode_slice(C.byref(st), slice_u64, out.ctypes.data_as(C.c_void_p), 1)
    return out, (list(st.history), list(st.weights))


def decode_all(data):
    """decode_all(reader) -> (channels, sample_rate, int16 ndarray)  (src/lib.rs:713-747)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = C.c_size_t(0)
    ch, sr = C.c_uint8(0), C.c_uint32(0)
    p = buf.ctypes.data_as(C.c_void_p) if buf.size else None
    rc = lib().oq_decode_all(p, buf.size, C.byref(ch), C.byref(sr), None, 0, C.byref(n))
    if rc:
        raise OracleError(DEC_ERRORS[rc], rc)
    out = np.empty(n.value, dtype=np.int16)
    rc = lib().oq_decode_all(p, buf.size, C.byref(ch), C.byref(sr), out.ctypes.data_as(C.c_void_p), out.size,
                             C.byref(n))
    assert rc == 0
    return ch.value, sr.value, out


def decode_all_into(buf, out):
    """decode_all (src/lib.rs:713-747) in ONE pass into a caller-allocated int16 array that is large enough (decode_all
    above walks the stream twice: once for the count).  buf: uint8 ndarray, out: int16 ndarray.  Returns
    (channels, sample_rate, samples written) -- the timed form of bench.py's CPU decode baseline."""
    n = C.c_size_t(0)
    ch, sr = C.c_uint8(0), C.c_uint32(0)
    rc = lib().oq_decode_all(buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(ch), C.byref(sr),
                             out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
    if rc:
        raise OracleError(DEC_ERRORS[rc], rc)
    if n.value > out.size:
        raise ValueError(f"stream holds {n.value} samples, buffer {out.size}")
    return ch.value, sr.value, n.value


class OracleDecoder:
    """QoaDecoder<Cursor<Vec<u8>>> (src/lib.rs:117-201, 627-693) over the C oracle."""

    def __init__(self, data=None, streaming=False):
        self.h = C.c_void_p(None)
        if streaming:
            rc = lib().oq_decoder_new_streaming(C.byref(self.h))
        else:
            buf = np.frombuffer(bytes(data), dtype=np.uint8)
            rc = lib().oq_decoder_new(buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size,
                                      C.byref(self.h))
        if rc:
            raise OracleError(DEC_ERRORS[rc], rc)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oq_decoder_free(self.h)
            self.h = None

    def mode(self):
        ch, sr, n = C.c_uint8(0), C.c_uint32(0), C.c_uint32(0)
        m = lib().oq_decoder_mode(self.h, C.byref(ch), C.byref(sr), C.byref(n))
        return ("Streaming",) if m == 1 else ("FixedSamples", ch.value, sr.value, n.value)

    def lms(self, channel):
        """The decoder's predictor state of one channel (src/lib.rs:63): (history, weights)."""
        st = Lms()
        if lib().oq_decoder_lms(self.h, channel, C.byref(st)):
            raise IndexError(channel)
        return list(st.history), list(st.weights)

    def current_frame_header(self):
        h = FrameHeader()
        lib().oq_decoder_current_frame_header(self.h, C.byref(h))
        return (h.num_channels, h.sample_rate, h.num_samples_per_channel)

    def next(self):
        it = Item()
        lib().oq_decoder_next(self.h, C.byref(it))
        if it.kind == ITEM_SAMPLE:
            return ("Sample", it.sample)
        if it.kind == ITEM_FRAME_HEADER:
            return ("FrameHeader", (it.header.num_channels, it.header.sample_rate,
                                    it.header.num_samples_per_channel))
        if it.kind == ITEM_END:
            return None
        return ("Err", DEC_ERRORS[it.error])

    def __iter__(self):
        while True:
            it = self.next()
            if it is None:
                return
            yield it
            if it[0] == "Err":
                return

    def decode_frame(self, frame):
        buf = np.frombuffer(bytes(frame), dtype=np.uint8)
        out = np.empty(65535 * 8 + 64, dtype=np.int16)
        n = C.c_size_t(0)
        rc = lib().oq_decoder_decode_frame(self.h, buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size,
                                           out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
        if rc:
            raise OracleError(DEC_ERRORS[rc], rc)
        return out[: n.value].copy()
