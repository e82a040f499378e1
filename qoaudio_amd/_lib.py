"""ctypes loader for libqoaudio_amd.so (the C ABI of include/qoaudio_amd.h).

The library is the product: if it is missing this module raises -- there is no Python or CPU
fallback anywhere in the package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QOAUDIO_AMD_LIB: development aid -- A/B timing of two builds of the library through the same Python harness)
LIB_PATH = os.environ.get("QOAUDIO_AMD_LIB") or os.path.join(_HERE, "libqoaudio_amd.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/qoaudio_amd.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "qoa_encode_strerror", "qoa_decode_strerror", "qoa_last_error_detail", "qoa_version",
    "qoa_encoded_size", "qoa_frame_size",
    "qoa_encoder_new", "qoa_encoder_free", "qoa_encoder_encode", "qoa_encoder_write_header",
    "qoa_encoder_encode_frame", "qoa_encoder_get_lms", "qoa_encoder_set_lms", "qoa_debug_encode_slice", "qoa_debug_encoder_set_plan_flags", "qoa_encode_all",
    "qoa_encode_plan_create", "qoa_encode_plan_free", "qoa_encode_plan_bytes", "qoa_encode_plan_run", "qoa_encode_plan_launch_info", "qoa_encode_plan_launch_shape", "qoa_encode_plan_wave_stats", "qoa_encode_plan_stats", "qoa_encode_plan_counters",
    "qoa_encode_batch_host", "qoa_host_pipeline_create", "qoa_host_pipeline_free", "qoa_host_pipeline_encode",
    "qoa_host_pipeline_decode", "qoa_decode_batch_host",
    "qoa_multi_create", "qoa_multi_create_flags", "qoa_multi_free", "qoa_multi_shard", "qoa_multi_encode", "qoa_multi_decode",
    "qoa_encode_batch_multi", "qoa_decode_batch_multi",
    "qoa_decoder_new", "qoa_decoder_new_reader", "qoa_decoder_new_streaming", "qoa_decoder_free", "qoa_decoder_mode",
    "qoa_decoder_current_frame_header", "qoa_decoder_total_duration", "qoa_decoder_next",
    "qoa_decoder_read_samples", "qoa_decoder_decode_frame", "qoa_decode_all",
    "qoa_stream_scan", "qoa_decode_plan_from_streams", "qoa_decode_plan_from_clips", "qoa_decode_plan_free",
    "qoa_decode_plan_bytes", "qoa_decode_plan_run", "qoa_decode_plan_mismatches",
]


class Desc(C.Structure):
    _fields_ = [("channels", C.c_uint8), ("sample_rate", C.c_uint32), ("samples", C.c_uint32)]


class Clip(C.Structure):
    _fields_ = [("pcm_offset", C.c_uint64), ("out_offset", C.c_uint64), ("desc", Desc)]


class LaunchShape(C.Structure):  # qoa_launch_shape
    _fields_ = [("struct_bytes", C.c_uint32), ("wavefronts", C.c_uint32), ("chains_per_wavefront", C.c_uint32),
                ("lds_table_form", C.c_uint32), ("helper_wavefronts", C.c_uint32), ("launches", C.c_uint32)]


WAVE_FORMS = 8
WAVE_FORM_NAMES = ("cheap_quotient", "exact_quotient", "penalty", "wide", "full_packed", "block_32bit", "one_slice", "blocks")


class WaveStats(C.Structure):  # qoa_wave_stats
    _fields_ = [("struct_bytes", C.c_uint32), ("wavefronts", C.c_uint32), ("slices", C.c_uint64 * WAVE_FORMS),
                ("issue_slots", C.c_uint32 * WAVE_FORMS), ("cycles_per_issue", C.c_double), ("loop_overhead", C.c_double),
                ("bound_cycles_per_slice", C.c_double),
                ("cycles_per_slice_min", C.c_double), ("cycles_per_slice_p50", C.c_double), ("cycles_per_slice_max", C.c_double),
                ("slowest_wavefront", C.c_uint32), ("slowest_channels", C.c_uint32), ("slowest_slices_of_chain", C.c_uint32),
                ("slowest_slices", C.c_uint32 * WAVE_FORMS), ("slowest_predicted_cycles_per_slice", C.c_double)]


class FrameHeader(C.Structure):
    _fields_ = [("num_channels", C.c_uint8), ("sample_rate", C.c_uint32), ("num_samples_per_channel", C.c_uint16)]


class Mode(C.Structure):
    _fields_ = [("kind", C.c_int), ("channels", C.c_uint8), ("sample_rate", C.c_uint32), ("samples", C.c_uint32)]


class StreamInfo(C.Structure):
    _fields_ = [("mode", Mode), ("terminal", C.c_int), ("in_new", C.c_int), ("n_frames", C.c_uint64),
                ("n_samples", C.c_uint64)]


class Item(C.Structure):
    _fields_ = [("kind", C.c_int), ("sample", C.c_int16), ("header", FrameHeader), ("error", C.c_int)]


# qoa_read_fn: int64_t (*)(void *ctx, uint8_t *buf, size_t len)
READ_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)


def build(force=False):
    """Compile the HIP kernels + C ABI for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; load it first so that this process has ONE HIP runtime (a second
    # copy loaded from /opt/rocm finds "no ROCm-capable device").  torch is plumbing only.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(qoaudio_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8)
    szp = C.POINTER(C.c_size_t)
    L.qoa_encode_strerror.restype = C.c_char_p
    L.qoa_encode_strerror.argtypes = [C.c_int]
    L.qoa_decode_strerror.restype = C.c_char_p
    L.qoa_decode_strerror.argtypes = [C.c_int]
    L.qoa_last_error_detail.restype = C.c_char_p
    L.qoa_version.restype = C.c_char_p
    L.qoa_encoded_size.restype = sz
    L.qoa_encoded_size.argtypes = [C.c_uint32, C.c_uint32]
    L.qoa_frame_size.restype = C.c_uint16
    L.qoa_frame_size.argtypes = [C.c_uint32, C.c_uint32]
    L.qoa_encoder_new.argtypes = [C.POINTER(Desc), C.POINTER(vp)]
    L.qoa_encoder_free.argtypes = [vp]
    L.qoa_encoder_free.restype = None
    L.qoa_encoder_encode.argtypes = [vp, vp, sz, vp, sz, szp]
    L.qoa_encoder_write_header.argtypes = [vp, vp]
    L.qoa_encoder_encode_frame.argtypes = [vp, vp, sz, vp, sz, szp, szp]
    L.qoa_encoder_get_lms.argtypes = [vp, vp]
    L.qoa_encoder_set_lms.argtypes = [vp, vp]
    L.qoa_debug_encode_slice.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.qoa_debug_encoder_set_plan_flags.argtypes = [vp, C.c_uint32]
    L.qoa_encode_all.argtypes = [vp, sz, C.POINTER(Desc), vp, sz, szp]
    L.qoa_encode_plan_create.argtypes = [C.POINTER(Clip), sz, C.c_uint32, C.POINTER(vp)]
    L.qoa_encode_plan_free.argtypes = [vp]
    L.qoa_encode_plan_free.restype = None
    L.qoa_encode_plan_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.qoa_encode_plan_run.argtypes = [vp, vp, vp, vp]
    L.qoa_encode_plan_launch_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.qoa_encode_plan_launch_shape.argtypes = [vp, C.POINTER(LaunchShape)]
    L.qoa_encode_plan_wave_stats.argtypes = [vp, C.POINTER(WaveStats)]
    L.qoa_encode_plan_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.qoa_encode_plan_counters.argtypes = [vp, C.POINTER(C.c_uint64), sz]
    L.qoa_encode_batch_host.argtypes = [C.POINTER(Clip), sz, vp, vp, C.c_uint32]
    L.qoa_decode_batch_host.argtypes = [C.POINTER(Clip), sz, vp, vp, C.c_uint32]
    L.qoa_host_pipeline_create.argtypes = [C.POINTER(vp)]
    L.qoa_host_pipeline_free.argtypes = [vp]
    L.qoa_host_pipeline_free.restype = None
    L.qoa_host_pipeline_encode.argtypes = [vp, C.POINTER(Clip), sz, vp, vp, C.c_uint32]
    L.qoa_host_pipeline_decode.argtypes = [vp, C.POINTER(Clip), sz, vp, vp, C.c_uint32]
    ip = C.POINTER(C.c_int)
    L.qoa_multi_create.argtypes = [ip, C.c_int, C.POINTER(vp)]
    L.qoa_multi_create_flags.argtypes = [ip, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.qoa_multi_free.argtypes = [vp]
    L.qoa_multi_free.restype = None
    L.qoa_multi_shard.argtypes = [sz, C.c_int, C.c_int, szp, szp]
    L.qoa_multi_encode.argtypes = [vp, C.POINTER(Clip), sz, C.POINTER(vp), vp, C.c_uint32]
    L.qoa_multi_decode.argtypes = [vp, C.POINTER(Clip), sz, vp, C.POINTER(vp), C.c_uint32]
    L.qoa_encode_batch_multi.argtypes = [ip, C.c_int, C.POINTER(Clip), sz, C.POINTER(vp), vp, C.c_uint32]
    L.qoa_decode_batch_multi.argtypes = [ip, C.c_int, C.POINTER(Clip), sz, vp, C.POINTER(vp), C.c_uint32]
    L.qoa_decoder_new.argtypes = [vp, sz, C.POINTER(vp)]
    L.qoa_decoder_new_reader.argtypes = [READ_FN, vp, C.POINTER(vp)]
    L.qoa_decoder_new_streaming.argtypes = [C.POINTER(vp)]
    L.qoa_decoder_free.argtypes = [vp]
    L.qoa_decoder_free.restype = None
    L.qoa_decoder_mode.argtypes = [vp, C.POINTER(Mode)]
    L.qoa_decoder_current_frame_header.argtypes = [vp, C.POINTER(FrameHeader)]
    L.qoa_decoder_total_duration.argtypes = [vp]
    L.qoa_decoder_total_duration.restype = C.c_double
    L.qoa_decoder_next.argtypes = [vp, C.POINTER(Item)]
    L.qoa_decoder_read_samples.argtypes = [vp, vp, sz]
    L.qoa_decoder_read_samples.restype = sz
    L.qoa_decoder_decode_frame.argtypes = [vp, vp, sz, vp, sz, szp]
    L.qoa_decode_all.argtypes = [vp, sz, u8p, C.POINTER(C.c_uint32), vp, sz, szp]
    L.qoa_stream_scan.argtypes = [vp, sz, C.POINTER(StreamInfo)]
    L.qoa_decode_plan_from_streams.argtypes = [C.POINTER(vp), szp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), sz,
                                               C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(vp)]
    L.qoa_decode_plan_from_clips.argtypes = [C.POINTER(Clip), sz, C.POINTER(vp)]
    L.qoa_decode_plan_free.argtypes = [vp]
    L.qoa_decode_plan_free.restype = None
    L.qoa_decode_plan_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.qoa_decode_plan_run.argtypes = [vp, vp, vp, vp]
    L.qoa_decode_plan_mismatches.argtypes = [vp, C.POINTER(C.c_uint32)]
    _lib = L
    return L
