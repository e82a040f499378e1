"""qoaudio_amd -- MI355X-native (gfx950) QOA encode/decode engine behind rafaelcaricio/qoaudio's API.

Host mirror of the crate's public API (api.py), batch entry points on device tensors (batch.py),
clip sharding across ranks (dist.py).  Everything computes in libqoaudio_amd.so (HIP kernels).
"""
from .api import (  # noqa: F401
    MAX_SLICES_PER_CHANNEL_PER_FRAME, QOA_FRAME_LEN, QOA_HEADER_SIZE, QOA_LMS_LEN, QOA_MAGIC,
    QOA_MAX_CHANNELS, QOA_SLICE_LEN, QOA_SLICES_PER_FRAME,
    DecodedQoa, DecodeError, EncodeError, FixedSamples, FrameHeader, ProcessingMode, QoaDecoder, QoaDesc,
    QoaEncoder, QoaItem, Sample, Streaming, decode_all, decode_all_capi, encode_all, encoded_size,
    open_and_decode_all,
)
