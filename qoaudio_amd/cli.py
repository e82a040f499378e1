"""WAV <-> QOA command line, the counterpart of the reference's examples/encode.rs and examples/decode.rs
(SURVEY 8(f) N3).  16-bit PCM WAV only, 1-8 channels, like the reference example (examples/encode.rs:32-95).

    python -m qoaudio_amd.cli encode in.wav out.qoa
    python -m qoaudio_amd.cli decode in.qoa out.wav
    python -m qoaudio_amd.cli info in.qoa
    python -m qoaudio_amd.cli encode-dir wav_dir qoa_dir     # every *.wav of a directory as ONE batch (host pipeline)
    python -m qoaudio_amd.cli decode-dir qoa_dir wav_dir     # every *.qoa of a directory as ONE batch

A single file is 1-8 serial chains and cannot use the GPU; the directory commands are what a caller
with many files should use: all files go through qoa_host_pipeline_{encode,decode} as one time-sliced batch.
"""
import argparse
import os
import struct
import sys
import wave

import numpy as np

from . import api


def encode(src: str, dst: str) -> int:
    with wave.open(src, "rb") as w:
        channels, width, rate, frames = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        if width != 2:
            print(f"error: only 16-bit PCM WAV is supported (got {8 * width}-bit)", file=sys.stderr)
            return 2
        pcm = np.frombuffer(w.readframes(frames), dtype="<i2")
    desc = api.QoaDesc(channels, rate, pcm.size // channels)
    data = api.encode_all(pcm, desc)
    with open(dst, "wb") as f:
        f.write(data)
    print(f"{src}: {channels} ch, {rate} Hz, {desc.samples} samples/ch -> {dst}: {len(data)} bytes "
          f"({pcm.nbytes / len(data):.2f}:1)")
    return 0


def decode(src: str, dst: str) -> int:
    d = api.open_and_decode_all(src)
    with wave.open(dst, "wb") as w:
        w.setnchannels(d.num_channels)
        w.setsampwidth(2)
        w.setframerate(d.sample_rate)
        w.writeframes(d.samples.astype("<i2").tobytes())
    print(f"{src} -> {dst}: {d.num_channels} ch, {d.sample_rate} Hz, {d.samples.size // max(d.num_channels, 1)} samples/ch")
    return 0


def _read_wav(path):
    with wave.open(path, "rb") as w:
        channels, width, rate, frames = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        if width != 2:
            raise wave.Error(f"{path}: only 16-bit PCM WAV is supported (got {8 * width}-bit)")
        return channels, rate, np.frombuffer(w.readframes(frames), dtype="<i2")


def encode_dir(src_dir: str, dst_dir: str) -> int:
    """Every *.wav of src_dir -> dst_dir/<name>.qoa, all files as one batch through the host pipeline."""
    import torch
    from . import batch
    names = sorted(n for n in os.listdir(src_dir) if n.lower().endswith(".wav"))
    if not names:
        print(f"error: no .wav files in {src_dir}", file=sys.stderr)
        return 2
    os.makedirs(dst_dir, exist_ok=True)
    clips, parts, po, oo = [], [], 0, 0
    for n in names:
        ch, rate, pcm = _read_wav(os.path.join(src_dir, n))
        desc = api.QoaDesc(ch, rate, pcm.size // ch)
        clips.append((po, oo, desc))
        parts.append(pcm)
        po += pcm.size
        oo += (api.encoded_size(ch, desc.samples) + 7) // 8 * 8
    pcm = torch.from_numpy(np.concatenate(parts).astype(np.int16))
    out = torch.zeros(oo, dtype=torch.uint8)
    batch.HostPipeline().encode(pcm, clips, out)
    host = out.numpy()
    for n, (_, o, d) in zip(names, clips):
        with open(os.path.join(dst_dir, os.path.splitext(n)[0] + ".qoa"), "wb") as f:
            f.write(host[o:o + api.encoded_size(d.channels, d.samples)].tobytes())
    print(f"{len(names)} files, {pcm.numel()} samples -> {dst_dir} ({pcm.numel() * 2 / max(oo, 1):.2f}:1)")
    return 0


def decode_dir(src_dir: str, dst_dir: str) -> int:
    """Every *.qoa of src_dir -> dst_dir/<name>.wav as one batch.  Files must be what an encoder writes: a fixed sample count
    in the file header and constant channels / rate (read from the first frame header); anything else -- streaming files,
    damaged frames -- is reported by the device-side header check and should go through `decode` one by one."""
    import torch
    from . import batch
    names = sorted(n for n in os.listdir(src_dir) if n.lower().endswith(".qoa"))
    if not names:
        print(f"error: no .qoa files in {src_dir}", file=sys.stderr)
        return 2
    os.makedirs(dst_dir, exist_ok=True)
    clips, blobs, po, oo = [], [], 0, 0
    for n in names:
        data = open(os.path.join(src_dir, n), "rb").read()
        if len(data) < 16 or data[:4] != b"qoaf":
            raise api.DecodeError(1)
        samples = struct.unpack(">I", data[4:8])[0]
        ch, rate = data[8], int.from_bytes(data[9:12], "big")
        desc = api.QoaDesc(ch, rate, samples)
        if samples == 0 or len(data) != api.encoded_size(ch, samples):
            print(f"error: {n}: not a fixed-length QOA file of the size its header implies; use `decode`", file=sys.stderr)
            return 2
        clips.append((po, oo, desc))
        blobs.append(np.frombuffer(data, np.uint8))
        po += samples * ch
        oo += (len(data) + 7) // 8 * 8
    enc = torch.zeros(oo, dtype=torch.uint8)
    host = enc.numpy()
    for (_, o, _), b in zip(clips, blobs):
        host[o:o + b.size] = b
    pcm = torch.zeros(po, dtype=torch.int16)
    batch.HostPipeline().decode(enc, clips, pcm)
    hp = pcm.numpy()
    for n, (p0, _, d) in zip(names, clips):
        with wave.open(os.path.join(dst_dir, os.path.splitext(n)[0] + ".wav"), "wb") as w:
            w.setnchannels(d.channels)
            w.setsampwidth(2)
            w.setframerate(d.sample_rate)
            w.writeframes(hp[p0:p0 + d.samples * d.channels].astype("<i2").tobytes())
    print(f"{len(names)} files, {po} samples -> {dst_dir}")
    return 0


def info(src: str) -> int:
    dec = api.QoaDecoder.open(src)
    print("mode:", dec.mode())
    dur = dec.total_duration()
    if dur is not None:
        print(f"duration: {dur:.3f} s")
    print("first frame:", dec.next())
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="qoaudio_amd.cli", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("encode", "decode", "encode-dir", "decode-dir"):
        p = sub.add_parser(name)
        p.add_argument("src")
        p.add_argument("dst")
    p = sub.add_parser("info")
    p.add_argument("src")
    a = ap.parse_args(argv)
    try:
        if a.cmd == "encode":
            return encode(a.src, a.dst)
        if a.cmd == "decode":
            return decode(a.src, a.dst)
        if a.cmd == "encode-dir":
            return encode_dir(a.src, a.dst)
        if a.cmd == "decode-dir":
            return decode_dir(a.src, a.dst)
        return info(a.src)
    except (api.EncodeError, api.DecodeError, OSError, wave.Error) as e:
        print(f"error: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
