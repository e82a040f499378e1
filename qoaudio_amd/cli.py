"""WAV <-> QOA command line, the counterpart of the reference's examples/encode.rs and examples/decode.rs
(SURVEY 8(f) N3).  16-bit PCM WAV only, 1-8 channels, like the reference example (examples/encode.rs:32-95).

    python -m qoaudio_amd.cli encode in.wav out.qoa
    python -m qoaudio_amd.cli decode in.qoa out.wav
    python -m qoaudio_amd.cli info in.qoa
"""
import argparse
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
    for name in ("encode", "decode"):
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
        return info(a.src)
    except (api.EncodeError, api.DecodeError, OSError, wave.Error) as e:
        print(f"error: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
