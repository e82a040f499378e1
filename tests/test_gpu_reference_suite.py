"""The reference crate's own tests (src/lib.rs:991-1298, fuzz targets, bench assertion), through the
Python mirror of its API and -- compiled from tests/cpp/reference_suite.cpp -- through the C++ mirror."""
import os
import subprocess

import numpy as np
import pytest

from conftest import FIXTURE, ROOT

pytestmark = pytest.mark.gpu


def test_iterating_through_whole_file(gpu, fixture_bytes):  # src/lib.rs:998-1032
    qoa = gpu.QoaDecoder(fixture_bytes)
    assert qoa.mode() == gpu.FixedSamples(2, 44100, 2394122)
    assert abs(qoa.total_duration() - 2394122 / 44100) < 1e-9
    headers = samples = 0
    while True:
        item = qoa.next()
        if item is None:
            break
        if isinstance(item, gpu.Sample):
            samples += 1 + qoa.read_samples(1 << 20).size  # bulk drain to the next header
        else:
            assert item.num_channels == 2 and item.sample_rate == 44100
            headers += 1
            assert item.num_samples_per_channel == (5120 if headers < 468 else 3082)
    assert (headers, samples) == (468, 2394122 * 2)


def test_bench_item_count_via_plain_iteration(gpu, fixture_bytes):  # benches/qoa_benchmarks.rs:8-12
    data = fixture_bytes[: 8 + 3 * 4136]  # three frames, item by item through the FFI
    assert sum(1 for _ in gpu.QoaDecoder(data)) == 3 * 5120 * 2 + 3


def test_decode_all(gpu, fixture_bytes):  # src/lib.rs:1068-1074
    d = gpu.decode_all(fixture_bytes)
    assert (d.sample_rate, d.num_channels, d.samples.size) == (44100, 2, 2394122 * 2)


@pytest.mark.parametrize("channels", [1, 2])
def test_encode_decode_sine(gpu, channels):  # src/lib.rs:1076-1146
    sr = 44100
    n = int(sr * 0.1)
    t = np.arange(n) / sr
    cols = [np.sin(t * 440.0 * 2.0 * np.pi) * 16384.0, np.cos(t * 440.0 * 2.0 * np.pi) * 16384.0][:channels]
    samples = np.stack(cols, axis=1).astype(np.int16).reshape(-1)  # `as i16` truncates toward zero
    encoded = gpu.encode_all(samples, gpu.QoaDesc(channels, sr, n))
    assert len(encoded) > gpu.QOA_HEADER_SIZE
    d = gpu.decode_all(encoded)
    assert (d.sample_rate, d.num_channels, d.samples.size) == (sr, channels, samples.size)


def test_encoder_errors(gpu):  # src/lib.rs:1148-1183
    z = np.zeros(1000, np.int16)
    for desc, kind in ((gpu.QoaDesc(0, 44100, 1000), "InvalidChannels"), (gpu.QoaDesc(1, 0, 1000), "InvalidSampleRate"),
                       (gpu.QoaDesc(1, 44100, 0), "InvalidSamples")):
        with pytest.raises(gpu.EncodeError) as e:
            gpu.encode_all(z, desc)
        assert e.value.kind == kind


def test_round_trip_audio(gpu):  # src/lib.rs:1185-1230
    samples = np.array([((i % 200) - 100) * 100 for i in range(1000)], dtype=np.int16)
    encoded = gpu.encode_all(samples, gpu.QoaDesc(1, 44100, 1000))
    d = gpu.decode_all(encoded)
    assert (d.sample_rate, d.num_channels, d.samples.size) == (44100, 1, 1000)
    assert int(np.abs(samples.astype(int) - d.samples.astype(int)).max()) < 8000


def test_full_file_round_trip(gpu, fixture_bytes):  # src/lib.rs:1232-1259
    original = gpu.decode_all(fixture_bytes)
    desc = gpu.QoaDesc(original.num_channels, original.sample_rate, original.samples.size // original.num_channels)
    decoded = gpu.decode_all(gpu.encode_all(original.samples, desc))
    assert (decoded.num_channels, decoded.sample_rate, decoded.samples.size) == (2, 44100, original.samples.size)
    rms = float(np.sqrt(np.mean((original.samples.astype(np.float64) - decoded.samples.astype(np.float64)) ** 2)))
    assert rms < 500.0


def test_fuzz_encode_round_trip_invariants(gpu):  # fuzz/fuzz_targets/encode_round_trip.rs
    r = np.random.default_rng(2024)
    for _ in range(40):
        data = r.integers(0, 256, int(r.integers(4, 4000)), dtype=np.uint8).tobytes()
        channels = data[0] % 8 + 1
        num_i16 = (len(data) - 1) // 2
        if num_i16 < channels:
            continue
        spc = num_i16 // channels
        samples = np.frombuffer(data[1:1 + spc * channels * 2], dtype="<i2")
        d = gpu.decode_all(gpu.encode_all(samples, gpu.QoaDesc(channels, 44100, spc)))
        assert (d.num_channels, d.sample_rate, d.samples.size) == (channels, 44100, samples.size)


def test_fuzz_iter_all_never_crashes(gpu, oracle):  # fuzz/fuzz_targets/iter_all.rs
    r = np.random.default_rng(77)
    good = oracle.encode_all(r.integers(-9000, 9000, 600 * 2).astype(np.int16), 2, 44100, 600)
    for i in range(60):
        b = bytearray(good if i % 2 else b"qoaf" + r.integers(0, 256, 200, dtype=np.uint8).tobytes())
        for _ in range(3):
            b[int(r.integers(0, min(len(b), 80)))] = int(r.integers(0, 256))
        try:
            n = sum(1 for _ in gpu.QoaDecoder(bytes(b)))
            assert n >= 0
        except gpu.DecodeError:
            pass


def test_cpp_mirror_runs_the_reference_suite(gpu, tmp_path):
    exe = tmp_path / "reference_suite"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_suite.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "qoaudio_amd"), "-lqoaudio_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "qoaudio_amd")])
    r = subprocess.run([str(exe), FIXTURE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failure(s)" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("ok   test_") == 11  # the crate's 10 + the batch wrapper (HostPipeline) against per-clip calls


def test_wav_cli_round_trip(gpu, oracle, tmp_path):
    """examples/encode.rs + examples/decode.rs equivalents: WAV -> QOA -> WAV."""
    import wave
    from qoaudio_amd import cli
    from conftest import rng_pcm
    x = rng_pcm(77, 3000 * 2)
    src, mid, dst = str(tmp_path / "a.wav"), str(tmp_path / "a.qoa"), str(tmp_path / "b.wav")
    with wave.open(src, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes(x.astype("<i2").tobytes())
    assert cli.main(["encode", src, mid]) == 0
    assert open(mid, "rb").read() == oracle.encode_all(x, 2, 22050, 3000)
    assert cli.main(["decode", mid, dst]) == 0
    with wave.open(dst, "rb") as w:
        assert (w.getnchannels(), w.getframerate(), w.getnframes()) == (2, 22050, 3000)
        got = np.frombuffer(w.readframes(3000), dtype="<i2")
    _, _, want = oracle.decode_all(open(mid, "rb").read())
    assert np.array_equal(got, want)
    assert cli.main(["info", mid]) == 0
    assert cli.main(["decode", src, dst]) == 1  # a WAV is not a QOA file


def test_wav_cli_directory_batches(gpu, oracle, tmp_path):
    """encode-dir / decode-dir: a directory of WAV files of different shapes as ONE batch through the host pipeline; every
    .qoa equals the oracle's encode_all of its WAV, every decoded WAV the oracle's decode."""
    import wave
    from qoaudio_amd import cli
    from conftest import rng_pcm
    wavs, qoas, back = tmp_path / "wav", tmp_path / "qoa", tmp_path / "back"
    wavs.mkdir()
    shapes = {"a": (2, 44100, 5120 * 3 + 77), "b": (1, 8000, 333), "c": (2, 44100, 5120 * 40), "d": (6, 48000, 5120 + 1),
              "e": (1, 22050, 5120 * 70)}
    pcm = {}
    for name, (ch, rate, n) in shapes.items():
        pcm[name] = rng_pcm(len(name) + n, n * ch)
        with wave.open(str(wavs / f"{name}.wav"), "wb") as w:
            w.setnchannels(ch)
            w.setsampwidth(2)
            w.setframerate(rate)
            w.writeframes(pcm[name].astype("<i2").tobytes())
    assert cli.main(["encode-dir", str(wavs), str(qoas)]) == 0
    for name, (ch, rate, n) in shapes.items():
        assert (qoas / f"{name}.qoa").read_bytes() == oracle.encode_all(pcm[name], ch, rate, n), name
    assert cli.main(["decode-dir", str(qoas), str(back)]) == 0
    for name, (ch, rate, n) in shapes.items():
        with wave.open(str(back / f"{name}.wav"), "rb") as w:
            assert (w.getnchannels(), w.getframerate(), w.getnframes()) == (ch, rate, n)
            got = np.frombuffer(w.readframes(n), dtype="<i2")
        assert np.array_equal(got, oracle.decode_all((qoas / f"{name}.qoa").read_bytes())[2]), name
    (qoas / "broken.qoa").write_bytes((qoas / "a.qoa").read_bytes()[:-8])     # size disagrees with its header
    assert cli.main(["decode-dir", str(qoas), str(back)]) == 2


def test_c_example_round_trip(gpu, tmp_path):
    """examples/roundtrip.c: the C ABI from plain C, encode_all + decode_all on host buffers."""
    import subprocess
    from test_cpp_mirror_builds import _build_c_example
    r = subprocess.run([str(_build_c_example(tmp_path))], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "176400 samples" in r.stdout and "2 ch, 44100 Hz" in r.stdout
