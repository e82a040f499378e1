"""CPU check: the C++ mirror header compiles against the C ABI and links with the built library."""
import os
import subprocess

from conftest import ROOT


def test_reference_suite_compiles_and_links(tmp_path):
    exe = tmp_path / "reference_suite"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_suite.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "qoaudio_amd"), "-lqoaudio_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "qoaudio_amd")])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout


def test_c_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "qoaudio_amd.h"\nint main(void){ qoa_desc d = {2, 44100, 10}; return (int)qoa_encoded_size(d.channels, d.samples) == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(tmp_path / "t"), "-L", os.path.join(ROOT, "qoaudio_amd"), "-lqoaudio_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "qoaudio_amd")])
    assert subprocess.run([str(tmp_path / "t")]).returncode == 0


def _build_c_example(tmp_path):
    exe = tmp_path / "roundtrip"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "roundtrip.c"), "-o", str(exe), "-L", os.path.join(ROOT, "qoaudio_amd"),
                           "-lqoaudio_amd", "-Wl,-rpath," + os.path.join(ROOT, "qoaudio_amd"), "-lm"])
    return exe


def test_c_example_builds(tmp_path):
    _build_c_example(tmp_path)
