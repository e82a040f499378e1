import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(1, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURE = os.path.join(GOLDEN, "julien_baker_sprained_ankle.qoa")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- fuzz seed -------------------------------------------------------------------------------------------------------------
# Every randomised GPU test draws from np.random.default_rng([FUZZ_SEED, case]).  FUZZ_SEED changes with every change of
# the tree -- so the driver's round-end run, the only execution the builder does not control, explores inputs no earlier
# run has seen -- and is printed in the header and with every failing test; QOA_FUZZ_SEED=<n> reproduces a run.  The
# snapshot that travels to the GPU box has no .git, so the seed is the digest of the sources themselves (what a commit id
# is a digest of): kernels, generators, host code, the oracle and the tests.
_FUZZ_TREE = ("qoaudio_amd", "include", "oracle", "tests", "tools/gen_slice_asm.py", "tools/gen_decode_asm.py", "bench.py")
_FUZZ_EXT = (".py", ".hip", ".cpp", ".h", ".hpp", ".inc", ".c")


def _tree_digest():
    import hashlib
    h = hashlib.sha1()
    files = []
    for top in _FUZZ_TREE:
        path = os.path.join(ROOT, top)
        if os.path.isfile(path):
            files.append(path)
            continue
        for d, dirs, names in os.walk(path):
            dirs[:] = sorted(x for x in dirs if x not in ("__pycache__", "_ref", "golden", "bin"))
            files += [os.path.join(d, n) for n in sorted(names) if n.endswith(_FUZZ_EXT)]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _fuzz_seed():
    env = os.environ.get("QOA_FUZZ_SEED")
    if env:
        return int(env, 0), "QOA_FUZZ_SEED"
    return int(_tree_digest()[:8], 16), "digest of the source tree"


FUZZ_SEED, FUZZ_SEED_FROM = _fuzz_seed()


def fuzz_rng(case):
    """The generator of fuzz case `case` (an int or a tuple of ints) under this run's seed."""
    key = [FUZZ_SEED] + (list(case) if isinstance(case, (tuple, list)) else [int(case)])
    return np.random.default_rng(key)


def pytest_report_header(config):
    return f"qoa fuzz seed: {FUZZ_SEED} ({FUZZ_SEED_FROM}); reproduce with QOA_FUZZ_SEED={FUZZ_SEED}"


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    # (-q drops the header: the seed must still be in the log of the driver's `pytest -x -q -m gpu`)
    terminalreporter.write_line(f"qoa fuzz seed: {FUZZ_SEED} ({FUZZ_SEED_FROM}), QOA_FUZZ_SCALE="
                                f"{os.environ.get('QOA_FUZZ_SCALE', '(default)')}")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.when == "call" and rep.failed and "fuzz" in item.nodeid:
        rep.sections.append(("qoa fuzz seed", f"QOA_FUZZ_SEED={FUZZ_SEED} QOA_FUZZ_SCALE={os.environ.get('QOA_FUZZ_SCALE', '(default)')} "
                                              f"reproduces this run ({item.nodeid})"))


def pytest_sessionstart(session):
    """A fresh checkout has no libqoaudio_amd.so (it is git-ignored): build it once for the suite.  hipcc
    cross-compiles for gfx950 without a GPU.  (The product itself never builds or falls back: qoaudio_amd._lib
    raises if the library is missing.)"""
    import subprocess
    so = os.path.join(ROOT, "qoaudio_amd", "libqoaudio_amd.so")
    srcs = [os.path.join(ROOT, "qoaudio_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "qoaudio_amd", "csrc"))
            if f.endswith((".hip", ".cpp", ".h", ".inc"))] + [os.path.join(ROOT, "include", "qoaudio_amd.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "qoaudio_amd", "csrc")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def fixture_bytes():
    with open(FIXTURE, "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def fixture_pcm(oracle, fixture_bytes):
    ch, sr, pcm = oracle.decode_all(fixture_bytes)
    assert (ch, sr, pcm.size) == (2, 44100, 2394122 * 2)
    return pcm


@pytest.fixture(scope="session")
def fixture_walk(oracle, fixture_bytes):
    """(slices, boundaries) of the fixture's own decode trajectory: tests/fixture_walk.py."""
    import fixture_walk as W
    return W.walk(oracle, fixture_bytes)


@pytest.fixture(scope="session")
def gpu():
    """The product package on a real device; fails loudly if the HIP extension is not there."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import qoaudio_amd
    from qoaudio_amd import _lib
    _lib.lib()
    return qoaudio_amd


def rng_pcm(seed, n, kind="mix"):
    """Deterministic int16 test signals (numpy only; values are shared by GPU path and oracle)."""
    r = np.random.default_rng(seed)
    if kind == "white":
        return r.integers(-32768, 32768, n, dtype=np.int64).astype(np.int16)
    if kind == "loud":  # +-full-scale alternation with noise: exercises qoa_div's wrapping case
        x = np.where(np.arange(n) % 2 == 0, 32767, -32768) + r.integers(-3, 4, n)
        return np.clip(x, -32768, 32767).astype(np.int16)
    if kind == "silence":
        return np.zeros(n, dtype=np.int16)
    if kind == "square":
        return (np.where((np.arange(n) // 37) % 2 == 0, 12000, -12000)).astype(np.int16)
    t = np.arange(n)
    x = 6000 * np.sin(t * 0.031) + 3000 * np.sin(t * 0.173 + 1.0) + 900 * np.sin(t * 0.71) + r.normal(0, 300, n)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)
