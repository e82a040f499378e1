"""CPU baseline legs of bench.py, run in a process that never touches the GPU.

    python -m oracle.cpu_bench <clips.npy> <sample_rate> <budget_seconds> [--no-all-cores] [--no-digests]

QOA_ORACLE_LIB=liboracle_native.so in the environment times the -march=native build instead (rebuilt first when
`--rebuild` is given: -march=native must be compiled on the machine it runs on).

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  bench.py copies a bounded sample of the very clips the GPU
encoded to a .npy file and starts this module as a child process; the child times the oracle's ref_style encoder
(the reference's algorithm including its rank-sorted early exits, src/lib.rs:516-593) on one thread, then on all
host cores (one clip per forked worker -- forking is safe here, there is no HIP context in this process), and
prints one JSON object.  The sha256 of every encoded clip goes back so the parent can compare bytes with the GPU's.
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

_CLIPS = None
_RATE = 0


def _encode(i):
    from oracle import oracle as O
    pcm = _CLIPS[i % _CLIPS.shape[0]]
    return len(O.encode_all(pcm.reshape(-1), pcm.shape[1], _RATE, pcm.shape[0], O.REF_STYLE))


def main(argv):
    global _CLIPS, _RATE
    path, _RATE, budget = argv[0], int(argv[1]), float(argv[2])
    all_cores = "--no-all-cores" not in argv
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if "--rebuild" in argv:
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        name = os.environ.get("QOA_ORACLE_LIB", "liboracle.so")
        subprocess.check_call(["make", "-s", "-B", "-C", here, name])
    from oracle import oracle as O
    O.lib()
    _CLIPS = np.load(path, mmap_mode="r")  # [n, samples, channels] int16
    n_avail, samples, ch = _CLIPS.shape
    digests = []
    t0 = time.perf_counter()
    for i in range(n_avail):
        enc = O.encode_all(np.ascontiguousarray(_CLIPS[i]).reshape(-1), ch, _RATE, samples, O.REF_STYLE)
        digests.append(hashlib.sha256(enc).hexdigest())
        if time.perf_counter() - t0 > budget:
            break
    dt = time.perf_counter() - t0
    n = len(digests)
    res = {"one_thread": {"value": n * samples * ch / dt, "clips": n, "seconds": dt, "samples": n * samples * ch},
           "sha256": digests}
    if all_cores:
        cores = os.cpu_count() or 1
        jobs = max(cores, min(4 * cores, 64))
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_encode, range(cores))  # warm: page the clips in, load the library in every worker
            t0 = time.perf_counter()
            pool.map(_encode, range(jobs))
            dta = time.perf_counter() - t0
        res["all_cores"] = {"value": jobs * samples * ch / dta, "cores": cores, "clips": jobs, "seconds": dta}
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1:])
