import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from qoaudio_amd import _lib
from oracle import oracle as O
sys.path.insert(0, 'tests')
from conftest import rng_pcm

def gpu_slice(state, smp, generic=0):
    L = _lib.lib()
    st = np.asarray(state, np.int32); s = np.asarray(smp, np.int16)
    tot = np.zeros(16, np.uint64); fst = np.zeros(16, np.uint64); sl = np.zeros(16, np.uint64); es = np.zeros((16, 8), np.int32)
    rc = L.qoa_debug_encode_slice(st.ctypes.data, s.ctypes.data, len(s), generic, tot.ctypes.data, fst.ctypes.data, sl.ctypes.data, es.ctypes.data)
    assert rc == 0
    return tot, fst, sl, es

x = rng_pcm(7919 * 1 + 5119, 5119, "white")
# find the first differing slice with the product
import qoaudio_amd as q
got = np.frombuffer(q.encode_all(x, q.QoaDesc(1, 44100, 5119)), np.uint8)
want = np.frombuffer(O.encode_all(x, 1, 44100, 5119), np.uint8)
d = np.nonzero(got != want)[0]
k = (d[0] - 32) // 8
print("first diff byte", d[0], "slice", k)
e = O.OracleEncoder(1, 44100, k * 20)
e.encode(x[:k * 20])
(hist, w), = e.lms_state()
print("entry state", hist, w)
smp = x[k * 20:(k + 1) * 20]
s, sf, st, totals = O.encode_slice(hist, w, smp)
print("oracle: sf", sf, hex(s))
for generic in (0, 1):
    tot, fst, sl, es = gpu_slice(hist + w, smp, generic)
    print("generic", generic, "tot equal:", list(tot) == totals)
    for i in range(16):
        if int(tot[i]) != totals[i]:
            print("  sf", i, "gpu", int(tot[i]), "oracle", totals[i], hex(int(sl[i])))
    # winner by key
    keys = [(int(tot[i]), int(fst[i]), i) for i in range(16)]
    print("  gpu argmin", min(keys)[2], hex(int(sl[min(keys)[2]])))
print("slice bytes got", got[32 + k * 8: 40 + k * 8].tobytes().hex(), "want", want[32 + k * 8:40 + k * 8].tobytes().hex())
