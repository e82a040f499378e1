"""The reference fixture's own decode trajectory, slice by slice (test helper, CPU oracle only).

The fixture (tests/golden/julien_baker_sprained_ankle.qoa, a reference-held vector) was written by an encoder other than
this repository's.  Walking it yields, for every one of its 239,414 slices, the predictor state the slice was decoded FROM
(seeded by each frame header's LMS block, src/lib.rs:271-281, then carried by src/lib.rs:300-318) and the 20 samples it decodes
to.  Two whole-file pins are built on that walk (tests/test_oracle_golden.py, mirrored on the device in
tests/test_gpu_parity.py):
  * decoder chain: the state at the end of frame k must equal the LMS block stored in frame k+1's header;
  * teacher-forced encoder: encode_slice from that entry state over those samples must reproduce the fixture's slice.
"""
import struct

import numpy as np


def frames(fixture_bytes):
    """Yield (frame_index, channels, samples_per_channel, [(history, weights) per channel], slice_bytes_offset, rows)."""
    pos, k = 8, 0
    while pos < len(fixture_bytes):
        hdr = int.from_bytes(fixture_bytes[pos:pos + 8], "big")
        ch, nspc, fsize = hdr >> 56, (hdr >> 16) & 0xffff, hdr & 0xffff
        lms = []
        for c in range(ch):
            b = fixture_bytes[pos + 8 + 16 * c: pos + 24 + 16 * c]
            v = struct.unpack(">8h", b)
            lms.append((list(v[:4]), list(v[4:])))
        rows = (fsize - 8 - 16 * ch) // (8 * ch)
        yield k, ch, nspc, lms, pos + 8 + 16 * ch, rows
        pos += fsize
        k += 1


def walk(oracle, fixture_bytes):
    """Decode the whole fixture slice by slice with oq_decode_slice.  Returns (slices, boundaries):
    slices     = list of dicts {frame, row, ch, entry: (h, w), pcm: int16[n_valid], word: fixture's u64, n: n_valid}
    boundaries = list of (frame k, channel, end-of-frame state (h, w), header LMS of frame k+1 (h, w))."""
    slices, boundaries = [], []
    prev_end = None
    for k, ch, nspc, lms, off, rows in frames(fixture_bytes):
        if prev_end is not None:
            for c in range(ch):
                boundaries.append((k - 1, c, prev_end[c], lms[c]))
        state = [(list(h), list(w)) for h, w in lms]
        for r in range(rows):
            n_valid = min(20, nspc - 20 * r)
            for c in range(ch):
                o = off + 8 * (r * ch + c)
                word = int.from_bytes(fixture_bytes[o:o + 8], "big")
                pcm, after = oracle.decode_slice(state[c][0], state[c][1], word)
                slices.append({"frame": k, "row": r, "ch": c, "entry": state[c], "pcm": pcm[:n_valid].copy(),
                               "word": word, "n": n_valid})
                state[c] = after
        prev_end = state
    return slices, boundaries


def penalty_free_argmin(oracle, entry, pcm):
    """argmin over the 16 scalefactors of (sum err^2, first-sample err^2, sf): the winner of src/lib.rs:516-593 with the
    weights-penalty term (:814-817, :576) left out -- what the fixture's older encoder computes."""
    e, _p, fe, _fp = oracle.slice_rank_terms(entry[0], entry[1], pcm)
    return min(range(16), key=lambda i: (e[i], fe[i], i))
