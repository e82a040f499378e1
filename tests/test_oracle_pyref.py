"""Differential test of the C oracle against an INDEPENDENT straight-line restatement of the reference's encoder
(tests/pyref_qoa.py, pure Python, written from src/lib.rs by a different route).  The reference's fixture pins the oracle
only where its content reaches (oracle/README.md: no slice of the KAT-1 prefix has a penalty-decided winner or a tied
total rank); these inputs are chosen to sit exactly where it does not:

  penalty   loud tonal material drives sum(w^2) past 0x8ff << 18, the weights penalty (src/lib.rs:814-817) enters the ranks
            and decides winners;
  ties      tiny-amplitude integers: many candidates with equal first-sample ranks (the stable insertion sort of :536-541
            orders them) and, now and then, equal total ranks (the strict `<` of :587 decides);
  wrap      +-full-scale alternation: residual * reciprocal leaves i32 at scalefactor 0/1 (qoa_div, :613-617);
  mix/white ordinary and noisy material, incl. short tails and frame boundaries.
"""
import numpy as np
import pytest

import pyref_qoa as P
from conftest import rng_pcm


def _signal(kind, n, seed):
    r = np.random.default_rng(seed)
    t = np.arange(n)
    if kind == "penalty":   # a loud two-resonance AR(4) process: the LMS grows its weights until sum(w^2) crosses 0x8ff << 18
        a = np.convolve([1, -2 * 0.9 * np.cos(0.4), 0.81], [1, -2 * 0.9 * np.cos(1.2), 0.81])
        e, x = r.normal(0, 1, n), np.zeros(n)
        for i in range(n):
            x[i] = e[i] - sum(a[k] * x[i - k] for k in range(1, 5) if i - k >= 0)
        x = x / np.abs(x).max() * 30000
    elif kind == "ties":
        x = r.integers(-3, 4, n) * (1 + (t // 700) % 3)  # tiny integers: small errors, many equal ranks
    elif kind == "wrap":
        x = np.where(t % 2 == 0, 32767, -32768) + r.integers(-2, 3, n)
    elif kind == "white":
        x = r.integers(-32768, 32768, n)
    else:
        return rng_pcm(seed, n, "mix")
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


CASES = [(ch, kind) for ch in range(1, 9) for kind in ("penalty", "ties", "wrap", "mix")] + [(1, "white"), (2, "white"), (8, "white")]


@pytest.mark.parametrize("channels,kind", CASES)
def test_oracle_equals_independent_python_restatement(oracle, channels, kind):
    # lengths: a short tail, one row, and (for 1-2 channels) a frame boundary with a 1-sample second frame
    lengths = [45, 20] + ([5121] if channels <= 2 and kind in ("penalty", "mix") else [260])
    for n in lengths:
        x = _signal(kind, n * channels, 1 + channels % 2)  # (seeds 1 and 2: both reach the penalty in the "penalty" class)
        want = P.encode_all(x, channels, 44100, n)
        assert oracle.encode_all(x, channels, 44100, n, oracle.REF_STYLE) == want
        assert oracle.encode_all(x, channels, 44100, n, oracle.EXHAUSTIVE) == want


KAT2_STATE = ([-2655, -1631, -965, -806], [-5363, 11763, -15085, 16783])  # a penalty-DECIDED slice
KAT2_PCM = [-857, -665, -532, -205, 97, 453, 1182, 2249, 3636, 4729, 5942, 7621, 8992, 9889, 10639, 11477, 12145, 12013, 11809, 11685]


def test_slice_search_from_states_at_the_penalty_threshold(oracle):
    """Slice-level differential test from entry states scattered around the penalty threshold (the KAT-2 state, perturbed),
    where the penalty term flips winners; both restatements must agree on slice word, scalefactor and end state, and the
    set must contain penalty-decided winners (else it proves nothing the stream tests do not)."""
    r = np.random.default_rng(2024)
    decided = 0
    for k in range(400):
        h = [int(v) for v in np.clip(np.array(KAT2_STATE[0]) + r.integers(-3000, 3000, 4), -32768, 32767)]
        w = [int(v) for v in np.array(KAT2_STATE[1]) + r.integers(-2500, 2500, 4) + (k % 5) * np.array([-300, 300, -300, 300])]
        base = np.array(KAT2_PCM) if k % 2 == 0 else np.cumsum(r.integers(-900, 900, 20)) + h[3]
        pcm = np.clip(base + r.integers(-60, 60, 20), -32768, 32767).astype(np.int16)[: 20 if k % 7 else 13]
        got = P.encode_slice(P.Lms(h, w), [int(v) for v in pcm], 0, len(pcm), 1)
        for search in (oracle.REF_STYLE, oracle.EXHAUSTIVE):
            s, sf, (bh, bw), _ = oracle.encode_slice(h, w, pcm, search)
            assert (s, sf, bh, bw) == (got[0], got[1], got[2].history, got[2].weights), (k, search)
        e, p, fe, fp = oracle.slice_rank_terms(h, w, pcm)
        win0 = min(range(16), key=lambda i: (e[i], fe[i], i))
        decided += win0 != got[1]
    assert decided >= 20, decided


def test_slices_with_tied_total_ranks(oracle):
    """Tiny-amplitude slices give small integer errors, so now and then two scalefactors reach the SAME minimal total rank:
    then the first-sample rank, then the scalefactor, decide (strict `<` at src/lib.rs:587 in rank-sorted order).  Both
    restatements must pick the same winner on every slice, and the set must contain such ties."""
    r = np.random.default_rng(77)
    tied = 0
    for k in range(40000):
        amp = (3, 6, 9)[k % 3]
        pcm = r.integers(-amp, amp + 1, 20).astype(np.int16)[: 20 if k % 11 else 7]
        h = [int(v) for v in r.integers(-amp, amp + 1, 4)]
        w = [0, 0, -8192, 16384] if k % 2 else [int(v) for v in np.array([0, 0, -8192, 16384]) + r.integers(-40, 41, 4)]
        s, sf, (bh, bw), tot = oracle.encode_slice(h, w, pcm, oracle.REF_STYLE)
        if sum(1 for v in tot if v == min(tot)) > 1:
            tied += 1
        elif k % 40:
            continue  # (the pure-Python side is the slow one: run it on every tie and on one in 40 of the rest)
        got = P.encode_slice(P.Lms(h, w), [int(v) for v in pcm], 0, len(pcm), 1)
        assert (s, sf, bh, bw) == (got[0], got[1], got[2].history, got[2].weights), k
        s2, sf2, _, _ = oracle.encode_slice(h, w, pcm, oracle.EXHAUSTIVE)
        assert (s2, sf2) == (s, sf), k
    assert tied >= 10, tied


def test_the_inputs_reach_what_they_are_meant_to_reach(oracle):
    """The penalty class must have penalty-DECIDED winners, the tie class tied totals, the wrap class wrapping divides --
    otherwise the test above would prove nothing beyond what the fixture already pins."""
    def walk(x, n):
        h, w = [0, 0, 0, 0], [0, 0, -8192, 16384]
        decided = ties = pen = 0
        for s in range(0, n, 20):
            seg = x[s:s + 20]
            e, p, fe, fp = oracle.slice_rank_terms(h, w, seg)
            tot = [e[i] + p[i] for i in range(16)]
            win = min(range(16), key=lambda i: (tot[i], fe[i] + fp[i], i))
            win0 = min(range(16), key=lambda i: (e[i], fe[i], i))
            decided += win != win0
            pen += p[win] > 0
            ties += sum(1 for i in range(16) if tot[i] == tot[win]) > 1
            _, sf, (h, w), _ = oracle.encode_slice(h, w, seg, oracle.REF_STYLE)
            assert sf == win
        return decided, ties, pen

    _, _, pen = walk(_signal("penalty", 5121, 2), 5121)
    assert pen >= 20, pen  # (penalty-DECIDED winners: test_slice_search_from_states_at_the_penalty_threshold)
    # ties: equal FIRST-sample ranks (what the stable insertion sort of :536-541 orders) are common in the class ...
    h, w = [0, 0, 0, 0], [0, 0, -8192, 16384]
    x, first_ties = _signal("ties", 5121, 9), 0
    for s in range(0, 5121, 20):
        e, p, fe, fp = oracle.slice_rank_terms(h, w, x[s:s + 20])
        first_ties += len(set(fe)) < 16
        _, _, (h, w), _ = oracle.encode_slice(h, w, x[s:s + 20], oracle.REF_STYLE)
    assert first_ties >= 50, first_ties
    # ... equal TOTAL ranks at the winner are rare: test_slices_with_tied_total_ranks collects them
    # wrap: the i32 product res * 65536 leaves i32 for |res| >= 32768 at scalefactor 0
    assert P.qoa_div(40000, 0) != 40000 and oracle.lib().oq_div(40000, 0) == P.qoa_div(40000, 0)
    assert P.qoa_div(-65535, 1) == oracle.lib().oq_div(-65535, 1)
