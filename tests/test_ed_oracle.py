"""CPU: the oracle's edit-distance polisher aligner (ongsid_ed_align_batch) against hand-made answers and an independent
brute-force infix edit distance."""
import numpy as np
import pytest
from ngspeciesid_amd._capi import ReadSet


def infix_distance(q, t):
    """min edit distance of q against any substring of t (rows = q, free start/end in t); letters outside ACGT never match"""
    ok = set("ACGTacgt")
    prev = [0] * (len(t) + 1)
    for i, a in enumerate(q, 1):
        cur = [i] + [0] * len(t)
        for j, b in enumerate(t, 1):
            eq = a in ok and b in ok and a.upper() == b.upper()
            cur[j] = min(prev[j - 1] + (0 if eq else 1), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return min(prev)


def test_known_answers(oracle):
    q = ReadSet.from_strings(["ACGTACGTTTGA", "ACGT", "", "ACNT", "acgt"]); t = ReadSet.from_strings(["TTTACGTACGTTGATTT", "ACGT", "GGACGTGG"])
    d, span, bp = oracle.ed_align_batch(q, t, [0, 1, 2, 3, 4, 1], [0, 1, 0, 1, 1, 2], window=5, bp_windows=4)
    assert list(d) == [1, 0, 0, 1, 0, 0]
    assert list(span[0]) == [0, 11, 3, 13]          # ACGTACGT-TGA inside TTT[ACGTACGTTGA]TTT with one read-only column
    assert list(span[1]) == [0, 3, 0, 3]
    assert list(span[2]) == [-1, -1, -1, -1]        # empty query aligns nothing
    assert list(span[5]) == [0, 3, 2, 5]            # leftmost best end, exact infix
    assert bp[0].tolist() == [[0, 1, 3, 4], [2, 6, 5, 9], [8, 11, 10, 13], [-1, -1, -1, -1]]
    assert list(d[3:5]) == [1, 0]                   # N matches nothing; lower case matches upper case


def test_distance_is_the_infix_edit_distance(oracle):
    rng = np.random.default_rng(5)
    A = "ACGT"
    qs, ts = [], []
    for _ in range(120):
        T = "".join(A[x] for x in rng.integers(0, 4, int(rng.integers(1, 90))))
        a = int(rng.integers(0, len(T))); b = int(rng.integers(a, len(T) + 1))
        q = list(T[a:b])
        for k in range(len(q)):
            u = rng.random()
            if u < 0.1: q[k] = A[rng.integers(0, 4)]
            elif u < 0.15: q[k] = ""
            elif u < 0.2: q[k] = q[k] + A[rng.integers(0, 4)]
            elif u < 0.22: q[k] = "N"
        qs.append("".join(q)); ts.append(T)
    Q = ReadSet.from_strings(qs); T = ReadSet.from_strings(ts); idx = np.arange(len(qs), dtype=np.uint32)
    d, span, _ = oracle.ed_align_batch(Q, T, idx, idx)
    for i, (q, t) in enumerate(zip(qs, ts)):
        assert d[i] == infix_distance(q, t), (q, t)
        if span[i][0] >= 0:
            assert 0 <= span[i][0] <= span[i][1] < len(q) and 0 <= span[i][2] <= span[i][3] < len(t)
