"""Independent anchor for the aligner (row a10): the optimal semi-global affine SCORE of the
oracle and of the HIP kernels equals a textbook Gotoh DP (tests/textbook_gotoh.py, pure Python, no code shared with either), and the
alignment either of them returns (one op per column) is a VALID alignment of the two sequences that reaches exactly that score.
Scores are tie-break independent; which of several optimal paths parasail 1.2.4 would return stays unpinned."""
import numpy as np
import pytest
from ngspeciesid_amd._capi import ReadSet
from textbook_gotoh import gotoh_semiglobal_score, score_of_alignment

ALPHA = "ACGT"


def _pairs(seed, n, lmin, lmax, wild=0.03):
    rng = np.random.default_rng(seed)
    qs, ts, opens = [], [], []
    for x in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        a = [ALPHA[c] for c in rng.integers(0, 4, L)]
        kind = x % 4
        if kind == 3:                                   # unrelated sequences
            b = [ALPHA[c] for c in rng.integers(0, 4, int(rng.integers(lmin, lmax + 1)))]
        else:                                           # a mutated copy, sometimes clipped / extended at the ends (free end gaps matter)
            b = []
            for c in a:
                r = rng.random()
                if r < 0.06: continue
                if r < 0.12: b.append(ALPHA[int(rng.integers(0, 4))])
                else: b.append(c)
                if rng.random() < 0.05: b += [ALPHA[int(rng.integers(0, 4))]] * int(rng.integers(1, 4))
            if kind == 1: b = b[int(rng.integers(0, 8)):]
            if kind == 2: a = a[:max(1, len(a) - int(rng.integers(0, 8)))]; b = [ALPHA[int(rng.integers(0, 4))] for _ in range(int(rng.integers(0, 6)))] + b
        if not b: b = ["A"]
        # wildcards and lower case: non-ACGT scores 0, case does not matter for the score (but '=' / 'X' compare the raw characters)
        for s in (a, b):
            for i in range(len(s)):
                r = rng.random()
                if r < wild: s[i] = "N" if r < wild / 2 else "R"
                elif r < wild + 0.02: s[i] = s[i].lower()
        qs.append("".join(a)); ts.append("".join(b)); opens.append(int(rng.integers(2, 6)))
    return qs, ts, np.asarray(opens, dtype=np.int32)


def _check(api, qs, ts, opens, match=2, mismatch=-2, ext=1):
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
    idx = np.arange(len(qs), dtype=np.uint32)
    score, ncols, nmatch, region = api.sg_align_batch(q, t, idx, idx, opens, ext, match, mismatch, 13, None)
    cscore, ops = api.sg_align_cigar_batch(q, t, idx, idx, opens, ext, match, mismatch)
    bad = []
    for i in range(len(qs)):
        want = gotoh_semiglobal_score(qs[i], ts[i], match, mismatch, int(opens[i]), ext)
        if int(score[i]) != want or int(cscore[i]) != want:
            bad.append((i, int(score[i]), int(cscore[i]), want)); continue
        got = score_of_alignment(qs[i], ts[i], ops[i], match, mismatch, int(opens[i]), ext)
        if got != want:
            bad.append((i, "path scores", got, want))
        assert len(ops[i]) == int(ncols[i]) and ops[i].count("=") == int(nmatch[i])
    assert not bad, "%d of %d pairs differ from the textbook DP (pair, got, got, want): %s" % (len(bad), len(qs), bad[:6])


def test_oracle_score_equals_textbook_gotoh(oracle):
    """5 200 random pairs of 8-70 bases (pure-Python DP: ~6 M cells) incl. wildcards, lower case, clipped ends, unrelated pairs, open 2..5"""
    qs, ts, opens = _pairs(2024, 5200, 8, 70)
    qs += ["A", "ACGT", "ACGTN", "acgtacgt", "GATTACA", "NNNN", "ACGT"]; ts += ["C", "ACGT", "NACGT", "ACGTACGT", "TTTTGATTACATTT", "NNNNNN", "TGCA"]
    opens = np.concatenate([opens, np.full(7, 3, dtype=np.int32)])
    _check(oracle, qs, ts, opens)


def test_oracle_score_other_scoring_schemes(oracle):
    """the DP, not the constants: (match, mismatch, ext) away from the reference's 2 / -2 / 1"""
    for sc in ((1, -3, 2), (4, -1, 1), (2, -2, 0), (3, -2, 3)):
        qs, ts, opens = _pairs(77 + sc[0], 300, 8, 60)
        opens = np.maximum(opens, sc[2])        # the closed form "open + (l-1) ext" assumes opening is not cheaper than extending
        _check(oracle, qs, ts, opens, match=sc[0], mismatch=sc[1], ext=sc[2])


@pytest.mark.gpu
def test_hip_score_equals_textbook_gotoh(gpu_api):
    """the same pairs through libngsid_hip.so: the packed-int16 kernel (k_align16.hip) and, with open > 16, the int32 kernel (k_align.hip)"""
    qs, ts, opens = _pairs(2024, 5200, 8, 70)
    _check(gpu_api, qs, ts, opens)
    qs, ts, opens = _pairs(9, 600, 8, 70)
    _check(gpu_api, qs, ts, opens + 15)         # opens 17..20: outside the 16-bit kernel's range -> int32 kernel
    for sc in ((1, -3, 2), (4, -1, 1), (2, -2, 0), (3, -2, 3)):
        qs, ts, opens = _pairs(77 + sc[0], 300, 8, 60)
        _check(gpu_api, qs, ts, np.maximum(opens, sc[2]), match=sc[0], mismatch=sc[1], ext=sc[2])
