"""CPU: (f4) primer / universal-tail trimming.  The library's bit-parallel infix locator (csrc/host_io.hip) against the oracle's full-matrix DP
(edlib itself is absent: parity unpinned, anchored on the reference's call site barcode_trimmer.py:34-60) and hand-made cases; remove_barcodes
against a plain re-statement of the reference's cut rule."""
import argparse
import numpy as np
import pytest
from ngspeciesid_amd import barcode_trimmer as bt
from ngspeciesid_amd import runtime


def _both(oracle, q, t, k, iupac=True):
    a = bt.infix_locate(q, t, k, iupac)
    b = bt.infix_locate(q, t, k, iupac, lib=oracle.lib, prefix="ongsid_")
    assert a == b, (q, t, k, a, b)
    return a


def test_hand_made_locations(oracle):
    assert _both(oracle, "ACGT", "TTTACGTTT", 0) == (0, 3, 6)
    assert _both(oracle, "ACGT", "TTTACTTTT", 0) is None
    assert _both(oracle, "ACGT", "TTTACTTTT", 1)[0] == 1
    assert _both(oracle, "ACGT", "ACGTACGT", 0) == (0, 0, 3)                      # first end position wins
    assert _both(oracle, "ARGT", "TTAGGTTT", 0) == (0, 2, 5)                      # R = A or G
    assert _both(oracle, "ARGT", "TTACGTTT", 0) is None and _both(oracle, "ARGT", "TTACGTTT", 0, False) is None
    assert _both(oracle, "ANNT", "GGACCTGG", 0) == (0, 2, 5) and _both(oracle, "ANNT", "GGACCTGG", 0, False) is None
    assert _both(oracle, "acgt", "TTACGTTT", 0) is None                            # case-sensitive, like edlib
    assert _both(oracle, "AAAA", "CCCC", 2) is None and _both(oracle, "", "ACGT", 3) is None and _both(oracle, "ACGT", "", 3) is None
    # smallest start among the optimal alignments ending at the first best end: a leading mismatch rather than a shifted insertion
    assert _both(oracle, "TACGT", "GGGCACGTGG", 1) == (1, 3, 7)


def test_random_against_the_dp(oracle):
    rng = np.random.default_rng(4)
    alpha = "ACGT"; iup = "ACGTMRWSYKVHDBN"
    for trial in range(1500):
        n = int(rng.integers(1, 90 if trial % 7 == 0 else 30)); m = int(rng.integers(1, 200))
        t = "".join(alpha[x] for x in rng.integers(0, 4, m))
        if trial % 3 == 0 and m > n + 4:                     # plant a noisy copy
            p0 = int(rng.integers(0, m - n)); q = list(t[p0:p0 + n])
            for _ in range(int(rng.integers(0, 4))):
                x = int(rng.integers(0, len(q))); op = int(rng.integers(0, 3))
                if op == 0: q[x] = alpha[int(rng.integers(0, 4))]
                elif op == 1 and len(q) > 1: del q[x]
                else: q.insert(x, alpha[int(rng.integers(0, 4))])
            q = "".join(q)
        else:
            q = "".join((iup if trial % 5 == 0 else alpha)[x] for x in rng.integers(0, 15 if trial % 5 == 0 else 4, n))
        _both(oracle, q, t, int(rng.integers(0, 6)), iupac=bool(trial % 2))


def test_remove_barcodes_cut_rule():
    tails = bt.get_universal_tails()
    assert tails["1_F_rc"] == "GCAATATCAGCACCAACAGAAA" and tails["2_R_fw"] == "GAAGATAGAGCGACAGGCAAGT"
    rng = np.random.default_rng(2)
    body = "".join("ACGT"[x] for x in rng.integers(0, 4, 600))
    args = argparse.Namespace(trim_window=150, primer_max_ed=2)
    center = "GG" + tails["1_F_fw"] + body + tails["2_R_fw"] + "TT"
    centers = [[10, 0, center, []], [5, 1, body, []]]
    assert bt.remove_barcodes(centers, tails, args) is True
    # start cut = the primer's last index (its last base stays), end cut = the primer's first base (barcode_trimmer.py:84-98)
    assert centers[0][2] == center[2 + 22 - 1:len(center) - 2 - 22] and centers[1][2] == body
    assert bt.remove_barcodes(centers, tails, args) is False or len(centers[0][2]) >= len(body)
    short = [[3, 2, tails["1_F_fw"] + body[:30], []]]               # shorter than two windows: the window is half the sequence (26 bases)
    bt.remove_barcodes(short, tails, argparse.Namespace(trim_window=150, primer_max_ed=2))
    assert short[0][2] == "C" + body[:30]
    tiny = [[3, 3, tails["1_F_fw"] + "ACGTACGTAC", []]]             # window of 16 bases: the 22-base primer cannot be found in it - nothing is cut, like the reference
    assert bt.remove_barcodes(tiny, tails, argparse.Namespace(trim_window=150, primer_max_ed=2)) is False


def test_read_barcodes(tmp_path):
    p = tmp_path / "primers.fa"; p.write_text(">ITS1\nTCCGTAGGTGAACCTGCGG\n>lsu\nggtccgtgtttcaagacgg\n")
    b = bt.read_barcodes(str(p))
    assert b["ITS1_fw"] == "TCCGTAGGTGAACCTGCGG" and b["ITS1_rc"] == "CCGCAGGTTCACCTACGGA" and b["lsu_fw"] == "ggtccgtgtttcaagacgg" and b["lsu_rc"] == "CCGTCTTGAAACACGGACC"
