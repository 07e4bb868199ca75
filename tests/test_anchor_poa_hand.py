"""Hand-derived partial-order-alignment vectors (rows a14 / a17).

spoa and racon are absent from /root/reference and from the image, so the POA half of the oracle is pinned on synthetic ground truth only.
These cases narrow what that leaves open: each one is small enough that the graph, the edge weights and the heaviest-bundle walk can be
worked out on paper from the published algorithm (spoa `-l 0 -r 0 -g -2`: m=+5 n=-4 linear gap -2, local alignment in file order, FASTQ edge
weight += (q[i-1]-33)+(q[i]-33), per node the in-edge of maximum weight, node score = that weight + predecessor score, start from the node of
maximum score, branch completion to a sink, backtrack).  The derivation is in the comment of every case.  They run in the mode that claims to
BE spoa's order: tile_depth = 0 (one graph, file order), trim = 0, band (64 columns) >= every sequence, so no build choice of DESIGN.md
section 2 takes part.  None of the cases depends on a tie-break between equal scores or equal weights.

CPU: the oracle.  GPU: the same vectors through libngsid_hip.so.
"""
import numpy as np
import pytest
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL, POA_GLOBAL

#      0         1
#      0123456789012345
X = "TGACTGCTACGCGTAT"          # no base equals its neighbour, no 3-mer occurs twice; edits below: position 6 (C -> A), a C inserted before / the A deleted at position 8
HI, MID, LO = "I", "5", "+"     # phred 40 / 20 / 10  ->  edge weight per read and edge: 80 / 40 / 20


def _sub(s, i, c): return s[:i] + c + s[i + 1:]


CASES = {
    # (1) majority substitution.  R1 = X seeds a chain of 16 nodes, every edge 40+40 = 80.  R2 = R3 = X with C->A at position 6: the best local
    #     alignment is end to end (15 matches, one mismatch: 75 - 4 = 71 > 45 for the longer flank alone), so 'A' becomes a node beside node 6 with
    #     edges 5->A and A->7 of weight 80 + 80 = 160, while 5->6 and 6->7 stay at 80 and every shared edge grows to 240.  Node 7 takes its heavier
    #     in-edge (160, from the new node); the last node has the largest score and is a sink; the backtrack passes through the new node.  Consensus = R2.
    "majority_substitution": ([X, _sub(X, 6, "A"), _sub(X, 6, "A")], [HI * 16] * 3, _sub(X, 6, "A")),
    # (2) quality outvotes the count.  R1 (phred 40) says C at position 6; R2, R3 (phred 10) say A.  Edges into / out of C: 80; of A: 20 + 20 = 40.
    #     Node 7 takes the edge from C.  Consensus = R1 although two of three reads disagree (spoa weighs edges by base quality).
    "quality_outvotes_count": ([X, _sub(X, 6, "A"), _sub(X, 6, "A")], [HI * 16, LO * 16, LO * 16], X),
    # (3) the same vote decided by count: FASTA input (no qualities) = weight 1 per base, 2 per edge and read.  Two reads with C (edges 4), three with A (6).
    "fasta_unit_weights": ([X, _sub(X, 6, "A"), _sub(X, 6, "A"), X, _sub(X, 6, "A")], None, _sub(X, 6, "A")),
    # (4) majority insertion.  R2 = R3 = X with a C between positions 7 (T) and 8 (A): 16 matches and one gap (80 - 2 = 78).  The new node C sits
    #     between nodes 7 and 8: edges 7->C and C->8 weigh 160, the direct edge 7->8 stays at 80 (R1 only).  Node 8 takes the edge from C.  Consensus = R2.
    "majority_insertion": ([X, X[:8] + "C" + X[8:], X[:8] + "C" + X[8:]], [HI * 16, HI * 17, HI * 17], X[:8] + "C" + X[8:]),
    # (5) majority deletion.  R2 = R3 = X without position 8 (A): the alignment skips node 8, a new edge 7->9 of weight 160 appears; 7->8 and 8->9
    #     stay at 80.  Node 9 takes the edge from 7.  Consensus = X without that base.
    "majority_deletion": ([X, X[:8] + X[9:], X[:8] + X[9:]], [HI * 16, HI * 15, HI * 15], X[:8] + X[9:]),
    # (6) minority edits lose.  Three reads X, one with the deletion of (5), one with the insertion of (4): edges of the X path 240 / 320, the
    #     deviating edges 80.  Consensus = X.
    "minority_indels": ([X, X[:8] + X[9:], X, X[:8] + "C" + X[8:], X], [HI * 16, HI * 15, HI * 16, HI * 17, HI * 16], X),
}

# (7) branch completion.  R1 = A + "CAT" with A = 28 bases, phred 20 (edges 40).  R2 = R3 = R4 = "GG" + "CAT", phred 40: locally only "CAT" aligns
#     (15; A holds no G, and "CA" elsewhere gives 10 at most), so "GG" becomes a new branch g1->g2->C.  Weights: g1->g2 and g2->C 3 x 80 = 240,
#     C->A and A->T 40 + 240 = 280, the chain of A 40 each, a28->C 40.
#     Scores (a source scores -1): a_k = 40 (k - 1) - 1, so a28 = 1079; g2 = 239; node C takes its HEAVIER in-edge (240 from g2, not 40 from a28):
#     C = 240 + 239 = 479, then 759 and 1039 at the sink.  The maximum (1079) is at a28, which is not a sink -> branch completion: the other
#     predecessors of a28's successors (g2) are discarded, the scores below a28 are recomputed (C = 40 + 1079 = 1119, then 1399, 1679) and the walk
#     ends at the sink.  Consensus = all of R1 - neither A alone (no completion) nor "GGCAT" (backtrack from the best sink).
A28 = "TAACTTACTTAATACCACCAACACACAC"      # no G at all and no second "CAT": every other local alignment of "GGCAT" scores <= 10 (checked by exhaustive Smith-Waterman when the
                                           # vector was made - a first attempt with G's in A aligned "G-G-CA" for 16 and the oracle rightly disagreed with the derivation)
assert len(A28) == 28 and "G" not in A28 and "CAT" not in A28
CASES["branch_completion"] = ([A28 + "CAT", "GGCAT", "GGCAT", "GGCAT"], [MID * 31, HI * 5, HI * 5, HI * 5], A28 + "CAT")


def _run(api, name, mode=POA_LOCAL, **kw):
    seqs, quals, want = CASES[name]
    rs = ReadSet.from_strings(seqs, quals)
    got = api.poa_consensus(rs, [0, len(seqs)], poa_params(mode=mode, tile_depth=0, band=64, trim=0, **kw))[0]
    assert got == want, "%s: got %s, derived %s" % (name, got, want)


@pytest.mark.parametrize("name", sorted(CASES))
def test_hand_derived_spoa_consensus_oracle(oracle, name):
    _run(oracle, name)


@pytest.mark.parametrize("name", ["majority_substitution", "quality_outvotes_count", "majority_insertion", "majority_deletion", "minority_indels"])
def test_hand_derived_global_racon_scores_oracle(oracle, name):
    """racon's window scores (+3 / -5 / -4, global): a mismatch (-5) beats two gaps (-8), one gap (-4) beats what it would cost to misalign; the
    graphs, weights and votes are those of the local cases (all reads span the whole graph)."""
    _run(oracle, name, mode=POA_GLOBAL, match=3, mismatch=-5, gap=-4)


def _polish_case():
    # racon: the backbone enters the window graph FIRST with weight 0; four reads (phred 40) agree on the amplicon and differ from the backbone
    # at one base: the backbone's node keeps edges of weight 0, the reads' node gets 4 x 80.  One window (82 < 500 bases), reads span it.
    truth = "GATTACAGGCTTAACCGTAGCTAGGCTAACGTTAGCCATGCAATCGGATCCTAGGTACCATTGACGGATACCTGAAGTCAGTC"
    bb = _sub(truth, 40, "A" if truth[40] != "A" else "C")
    return truth, bb


def test_hand_derived_polish_oracle(oracle):
    truth, bb = _polish_case()
    rs = ReadSet.from_strings([truth] * 4, [HI * len(truth)] * 4)
    got, used = oracle.polish(ReadSet.from_strings([bb]), rs, [0, 4], polish_params(iters=1, k=13, w=20, tile_depth=0, band=128, trim=0))
    assert got[0] == truth and int(used[0]) == 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hand_derived_spoa_consensus_hip(gpu_api, name):
    _run(gpu_api, name)
    if name != "fasta_unit_weights" and name != "branch_completion":
        _run(gpu_api, name, mode=POA_GLOBAL, match=3, mismatch=-5, gap=-4)


@pytest.mark.gpu
def test_hand_derived_polish_hip(gpu_api):
    truth, bb = _polish_case()
    rs = ReadSet.from_strings([truth] * 4, [HI * len(truth)] * 4)
    got, used = gpu_api.polish(ReadSet.from_strings([bb]), rs, [0, 4], polish_params(iters=1, k=13, w=20, tile_depth=0, band=128, trim=0))
    assert got[0] == truth and int(used[0]) == 4
