"""CPU: the consensus / polish oracle against synthetic ground truth (spoa/racon themselves are absent: parity unpinned)."""
import numpy as np
import pytest
from util_seq import edit_distance
from ngspeciesid_amd import synth
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL, POA_GLOBAL


def make_set(n, L=500, mu=17.0, seed=3, rc_fraction=0.0, nsp=1):
    sp = synth.make_species(nsp, L, 0.15, seed=seed)
    rd = synth.make_reads(sp, n, mu=mu, seed=seed + 10, rc_fraction=rc_fraction)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    return sp, rd, rs


def interior_ed(cons, truth, slack=4):
    """edit distance ignoring up to `slack` overhanging bases at either end (heaviest-bundle end artefacts)."""
    best = None
    for a in range(0, slack + 1):
        for b in range(0, slack + 1):
            c = cons[a:len(cons) - b] if b else cons[a:]
            d = edit_distance(c, truth)
            best = d if best is None else min(best, d)
    return best


def test_identical_reads_return_the_read(oracle):
    s = "ACGTTGCATGCATGCCGATAGCTAGCTAGGATCGATCGATTTAGCGCGATATCGCGATCGATCGGGATATATCGCGC"
    rs = ReadSet.from_strings([s] * 5, ["I" * len(s)] * 5)
    for mode in (POA_LOCAL, POA_GLOBAL):
        assert oracle.poa_consensus(rs, [0, 5], poa_params(mode=mode, band=64))[0] == s
    assert oracle.poa_consensus(ReadSet.from_strings([s], ["I" * len(s)]), [0, 1], poa_params())[0] == s


@pytest.mark.parametrize("D", [0, 8])
def test_spoa_consensus_recovers_amplicon(oracle, D):
    sp, rd, rs = make_set(60, L=500)
    truth = sp[0].tobytes().decode()
    c = oracle.poa_consensus(rs, [0, rs.n], poa_params(tile_depth=D, band=128))[0]
    assert edit_distance(rs.get(0)[0], truth) > 10
    assert interior_ed(c, truth) == 0


def test_polish_keeps_or_improves(oracle):
    sp, rd, rs = make_set(80, L=600, rc_fraction=0.5)      # both strands: the polisher must orient the reads
    truth = sp[0].tobytes().decode()
    draft = rs.get(int(np.nonzero(rd["strand"].numpy() == 0)[0][0]))[0]          # a raw forward read as backbone
    d0 = edit_distance(draft, truth)
    out, used = oracle.polish(ReadSet.from_strings([draft]), rs, [0, rs.n], polish_params(iters=2, tile_depth=8, band=128))
    assert used[0] >= 70
    assert interior_ed(out[0], truth) <= 1 < d0


def test_multiple_groups_and_empty(oracle):
    sp, rd, rs = make_set(40, L=300, nsp=2)
    spc = rd["species"].numpy()
    order = np.argsort(spc, kind="stable")
    seqs = [rs.get(i)[0] for i in order]; quals = [rs.get(i)[1] for i in order]
    rs2 = ReadSet.from_strings(seqs, quals)
    n0 = int((spc == 0).sum())
    cons = oracle.poa_consensus(rs2, [0, n0, n0, rs2.n], poa_params(tile_depth=8, band=128))
    assert cons[1] == ""                                          # empty group -> empty consensus
    assert interior_ed(cons[0], sp[0].tobytes().decode()) <= 1 and interior_ed(cons[2], sp[1].tobytes().decode()) <= 1
