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


@pytest.mark.parametrize("mu", [14.0, 17.0])
def test_trimmed_drafts_and_polish_equal_the_amplicon_on_noisy_reads(oracle, mu):
    """coverage-trimmed tile consensuses (poa trim = 1): the draft of 1 500 noisy 750-base reads is within 2 edits of the amplicon with NO
    end slack, and one polishing iteration returns the amplicon exactly; a backbone with junk overhangs is polished back to it as well."""
    sp, rd, rs = make_set(1500, L=750, mu=mu, seed=11)
    truth = sp[0].tobytes().decode()
    draft = oracle.poa_consensus(rs, [0, rs.n], poa_params(tile_depth=8, band=128, trim=1))[0]
    assert edit_distance(draft, truth) <= 2
    raw = oracle.poa_consensus(rs, [0, rs.n], poa_params(tile_depth=8, band=128, trim=0))[0]
    assert edit_distance(draft, truth) <= edit_distance(raw, truth)
    for bb in (draft, "CC" + truth + "GCCATAAATG"):
        pol, used = oracle.polish(ReadSet.from_strings([bb]), rs, [0, rs.n], polish_params(iters=1, k=13, w=20, tile_depth=8, band=128, trim=2))
        assert pol[0] == truth


def test_rc_identity_equals_the_reference(oracle):
    """(a15) consensus.highest_aln_identity (consensus.py:129-145): golden produced by the reference function itself (oracle/make_golden.py,
    aligner behind the parasail shim = this oracle's scalar aligner, so the alignment stage is self-referential; the identity arithmetic is not)."""
    import os
    from oracle_lib import GOLD
    g = np.load(os.path.join(GOLD, "align_sample_h1.npz"))
    comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}
    n = len(g["identity"])
    qs = [g["q"][int(g["q_off"][i]):int(g["q_off"][i + 1])].tobytes().decode() for i in range(n)]
    ts = [g["t"][int(g["t_off"][i]):int(g["t_off"][i + 1])].tobytes().decode() for i in range(n)]
    rcs = ["".join(chr(comp.get(ord(c), 78)) for c in reversed(t)) for t in ts]
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts + rcs)
    score, ncols, nmatch, _ = oracle.sg_align_batch(q, t, list(range(n)) * 2, list(range(2 * n)), 3, 1, 2, -2, 13, None)
    ident = np.maximum(nmatch[:n] / ncols[:n].astype(np.float64), nmatch[n:] / ncols[n:].astype(np.float64))
    assert np.array_equal(ident, g["identity"])


@pytest.mark.parametrize("L", [507, 1003])
def test_short_tail_window_is_merged(oracle, L):
    """an amplicon a few bases longer than a multiple of the 500-base window: racon would polish the 7- / 3-base tail from the reads that carry
    insertions there (only layers of >= 10 bases enter a window); here the short tail belongs to the window before it and the result is exact"""
    sp = synth.make_species(1, L, 0.15, seed=21)
    while len(sp[0]) % 500 >= 40 or len(sp[0]) % 500 == 0:             # indels of the species generator move the length: insist on a short tail
        sp = [np.concatenate([sp[0], sp[0][:3]])] if len(sp[0]) % 500 == 0 else [sp[0][:len(sp[0]) - 1]]
    rd = synth.make_reads(sp, 600, mu=15.0, seed=31)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    truth = sp[0].tobytes().decode()
    assert 0 < len(truth) % 500 < 40
    pol, used = oracle.polish(ReadSet.from_strings([truth]), rs, [0, rs.n], polish_params(iters=2, k=13, w=20, tile_depth=8, band=0, trim=2))
    assert pol[0] == truth
    noisy_bb = truth[:200] + truth[203:len(truth) - 2] + "A" + truth[-2:]
    pol, used = oracle.polish(ReadSet.from_strings([noisy_bb]), rs, [0, rs.n], polish_params(iters=2, k=13, w=20, tile_depth=8, band=0, trim=2))
    assert pol[0] == truth


def _engine(oracle, e):
    import ctypes
    return oracle.lib.ongsid_debug_poa_engine(ctypes.c_int32(e))


def test_rank_engine_equals_node_engine(oracle):
    """oracle/ngsid_oracle_poa_rank.c (the tile engine on a rank-ordered graph, the representation of csrc/k_poa.hip) returns the bytes of the node-indexed
    engine that defines the semantics: drafts (local mode) and polishing windows (global / semi-global layers), several depths incl. one graph per group,
    noisy reads, graphs that are closed early (small node capacity), narrow bands (band-edge redo), unit weights, both strands, coverage output."""
    cases = []
    sp, rd, rs = make_set(90, L=400, mu=13.0, seed=5)
    for D in (0, 4, 6):
        for trim in (0, 1):
            cases.append(("draft", rs, dict(tile_depth=D, band=64, trim=trim)))
    cases.append(("draft", rs, dict(tile_depth=0, band=64, node_cap=18)))                 # graph closed early, restarted
    cases.append(("draft", rs, dict(tile_depth=5, band=128, node_cap=20, trim=1)))
    sp2, rd2, rs2 = make_set(60, L=900, mu=11.0, seed=8)
    cases.append(("draft", rs2, dict(tile_depth=6, band=64, trim=1)))                     # noisy: band-edge redo, far predecessors
    cases.append(("draft", ReadSet(rs2.seq, None, rs2.off), dict(tile_depth=0, band=128)))   # unit weights: ties between equal weights
    sp3, rd3, rs3 = make_set(120, L=1100, mu=14.0, seed=9, rc_fraction=0.5)
    res = {}
    for e in (0, 1):
        old = _engine(oracle, e)
        try:
            out = []
            for kind, r, kw in cases:
                cc = oracle.poa_consensus_cov(r, [0, r.n // 2, r.n], poa_params(**kw))
                out.append(([c for c, _ in cc], [v.tolist() for _, v in cc]))
            for D, trim, it in ((6, 2, 2), (0, 1, 1), (4, 0, 2)):
                bb = rs3.get(int(np.nonzero(rd3["strand"].numpy() == 0)[0][0]))[0]
                out.append(oracle.polish(ReadSet.from_strings([bb, sp3[0].tobytes().decode()[:-3]]), rs3, [0, rs3.n // 2, rs3.n], polish_params(iters=it, k=13, w=20, tile_depth=D, band=0, trim=trim, stop_when_stable=0)))
            res[e] = out
        finally:
            _engine(oracle, old)
    assert len(res[0]) == len(res[1])
    for a, b in zip(res[0], res[1]):
        assert a[0] == b[0]
        assert list(a[1]) == list(b[1])


def test_reference_order_rules_of_round_5(oracle):
    """round 5: the reference-order mode of the polisher - ONE graph per window, layers in first-position order, no trimming - degrades exact
    backbones at ~2 000 layers per window (junction insertions).  The two racon rules this build replaces (overlap-span clipping, sub-graph alignment: oracle-only
    switches, ongsid_debug_polish_rules bits 0 / 1) do not remove that; creating no source / sink nodes from layer ends (bit 2, a probe) does, which names the cause:
    global alignment to a DAG may start at ANY in-edge-less node, so one layer's leading insertion captures the layers after it.  One cluster, one iteration from the
    exact amplicon; the ten-cluster / three-iteration record is profiles/r05_reference_order.json (tools/r05_reference_order.py), whose summary is pinned here too."""
    import ctypes, json, os
    from ngspeciesid_amd.hostutil import subset_reads
    from util_seq import edit_distance
    sp = synth.make_species(5, 750, 0.15, seed=1)
    rd = synth.make_reads([sp[2]], 2000, mu=17.0, seed=11)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    score, err, keep = oracle.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    sub = subset_reads(rs, idx)
    truth = sp[2].tobytes().decode()
    prm = polish_params(iters=1, k=13, w=20, tile_depth=0, band=0, node_cap=160, trim=1, stop_when_stable=0)
    got = {}
    try:
        for rules in (0, 7):
            oracle.lib.ongsid_debug_polish_rules(ctypes.c_int32(rules))
            got[rules] = oracle.polish(ReadSet.from_strings([truth]), sub, [0, sub.n], prm)[0][0]
    finally:
        oracle.lib.ongsid_debug_polish_rules(ctypes.c_int32(0))
    assert edit_distance(got[0], truth) >= 1 and len(got[0]) > len(truth)          # the round-4 restatement inserts at the window junction
    assert got[7] == truth                                                          # both racon rules + no source / sink nodes from layer ends: a fixed point
    rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_reference_order.json")))
    S = rec["summary_over_10_synthetic_clusters"]
    assert S["0"]["clusters_where_polishing_increases_the_distance_of_the_draft"] == 9 and S["3"]["clusters_where_polishing_increases_the_distance_of_the_draft"] == 7
    assert S["7"]["clusters_where_polishing_increases_the_distance_of_the_draft"] == 0 and S["7"]["clusters_where_the_exact_amplicon_is_not_a_fixed_point"] == 0
    assert S["7"]["sum_of_edits_after_3_iterations_from_draft"] <= S["7"]["sum_of_edits_of_the_drafts"]


def test_overlap_span_clipping_polishes_a_trimmed_backbone(oracle):
    """round 5, ngsid_polish_params_t.aln_mode = 3: of the read -> backbone alignment only the columns between the first and the last run of 15 equal columns count
    (minimap2's chain ends in front of racon's edlib call).  Reads that carry primers at both ends against a backbone that has been primer-trimmed: the whole-read
    aligner (mode 1) forces the primer bases into the backbone's ends and the polished sequence grows by a primer; with clipping and trim 3 (tile consensuses trimmed as in
    the shipped mode, but the tile that ends a window keeps the ends of its backbone where no layer reaches them - racon's rule for NGS windows) the trimmed backbone comes
    back as the amplicon body, and an untrimmed backbone is returned as it is under either mode."""
    from ngspeciesid_amd import barcode_trimmer
    tails = barcode_trimmer.get_universal_tails()
    body = synth.make_species(1, 520, 0.15, seed=8)[0].tobytes().decode()
    amp = tails["1_F_fw"] + body + tails["2_R_fw"]
    rd = synth.make_reads([np.frombuffer(amp.encode(), dtype=np.uint8)], 300, mu=16.0, seed=3, rc_fraction=0.5)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    def run(bb, mode, trim):
        return oracle.polish(ReadSet.from_strings([bb]), rs, [0, rs.n], polish_params(iters=2, k=13, w=20, tile_depth=6, band=0, trim=trim, aln_mode=mode, stop_when_stable=0))[0][0]
    assert run(body, 3, 3) == body
    grown = run(body, 1, 2)
    assert len(grown) > len(body) + 15 and body in grown or edit_distance(grown, body) >= 15          # mode 1 pulls a primer back in
    assert run(amp, 3, 3) == amp and run(amp, 1, 2) == amp


def test_small_units_run_as_one_graph(oracle):
    """round 6, single_below (oracle run_unit): a group below the threshold == the plain single-graph order with room for ten times its first sequence (tile_depth 0,
    node_cap NGSID_POA_SINGLE_NODE_CAP), a group at or above it == the tiled hierarchy; the polisher applies the same rule per window."""
    sp, rd, rs = make_set(90, L=300, mu=13.0, seed=31)
    goff = [0, 20, 60, 90]                                  # 20, 40 and 30 reads
    prm = poa_params(tile_depth=4, band=0, trim=1, single_below=32)
    got = oracle.poa_consensus(rs, goff, prm)
    one = oracle.poa_consensus(rs, goff, poa_params(tile_depth=0, band=0, trim=1, node_cap=160))
    tiled = oracle.poa_consensus(rs, goff, poa_params(tile_depth=4, band=0, trim=1))
    assert got == [one[0], tiled[1], one[2]]
    assert oracle.poa_consensus(rs, goff, poa_params(tile_depth=4, band=0, trim=1, single_below=0)) == tiled
    bb = ReadSet.from_strings([rs.get(0)[0]])
    a = oracle.polish(bb, rs, [0, 30], polish_params(iters=2, tile_depth=4, band=0, trim=2, single_below=32))[0]
    b = oracle.polish(bb, rs, [0, 30], polish_params(iters=2, tile_depth=0, band=0, trim=2, node_cap=160))[0]
    assert a == b
