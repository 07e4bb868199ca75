"""Round 5: reads beyond 16 384 bases.  The reference has no length cap in its clustering (cluster.py:239-334); this build scores, sketches and
clusters reads of up to NGSID_MAX_READ_LEN = 65 535 bases: the minimizer kernel's LONG layout (read not staged, windows scanned directly), the int32 aligner for pairs with
a sequence above 4 000 bases (a class of its own in a partitioned batch, so that the short pairs keep their int16 instances).  HIP == oracle on all of it."""
import numpy as np
import pytest
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _s(a):
    return ACGT[a].tobytes().decode()


def _noisy(rng, a, e):
    b = a.copy()
    m = rng.random(len(b)) < e * 0.5; b[m] = rng.integers(0, 4, int(m.sum()))
    b = b[rng.random(len(b)) > e * 0.25]
    ins = np.nonzero(rng.random(len(b)) < e * 0.25)[0]
    return np.insert(b, ins, rng.integers(0, 4, len(ins)))


def _q(rng, n, lo, hi):
    return (rng.integers(lo, hi, n) + 33).astype(np.uint8).tobytes().decode()


@pytest.mark.parametrize("kw", [(13, 20), (25, 30), (15, 50)])
def test_minimizers_up_to_65535_bases(gpu_api, oracle, kw):
    """one call with reads of 65 535, 40 000, 16 385 (the first length of the LONG layout), 16 384 and 700 bases, with homopolymer runs: every output == the oracle's"""
    k, w = kw
    rng = np.random.default_rng(k + 7)
    seqs, quals = [], []
    for L in (65535, 700, 40000, 16385, 16384, 23, 30001):
        a = rng.integers(0, 4, L); rep = rng.random(L) < 0.15; a[1:][rep[1:]] = a[:-1][rep[1:]]
        seqs.append(_s(a)); quals.append(_q(rng, L, 2, 45))
    seqs.append("A" * 20000 + "C" * 20000); quals.append("5" * 40000)                     # 40 kb that compress to two letters: shorter than k
    seqs.append(("ACGTTGCA" * 4000)[:30000]); quals.append("I" * 30000)                   # periodic: equal codes, leftmost minimum
    rs = ReadSet.from_strings(seqs, quals)
    got = gpu_api.hpc_minimizers(rs, k, w); exp = oracle.hpc_minimizers(rs, k, w)
    for nm, a, b in zip(["moff", "codes", "pos", "hpc_len"], got[:4], exp[:4]):
        assert np.array_equal(a, b), nm
    assert np.array_equal(got[4], exp[4], equal_nan=True)
    sc = gpu_api.score_reads(rs, k if k <= 21 else 13, 7.0); so = oracle.score_reads(rs, k if k <= 21 else 13, 7.0)
    assert np.array_equal(sc[0], so[0]) and np.array_equal(sc[1], so[1]) and np.array_equal(sc[2], so[2])


def test_long_layout_equals_the_goldens(oracle):
    """the LONG layout forced on the reference's own reads (ngsid_ctx_option minimizers_mode = 4): the golden minimizers of cluster.py:16-39"""
    import os
    from ngspeciesid_amd import runtime
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "minimizers_sample_h1.npz"))
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    for k, w in ((13, 20), (15, 50), (25, 30)):
        exp = oracle.hpc_minimizers(rs, k, w)
        api = runtime.new_api(options={"minimizers_mode": 4})
        try:
            got = api.hpc_minimizers(rs, k, w)
        finally:
            api.close()
        for nm, a, b in zip(["moff", "codes", "pos", "hpc_len"], got[:4], exp[:4]):
            assert np.array_equal(a, b), (k, w, nm)
        assert np.array_equal(got[4], exp[4], equal_nan=True)


def test_scores_of_reads_beyond_65535_bases(gpu_api, oracle):
    """the scorer has no cap (32-bit quality histogram above 65 535 bases): such reads are scored and written by the CLI, not clustered"""
    rng = np.random.default_rng(3)
    seqs = [_s(rng.integers(0, 4, L)) for L in (70000, 800, 131072)]
    quals = [_q(rng, len(s), 3, 40) for s in seqs]
    quals[2] = "5" * 131072                                                               # one character more than 65 535 times
    rs = ReadSet.from_strings(seqs, quals)
    a = gpu_api.score_reads(rs, 13, 7.0); b = oracle.score_reads(rs, 13, 7.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_align_long_pairs_equal_the_oracle(gpu_api, oracle):
    """the int32 aligner above the old cap: a 40 kb x 40 kb pair at ~10 % divergence (score beyond int16), 20 kb, long against short both ways, a 65 535-base query"""
    rng = np.random.default_rng(5)
    t40 = rng.integers(0, 4, 40000); t20 = rng.integers(0, 4, 20000); t65 = rng.integers(0, 4, 65535)
    qs = [_s(_noisy(rng, t40, 0.10)), _s(_noisy(rng, t20, 0.12)[150:]), _s(t40[1000:1300]), _s(_noisy(rng, t40, 0.05)), _s(t65), "ACGT"]
    ts = [_s(t40), _s(t20[:19800]), _s(t40), _s(t40[20000:20300]), _s(t65[100:3000]), _s(t20)]
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
    idx = np.arange(len(qs), dtype=np.uint32)
    opens = np.array([3, 2, 5, 4, 3, 2], dtype=np.int32); mids = np.array([9, 8, 10, 9, 7, 3], dtype=np.int32)
    got = gpu_api.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    exp = oracle.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    for nm, a, b in zip(["score", "ncols", "nmatch", "region"], got, exp):
        assert np.array_equal(a, b), (nm, a, b)
    assert got[0][0] > 32767


def test_a_large_batch_keeps_its_short_pairs_in_the_int16_classes(gpu_api, oracle):
    """4 300 pairs of 260 - 900 bases with nine pairs above 4 000 bases among them (long query, long target, both): the partitioned batch sends the nine through the
    int32 kernel as a class of their own; every pair == the oracle, and == the same batch through the int32 kernel alone (ngsid_ctx_option align32)"""
    from ngspeciesid_amd import runtime
    rng = np.random.default_rng(11)
    qs, ts = [], []
    for i in range(4300):
        L = int(rng.integers(260, 420)) if i % 7 else int(rng.integers(700, 900))
        a = rng.integers(0, 4, L); qs.append(_s(_noisy(rng, a, 0.08))); ts.append(_s(a))
    big = rng.integers(0, 4, 9000)
    for j, (x, y) in enumerate([(big[:5000], big[:5000]), (big[:300], big), (big, big[4000:4400]), (big[:4001], big[:700]), (big[:700], big[:4001]), (big[:4000], big[:4000]),
                                (big, big), (big[100:6000], big[:300]), (big[:350], big[2000:8000])]):
        p = 13 + 477 * j
        qs[p] = _s(_noisy(rng, x, 0.06)); ts[p] = _s(y)
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
    idx = np.arange(len(qs), dtype=np.uint32)
    opens = rng.integers(2, 6, len(qs)).astype(np.int32); mids = rng.integers(5, 12, len(qs)).astype(np.int32)
    got = gpu_api.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    exp = oracle.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    for nm, a, b in zip(["score", "ncols", "nmatch", "region"], got, exp):
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, (nm, bad[:8], a[bad[:8]], b[bad[:8]])
    api32 = runtime.new_api(options={"align32": 1})
    try:
        g32 = api32.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    finally:
        api32.close()
    for a, b in zip(got, g32):
        assert np.array_equal(a, b)


def test_a_40kb_read_clusters_like_the_oracle_says(gpu_api, oracle):
    """VERDICT r4 item 8's criterion: 40 kb and 17 kb ONT-like reads (two long templates, ~8 - 12 % error by their quality strings) among 240 amplicon reads of two
    750-base species - representatives, memberships, mapping triples and counters of ngsid_cluster_greedy == the oracle's; the long reads join THEIR templates' clusters"""
    rng = np.random.default_rng(17)
    tA = rng.integers(0, 4, 40000); tB = rng.integers(0, 4, 17000); s1 = rng.integers(0, 4, 750); s2 = rng.integers(0, 4, 750)
    recs = []
    for tpl, tag, n, e in ((tA, "A", 2, 0.09), (tB, "B", 3, 0.11), (s1, "s1", 120, 0.06), (s2, "s2", 120, 0.06)):
        for i in range(n):
            a = _noisy(rng, tpl, e)
            recs.append((tag, _s(a), _q(rng, len(a), 6, 16) if len(tpl) > 1000 else _q(rng, len(a), 9, 22)))
    a = _noisy(rng, tA, 0.02); recs.append(("A", _s(a), _q(rng, len(a), 25, 40)))           # a clean long read: founds the cluster of template A
    rs0 = ReadSet.from_strings([r[1] for r in recs], [r[2] for r in recs])
    score, err, keep = oracle.score_reads(rs0, 13, 7.0)
    order = np.argsort(-score, kind="stable")
    order = order[keep[order].astype(bool)]
    rs = ReadSet.from_strings([recs[i][1] for i in order], [recs[i][2] for i in order])
    tags = [recs[i][0] for i in order]
    prm = cluster_params(k=13, w=20, p_shared=select_p_table(13, 20))
    got = gpu_api.cluster_greedy(rs, prm); exp = oracle.cluster_greedy(rs, prm)                 # rep, hpc error rate, status, counters (mapped, aln_passed, aln_called, ...)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[2], exp[2]) and np.array_equal(got[3], exp[3]), (got[3], exp[3])
    assert np.array_equal(got[1], exp[1], equal_nan=True)
    rep = np.asarray(got[0])
    for tag in ("A", "B"):
        members = [i for i, t in enumerate(tags) if t == tag]
        assert len(members) >= 3 and len({int(rep[i]) for i in members}) == 1, (tag, [int(rep[i]) for i in members])      # one cluster per long template
    assert int(np.asarray(got[3]).max()) > 0


def test_cli_clusters_long_reads_and_names_the_consensus_limit(gpu_api, oracle, tmp_path):
    """the CLI with 20 kb reads among the reference's sample_h1 reads: without --consensus every output file == the run with the oracle as the backend (the long reads
    are clustered, not singletons); with --consensus and a cutoff that selects the long cluster the run ends with a message that names the cluster and the limit"""
    import os
    from oracle_lib import GOLD
    from ngspeciesid_amd import cli as _cli, fastpath
    from test_gpu_cli import _files
    rng = np.random.default_rng(23)
    tpl = rng.integers(0, 4, 20000)
    recs = open(os.path.join(GOLD, "sample_h1.fastq")).read().rstrip("\n").split("\n")
    for i in range(4):
        a = _noisy(rng, tpl, 0.04)
        recs += ["@long_%d" % i, _s(a), "+", _q(rng, len(a), 18, 30)]
    fq = tmp_path / "in.fastq"; fq.write_text("\n".join(recs) + "\n")
    res = {}
    for name, api in (("hip", gpu_api), ("oracle", oracle)):
        out = str(tmp_path / ("out_" + name))
        args = _cli.build_parser().parse_args(["--ont", "--fastq", str(fq), "--outfolder", out, "--t", "1"])
        os.makedirs(out, exist_ok=True)
        fastpath.main(args, api=api)
        res[name] = _files(out)
    assert sorted(res["hip"]) == sorted(res["oracle"])
    for k in res["hip"]:
        assert res["hip"][k] == res["oracle"][k], k
    rows = [l.split("\t") for l in res["hip"]["final_clusters.tsv"].decode().splitlines()]
    longs = {r[0] for r in rows if r[1].startswith("long_")}
    assert len(longs) == 1 and sum(1 for r in rows if r[0] in longs) == 4          # one cluster of the four long reads
    out = str(tmp_path / "out_cons"); os.makedirs(out, exist_ok=True)
    args = _cli.build_parser().parse_args(["--ont", "--fastq", str(fq), "--outfolder", out, "--t", "1", "--consensus", "--racon", "--abundance_ratio", "0.01", "--rc_identity_threshold", "0.9"])
    with pytest.raises(ValueError, match="POA engine forms consensus of reads up to 13107"):
        fastpath.main(args, api=gpu_api)


def test_lengths_around_the_layout_and_class_boundaries(gpu_api, oracle):
    """reads of 16 382 ... 16 387 bases (the LDS-staged layouts end at 16 384, the LONG layout starts at 16 385) at four (k, w) pairs = the register, the stored / lean and the
    two-word layouts on the short side; and a partitioned batch whose queries and targets sit at 3 998 ... 4 002 bases (the int16 classes end at 4 000): all == the oracle"""
    rng = np.random.default_rng(29)
    seqs, quals = [], []
    for L in (16382, 16383, 16384, 16385, 16386, 16387, 800, 16385):
        a = rng.integers(0, 4, L); rep = rng.random(L) < 0.2; a[1:][rep[1:]] = a[:-1][rep[1:]]
        seqs.append(_s(a)); quals.append(_q(rng, L, 2, 45))
    rs = ReadSet.from_strings(seqs, quals)
    for k, w in ((13, 20), (15, 50), (21, 21), (28, 40)):
        got = gpu_api.hpc_minimizers(rs, k, w); exp = oracle.hpc_minimizers(rs, k, w)
        for nm, a, b in zip(["moff", "codes", "pos", "hpc_len"], got[:4], exp[:4]):
            assert np.array_equal(a, b), (k, w, nm)
        assert np.array_equal(got[4], exp[4], equal_nan=True)
    qs, ts = [], []
    for i in range(4200):
        L = int(rng.integers(257, 300)); a = rng.integers(0, 4, L); qs.append(_s(_noisy(rng, a, 0.07))); ts.append(_s(a))
    base = rng.integers(0, 4, 4100); j = 0
    for ql in (3998, 3999, 4000, 4001, 4002):
        for tl in (3999, 4000, 4001):
            p = 7 + 270 * j; j += 1
            q = _noisy(rng, base, 0.03)
            q = q[:ql] if len(q) >= ql else np.concatenate([q, rng.integers(0, 4, ql - len(q))])
            qs[p] = _s(q); ts[p] = _s(base[:tl])
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
    assert int(np.diff(q.off.astype(np.int64)).max()) == 4002
    idx = np.arange(len(qs), dtype=np.uint32)
    opens = rng.integers(2, 6, len(qs)).astype(np.int32); mids = rng.integers(5, 12, len(qs)).astype(np.int32)
    got = gpu_api.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    exp = oracle.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    for nm, a, b in zip(["score", "ncols", "nmatch", "region"], got, exp):
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, (nm, bad[:8], a[bad[:8]], b[bad[:8]])
