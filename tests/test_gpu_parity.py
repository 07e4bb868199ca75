"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the golden fixtures.  Bit-exact integer results."""
import os, json
import numpy as np
import pytest
from oracle_lib import GOLD
from ngspeciesid_amd._capi import ReadSet, cluster_params
from test_oracle_golden import _acc_rank, _load

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _dump(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)


@pytest.mark.parametrize("tag", ["sample_h1", "synth200"])
@pytest.mark.parametrize("kw", [(13, 20), (15, 50), (10, 100), (21, 21)])
def test_minimizers(gpu_api, oracle, tag, kw):
    g = _load("minimizers_%s.npz" % tag)
    k, w = kw
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    got = gpu_api.hpc_minimizers(rs, k, w)
    exp = oracle.hpc_minimizers(rs, k, w)
    names = ["moff", "codes", "pos", "hpc_len"]
    for nm, a, b in zip(names, got[:4], exp[:4]):
        if not np.array_equal(a, b):
            _dump("fail_minimizers_%s_%d_%d" % (tag, k, w), **{"got_" + n: x for n, x in zip(names, got[:4])}, **{"exp_" + n: x for n, x in zip(names, exp[:4])})
        assert np.array_equal(a, b), nm
    assert np.array_equal(got[4], exp[4], equal_nan=True)          # HPC error rate: same op order -> bit-exact doubles
    assert np.array_equal(got[1], g["codes_%d_%d" % (k, w)])         # and against the reference's own output


@pytest.mark.parametrize("kw", [(25, 30), (30, 35), (22, 22), (32, 40)])
def test_minimizers_k_above_21(gpu_api, oracle, kw):
    g = _load("minimizers_widek.npz")
    k, w = kw
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    got = gpu_api.hpc_minimizers(rs, k, w); exp = oracle.hpc_minimizers(rs, k, w)
    for a, b in zip(got[:4], exp[:4]):
        assert np.array_equal(a, b)
    assert np.array_equal(got[4], exp[4], equal_nan=True)
    assert np.array_equal(got[1], g["rank_%d_%d" % (k, w)]) and np.array_equal(got[2], g["pos_%d_%d" % (k, w)])        # the reference's own k-mers


@pytest.mark.parametrize("kw", [(13, 20), (25, 30)])
def test_minimizers_at_the_maximum_read_length(gpu_api, oracle, kw):
    """the one-wave-per-read layout needs 11 (k <= 21) / 19 (k > 21) LDS bytes per base - more than a CU has at NGSID_MAX_READ_LEN = 16 384.
    Long reads run the LEAN layout (codes rebuilt from the HPC letters on the fly); same output as the oracle at 16 384, 15 000 and 9 000 bases,
    next to short reads in the same call."""
    k, w = kw
    rng = np.random.default_rng(k)
    seqs, quals = [], []
    for L in (16384, 15000, 9000, 700, 16384):
        a = rng.integers(0, 4, L); rep = rng.random(L) < 0.15; a[1:][rep[1:]] = a[:-1][rep[1:]]          # some homopolymer runs
        seqs.append("".join("ACGT"[x] for x in a)); quals.append("".join(chr(33 + int(x)) for x in rng.integers(2, 45, L)))
    rs = ReadSet.from_strings(seqs, quals)
    got = gpu_api.hpc_minimizers(rs, k, w); exp = oracle.hpc_minimizers(rs, k, w)
    for nm, a, b in zip(["moff", "codes", "pos", "hpc_len"], got[:4], exp[:4]):
        assert np.array_equal(a, b), nm
    assert np.array_equal(got[4], exp[4], equal_nan=True)


@pytest.mark.parametrize("kw", [(13, 20), (15, 50), (30, 35)])
def test_minimizers_lean_layout_equals_the_goldens(kw):
    """the lean layout forced on ordinary reads (ngsid_ctx_option minimizers_lean=1, a context of its own): the reference's own codes / ranks"""
    import ctypes as C
    from ngspeciesid_amd import runtime
    k, w = kw
    api = runtime.new_api(options={"minimizers_lean": 1})
    try:
        if k <= 21:
            g = _load("minimizers_sample_h1.npz"); rs = ReadSet(g["seq"], g["qual"], g["off"])
            got = api.hpc_minimizers(rs, k, w)
            assert np.array_equal(got[1], g["codes_%d_%d" % (k, w)])
        else:
            g = _load("minimizers_widek.npz"); rs = ReadSet(g["seq"], g["qual"], g["off"])
            got = api.hpc_minimizers(rs, k, w)
            assert np.array_equal(got[1], g["rank_%d_%d" % (k, w)]) and np.array_equal(got[2], g["pos_%d_%d" % (k, w)])
    finally:
        api.close()


@pytest.mark.parametrize("kw", [(13, 20), (21, 21), (13, 40), (12, 43), (5, 5), (16, 31)])
def test_minimizer_layouts_agree(gpu_api, oracle, kw):
    """round 3: three layouts of the minimizer kernel - window minima in registers (the default for windows of up to 16 k-mers), the stored sparse table, the lean
    long-read layout - give the oracle's output, on the reference's goldens where they exist and on reads around the chunk boundaries of the register layout"""
    from ngspeciesid_amd import runtime
    k, w = kw
    g = _load("minimizers_sample_h1.npz")
    rng = np.random.default_rng(k * 100 + w)
    extra = []
    for L in list(range(k, k + 4)) + [w - 1, w, w + 1, 57 + k - 1, 58 + k - 1, 64 + k, 2 * 57 + k, 750, 3000]:
        if L < 1: continue
        a = rng.integers(0, 4, L); extra.append(("".join("ACGT"[x] for x in a), "".join(chr(33 + int(x)) for x in rng.integers(2, 45, L))))
    extra.append(("AC" * 200, "I" * 400)); extra.append(("A" * 300, "5" * 300)); extra.append(("ACGTN" * 60, "+" * 300))
    base = ReadSet(g["seq"], g["qual"], g["off"])
    rs = ReadSet.from_strings([base.get(i)[0] for i in range(base.n)] + [e[0] for e in extra], [base.get(i)[1] for i in range(base.n)] + [e[1] for e in extra])
    exp = oracle.hpc_minimizers(rs, k, w)
    for mode in (0, 1, 2, 3):
        api = gpu_api if mode == 0 else runtime.new_api(options={"minimizers_mode": mode})
        try:
            got = api.hpc_minimizers(rs, k, w)
        finally:
            if mode: api.close()
        for nm, a, b in zip(["moff", "codes", "pos", "hpc_len"], got[:4], exp[:4]):
            assert np.array_equal(a, b), (mode, nm)
        assert np.array_equal(got[4], exp[4], equal_nan=True)


@pytest.mark.parametrize("kw", [(13, 20), (25, 30), (32, 40)])
def test_minimizers_in_many_small_chunks(gpu_api, oracle, kw):
    """round 5: the minimizer kernel writes into a bounded sparse scratch, one chunk of reads at a time, and a gather appends every chunk to the compact CSR
    (ngsid_minimizers_csr).  With chunks of ~2 000 bases (option minimizers_chunk_bases; the default is 256 M) the golden read set takes dozens of chunks: same CSR, same
    HPC lengths and error rates as one chunk and as the oracle - also for k > 21, where the second code word travels through the chunks and the rename pass runs on the
    compact arrays; and the clustering built on it equals the oracle's."""
    from ngspeciesid_amd import runtime
    from ngspeciesid_amd._capi import cluster_params
    from ngspeciesid_amd.ptable import select_p_table
    k, w = kw
    g = _load("minimizers_sample_h1.npz")
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    exp = oracle.hpc_minimizers(rs, k, w)
    small = runtime.new_api(options={"minimizers_chunk_bases": 2000})
    try:
        for api in (gpu_api, small):
            got = api.hpc_minimizers(rs, k, w)
            for nm, a, b in zip(["moff", "codes", "pos", "hpc_len"], got[:4], exp[:4]):
                assert np.array_equal(a, b), nm
            assert np.array_equal(got[4], exp[4], equal_nan=True)
        if k <= 30:
            prm = cluster_params(k=k, w=w, p_shared=select_p_table(k, w))
            a = small.cluster_greedy(rs, prm); b = oracle.cluster_greedy(rs, prm)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    finally:
        small.close()


def _rand_pairs(rng, n, lmin, lmax, sim=True):
    qs, ts = [], []
    for _ in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        a = rng.integers(0, 4, L)
        if sim:
            b = a.copy()
            m = rng.random(L) < 0.08
            b[m] = rng.integers(0, 4, int(m.sum()))
            keep = rng.random(L) > 0.04
            b = b[keep]
            ins = rng.integers(0, 4, int(rng.integers(0, 6)))
            cut = int(rng.integers(0, len(b) + 1))
            b = np.concatenate([b[:cut], ins, b[cut:]])
            if rng.random() < 0.3:
                b = b[int(rng.integers(0, 20)):]
            if rng.random() < 0.3:
                a = a[:len(a) - int(rng.integers(0, 20))]
        else:
            b = rng.integers(0, 4, int(rng.integers(lmin, lmax + 1)))
        qs.append("".join("ACGT"[x] for x in a)); ts.append("".join("ACGT"[x] for x in b))
    return qs, ts


@pytest.mark.parametrize("case", [("tiny", 40, 1, 40), ("rpl4", 60, 100, 256), ("rpl8", 40, 300, 512), ("rpl12", 60, 600, 768),
                                  ("rpl16", 30, 800, 1024), ("strips", 12, 1100, 2600), ("random", 40, 50, 400), ("many", 9000, 60, 140)])
def test_align_vs_oracle(gpu_api, oracle, case):
    name, n, lmin, lmax = case
    import zlib
    seed = zlib.crc32(name.encode()) % 1000          # stable across processes (hash() is randomised by PYTHONHASHSEED)
    rng = np.random.default_rng(seed)
    qs, ts = _rand_pairs(rng, n, lmin, lmax, sim=(name != "random"))
    if name == "tiny":
        qs += ["A", "ACGT", "ACGTN", "acgtacgt", "GATTACA"]; ts += ["C", "ACGT", "NACGT", "ACGTACGT", "TTTTGATTACATTT"]
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
    idx = np.arange(len(qs), dtype=np.uint32)
    opens = rng.integers(2, 6, len(qs)).astype(np.int32); mids = rng.integers(-1, 14, len(qs)).astype(np.int32)
    got = gpu_api.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    exp = oracle.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids)
    for nm, a, b in zip(["score", "ncols", "nmatch", "region"], got, exp):
        if not np.array_equal(a, b):
            _dump("fail_align_%s" % name, got=np.stack(got), exp=np.stack(exp))
            bad = np.nonzero(a != b)[0]
            raise AssertionError("seed %d: %s differs at pairs %s: got %s exp %s (len q %s t %s)" % (seed, nm, bad[:8], a[bad[:8]], b[bad[:8]], [len(qs[i]) for i in bad[:8]], [len(ts[i]) for i in bad[:8]]))


def test_align_golden(gpu_api):
    g = _load("align_sample_h1.npz")
    q = ReadSet(g["q"], None, g["q_off"]); t = ReadSet(g["t"], None, g["t_off"])
    idx = np.arange(q.n, dtype=np.uint32)
    score, ncols, nmatch, region = gpu_api.sg_align_batch(q, t, idx, idx, g["open"], 1, 2, -2, 13, g["match_id"])
    assert np.array_equal(score, g["score"]) and np.array_equal(ncols, g["n_cols"]) and np.array_equal(nmatch, g["n_match"])
    qlen = np.diff(g["q_off"].astype(np.int64))
    assert np.array_equal(region / qlen.astype(np.float64), g["ratio"])


@pytest.mark.parametrize("tag", ["sample_h1", "synth2k_d15", "synth600_d10_q14", "synth300_ccs", "synth1200_k25", "synth1200_k30"])
def test_cluster_t1(gpu_api, oracle, tag):
    g = _load("cluster_%s.npz" % tag)
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    prm = cluster_params(k=int(g["k"]), w=int(g["w"]), p_shared=g["p_table"])
    ar = _acc_rank([str(a) for a in g["acc"]])
    rep, herr, st, cnt = gpu_api.cluster_greedy(rs, prm, acc_rank=ar)
    orep, oherr, ost, ocnt = oracle.cluster_greedy(rs, prm, acc_rank=ar)
    if not (np.array_equal(rep, orep) and np.array_equal(st, ost)):
        _dump("fail_cluster_%s" % tag, rep=rep, orep=orep, st=st, ost=ost, cnt=cnt, ocnt=ocnt)
        bad = np.nonzero((rep != orep) | (st != ost))[0]
        raise AssertionError("cluster %s: %d reads differ, first %s got rep %s st %s exp rep %s st %s; counters %s vs %s" %
                             (tag, len(bad), bad[:10], rep[bad[:10]], st[bad[:10]], orep[bad[:10]], ost[bad[:10]], cnt, ocnt))
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(herr, oherr, equal_nan=True)
    assert np.array_equal(rep, g["t1_rep_of"])                      # = the reference's own membership
    assert [int(c) for c in cnt[:3]] == [int(c) for c in g["t1_counters"]]


def test_cluster_seeded_merge_round(gpu_api, oracle):
    """merge-round semantics (cluster.py:221-223,243-248): lower-batch representatives seed the index, higher-batch ones are re-clustered."""
    g = _load("cluster_synth2k_d15.npz")
    rs_all = ReadSet(g["seq"], g["qual"], g["off"])
    n = rs_all.n
    # take 400 reads, pretend the first half is batch 1 and the second batch 2, everything is a representative with a known error rate
    sel = np.arange(0, 400)
    seqs = [rs_all.get(i)[0] for i in sel]; quals = [rs_all.get(i)[1] for i in sel]
    rs = ReadSet.from_strings(seqs, quals)
    prm = cluster_params(k=13, w=20, p_shared=g["p_table"])
    prev = np.where(sel < 200, 1, 2).astype(np.int32)
    _, _, _, _, he = oracle.hpc_minimizers(rs, 13, 20)
    ar = _acc_rank([str(a) for a in g["acc"][sel]])
    got = gpu_api.cluster_greedy(rs, prm, acc_rank=ar, prev_batch=prev, known_err=he)
    exp = oracle.cluster_greedy(rs, prm, acc_rank=ar, prev_batch=prev, known_err=he)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b, equal_nan=True)


def test_sg_align_cigar_vs_oracle(gpu_api, oracle):
    """(boundary 8b) the alignment columns ('=XID', free end gaps included) from the HIP aligner == the oracle's, and consistent with the
    statistics of ngsid_sg_align_batch; the parasail-shaped front returns the same CIGAR as the test shim the goldens were produced with."""
    rng = np.random.default_rng(12)
    qs, ts = _rand_pairs(rng, 60, 30, 900)
    qs += ["ACGT", "A", "", "GATTACAGATTACA", "acgtNNacgt"]; ts += ["ACGT", "C", "ACG", "CCCCGATTACAGATTACATTTT", "ACGTACGT"]
    q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
    idx = np.arange(len(qs)); opens = rng.integers(2, 6, len(qs))
    s1, o1 = gpu_api.sg_align_cigar_batch(q, t, idx, idx, opens)
    s2, o2 = oracle.sg_align_cigar_batch(q, t, idx, idx, opens)
    assert np.array_equal(s1, s2) and o1 == o2
    sc, ncols, nmatch, _ = gpu_api.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, None)
    assert np.array_equal(sc, s1) and [len(o) for o in o1] == ncols.tolist() and [o.count("=") for o in o1] == nmatch.tolist()
    for a, b, o in zip(qs, ts, o1):
        assert o.count("=") + o.count("X") + o.count("I") == len(a) and o.count("=") + o.count("X") + o.count("D") == len(b)
    from ngspeciesid_amd import parasail_like
    r = parasail_like.sg_trace_scan_16(qs[0], ts[0], int(opens[0]), 1, parasail_like.matrix_create("ACGT", 2, -2), api=gpu_api)
    assert r.score == s1[0] and not r.saturated and r.cigar.decode == parasail_like._Cigar(o1[0]).decode


def test_poa_consensus_coverage_vs_oracle(gpu_api, oracle):
    from ngspeciesid_amd import synth
    from ngspeciesid_amd._capi import poa_params
    sp = synth.make_species(3, 400, 0.15, seed=4)
    seqs, quals, off = [], [], [0]
    for g in range(3):
        rd = synth.make_reads([sp[g]], 50, mu=15.0, seed=70 + g)
        seqs.append(rd["seq"].numpy()); quals.append(rd["qual"].numpy()); off += list(off[-1] + rd["off"].numpy()[1:])
    rs = ReadSet(np.concatenate(seqs), np.concatenate(quals), np.array(off, dtype=np.uint64))
    for prm in (poa_params(tile_depth=8, band=64, trim=1), poa_params(tile_depth=0, band=128, node_cap=64), poa_params(tile_depth=8, band=128, trim=0)):
        a = gpu_api.poa_consensus_cov(rs, [0, 50, 100, 150], prm); b = oracle.poa_consensus_cov(rs, [0, 50, 100, 150], prm)
        assert [x[0] for x in a] == [x[0] for x in b]
        for x, y in zip(a, b):
            assert np.array_equal(x[1], y[1])
        assert [x[0] for x in a] == gpu_api.poa_consensus(rs, [0, 50, 100, 150], prm)
        assert all(int(x[1].max()) <= 50 and len(x[1]) == len(x[0]) for x in a)
