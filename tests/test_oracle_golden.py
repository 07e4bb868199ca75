"""Pin the CPU oracle against outputs of the reference's own Python (tests/golden/*, made by oracle/make_golden.py)."""
import os
import numpy as np
import pytest
from oracle_lib import GOLD
from ngspeciesid_amd._capi import ReadSet, cluster_params


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.mark.parametrize("tag", ["sample_h1", "synth200"])
@pytest.mark.parametrize("kw", [(13, 20), (15, 50), (10, 100), (21, 21)])
def test_minimizers_hpc(oracle, tag, kw):
    g = _load("minimizers_%s.npz" % tag)
    k, w = kw
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    moff, codes, pos, hl, he = oracle.hpc_minimizers(rs, k, w)
    assert np.array_equal(hl, np.diff(g["hpc_off"].astype(np.int64)))          # cluster.py:265
    assert np.array_equal(moff, g["moff_%d_%d" % (k, w)])                       # cluster.py:16-39
    assert np.array_equal(codes, g["codes_%d_%d" % (k, w)])
    assert np.array_equal(pos, g["pos_%d_%d" % (k, w)])
    # error rate: reference sums in set (hash) order, this build in ascending character order -> <= 4 ulp
    ref = g["hpc_err"]
    assert np.all(np.abs(he - ref) <= 16 * np.spacing(ref))   # summation order differs (hash order vs ascending), a few ulp


def test_scores(oracle):
    g = _load("scores_sample_h1.npz")
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    score, err, keep = oracle.score_reads(rs, int(g["k"]), 7.0)
    assert np.array_equal(keep, g["keep"])
    m = g["keep"] == 1
    assert np.array_equal(score[m], g["score"][m])                              # sliding product replayed op for op
    assert np.all(np.abs(err[m] - g["err"][m]) <= 16 * np.spacing(g["err"][m]))


def _acc_rank(acc):
    order = np.argsort(np.array(acc, dtype=object), kind="stable")
    rank = np.zeros(len(acc), dtype=np.uint32)
    r = 0
    for j, i in enumerate(order):
        if j and acc[order[j - 1]] != acc[i]:
            r += 1
        rank[i] = r
    return rank


@pytest.mark.parametrize("kw", [(25, 30), (30, 35), (22, 22), (32, 40)])
def test_minimizers_k_above_21(oracle, kw):
    """k > 21 (the reference's table goes to k = 30): positions == cluster.get_kmer_minimizers, codes == the dense rank of the reference's k-mer
    STRINGS among all minimizers of the call (tests/golden/minimizers_widek.npz, oracle/make_golden_widek.py)"""
    g = _load("minimizers_widek.npz")
    k, w = kw
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    moff, codes, pos, hl, he = oracle.hpc_minimizers(rs, k, w)
    assert np.array_equal(moff, g["moff_%d_%d" % (k, w)]) and np.array_equal(pos, g["pos_%d_%d" % (k, w)])
    assert np.array_equal(codes, g["rank_%d_%d" % (k, w)])


@pytest.mark.parametrize("tag", ["sample_h1", "synth2k_d15", "synth600_d10_q14", "synth300_ccs", "synth1200_k25", "synth1200_k30"])
def test_cluster_t1(oracle, tag):
    g = _load("cluster_%s.npz" % tag)
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    prm = cluster_params(k=int(g["k"]), w=int(g["w"]), p_shared=g["p_table"])
    n = rs.n
    bm = np.zeros(n, dtype=np.int32); ns = np.zeros(n, dtype=np.int32); ra = np.zeros(n)
    import ctypes as C
    oracle.lib.ongsid_debug_enable_trace(bm.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p), ra.ctypes.data_as(C.c_void_p), C.c_uint64(n))
    try:
        rep, herr, st, cnt = oracle.cluster_greedy(rs, prm, acc_rank=_acc_rank([str(a) for a in g["acc"]]))
    finally:
        oracle.lib.ongsid_debug_enable_trace(None, None, None, C.c_uint64(0))
    assert np.array_equal(rep, g["t1_rep_of"])                                   # identical cluster membership
    assert [int(c) for c in cnt[:3]] == [int(c) for c in g["t1_counters"]]       # cluster.py:349-351
    # per-read mapping-stage triple (cluster.py:302)
    assert np.array_equal(bm, g["t1_tr_best"])
    assert np.array_equal(ns, g["t1_tr_nshared"])
    assert np.allclose(ra, g["t1_tr_ratio"], rtol=0, atol=0)
    reps = g["t1_rep_err"]
    m = ~np.isnan(reps)
    assert np.all(np.abs(herr[m] - reps[m]) <= 16 * np.spacing(reps[m]))


def test_align_windows(oracle):
    g = _load("align_sample_h1.npz")
    q = ReadSet(g["q"], None, g["q_off"]); t = ReadSet(g["t"], None, g["t_off"])
    n = q.n
    idx = np.arange(n, dtype=np.uint32)
    score, ncols, nmatch, region = oracle.sg_align_batch(q, t, idx, idx, g["open"], 1, 2, -2, 13, g["match_id"])
    assert np.array_equal(score, g["score"])
    assert np.array_equal(ncols, g["n_cols"])
    assert np.array_equal(nmatch, g["n_match"])
    qlen = np.diff(g["q_off"].astype(np.int64)); tlen = np.diff(g["t_off"].astype(np.int64))
    assert np.array_equal(region / qlen.astype(np.float64), g["ratio"])          # cluster.py:167
    assert np.array_equal(region / tlen.astype(np.float64), g["target_ratio"])   # cluster.py:168
