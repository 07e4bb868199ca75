"""CPU: the host-side batch + tree-merge schedule (ngspeciesid_amd.parallelize) against the reference's --t N results.
The clustering backend here is the CPU oracle (as a stand-in for the C-ABI library); on the GPU the same test runs in
tests/test_gpu_parity.py::test_tree_merge."""
import os
import numpy as np
import pytest
from oracle_lib import GOLD
from ngspeciesid_amd import parallelize
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.hostutil import acc_rank, make_cluster_fn


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_batch_list_quirks():
    g = _load("batch_list.npz")
    for name in ("eq8_c4", "eq8_c8", "ragged_c3", "fill_c2", "one_c4", "n13_c8"):
        got = [b - a for a, b in parallelize.batch_list_total_nt(g[name + "_lens"], int(g[name + "_cores"]))]
        assert got == g[name + "_sizes"].tolist(), name                     # parallelize.py:54-67 incl. the trailing empty batch


def run_tree(api, tag, t):
    g = _load("cluster_%s.npz" % tag)
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    prm = cluster_params(k=int(g["k"]), w=int(g["w"]), p_shared=g["p_table"])
    rank = acc_rank([str(a) for a in g["acc"]])
    lens = np.diff(g["off"].astype(np.int64))
    rep_of, herr, joins = parallelize.tree_cluster(make_cluster_fn(api, rs, rank, prm), lens, g["score"], t)
    assert np.array_equal(rep_of, g["t%d_rep_of" % t]), "membership differs from the reference --t %d" % t
    # list order inside every cluster = the reference's clusters dict (consensus.py:257-263 feeds spoa in this order)
    cl = parallelize.cluster_lists_from_joins(rs.n, joins)
    keys, off, order = g["t%d_cl_keys" % t], g["t%d_cl_off" % t], g["t%d_cl_order" % t]
    for i, kk in enumerate(keys):
        assert cl[int(kk)] == order[off[i]:off[i + 1]].tolist()
    return rep_of


@pytest.mark.parametrize("tag,t", [("sample_h1", 1), ("sample_h1", 2), ("sample_h1", 4), ("sample_h1", 8), ("synth2k_d15", 2), ("synth2k_d15", 8),
                                   ("synth600_d10_q14", 4), ("synth300_ccs", 4), ("synth1200_k25", 2), ("synth1200_k30", 2)])
def test_tree_merge_matches_reference(oracle, tag, t):
    run_tree(oracle, tag, t)


def run_round1_then_c_merge(api, tag, t):
    """round 1 per batch (what every GPU does with its shard) + ngsid_merge_representatives (the C schedule, include/ngsid_merge_schedule.h)
    must give the reference's --t N membership, like parallelize.tree_cluster does."""
    from ngspeciesid_amd.hostutil import subset_reads
    g = _load("cluster_%s.npz" % tag)
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    prm = cluster_params(k=int(g["k"]), w=int(g["w"]), p_shared=g["p_table"])
    rank = acc_rank([str(a) for a in g["acc"]])
    lens = np.diff(g["off"].astype(np.int64))
    rep_of = np.arange(rs.n, dtype=np.int64); herr = np.full(rs.n, np.nan); batch = np.zeros(rs.n, dtype=np.int32)
    for b, (a0, a1) in enumerate(parallelize.batch_list_total_nt(lens, t)):
        if a1 <= a0:
            continue
        idx = np.arange(a0, a1)
        r, he, st, _ = api.cluster_greedy(subset_reads(rs, idx), prm, acc_rank=rank[idx])
        rep_of[idx] = idx[r]; herr[idx] = he; batch[idx] = b + 1
    reps = np.nonzero(rep_of == np.arange(rs.n))[0]
    m = api.merge_representatives(subset_reads(rs, reps), prm, g["score"][reps], herr[reps], batch[reps], t, acc_rank=rank[reps])
    final = reps[m][np.searchsorted(reps, rep_of)]
    assert np.array_equal(final, g["t%d_rep_of" % t]), "membership differs from the reference --t %d" % t


@pytest.mark.parametrize("tag,t", [("sample_h1", 2), ("sample_h1", 4), ("sample_h1", 8), ("synth2k_d15", 8), ("synth600_d10_q14", 4), ("synth300_ccs", 4)])
def test_c_merge_schedule_matches_reference(oracle, tag, t):
    run_round1_then_c_merge(oracle, tag, t)
