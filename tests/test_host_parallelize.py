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


def test_round_dumps_of_parallel_clustering_equal_the_reference(oracle, tmp_path, monkeypatch):
    """parallelize.py:85-104,193: after every round but the last, parallel_clustering leaves <outfolder>/<it>/pre_clusters.csv and cluster_origins.csv.  The reference's own files
    (tests/golden/sample_h1_t4_round_dumps.json, written by oracle/make_golden_rounds.py: the reference imported and run on its test/sample_h1.fastq with --t 4) against the
    reference-shaped function of this package on the oracle backend: pre_clusters.csv byte for byte, cluster_origins.csv field for field (the error-rate column to a few ulp:
    the reference sums it in hash order)."""
    import argparse, json, os
    from oracle_lib import GOLD
    from ngspeciesid_amd import runtime, parallelize, get_sorted_fastq_for_cluster, help_functions
    from ngspeciesid_amd.ptable import p_emp_probs_dict
    monkeypatch.setattr(runtime, "get_api", lambda device=None: oracle)
    gold = json.load(open(os.path.join(GOLD, "sample_h1_t4_round_dumps.json")))
    out = str(tmp_path / "o"); os.makedirs(out)
    args = argparse.Namespace(k=13, w=20, min_shared=5, mapped_threshold=0.7, aligned_threshold=0.4, symmetric_map_align_thresholds=False, min_fraction=0.8, min_prob_no_hits=0.1,
                              print_output=10 ** 9, nr_cores=4, batch_type="total_nt", quality_threshold=7.0, outfolder=out, fastq=os.path.join(GOLD, "sample_h1.fastq"), use_old_sorted_file=False)
    args.outfile = os.path.join(out, "sorted.fastq")
    path = get_sorted_fastq_for_cluster.main(args)
    with open(path) as fh:
        reads = [(i, 0, acc, s, q, float(acc.rsplit("_", 1)[1])) for i, (acc, (s, q)) in enumerate(help_functions.readfq(fh))]
    parallelize.parallel_clustering(reads, p_emp_probs_dict(13, 20), args)
    got = {}
    for it in os.listdir(out):
        if it.isdigit():
            for f in os.listdir(os.path.join(out, it)):
                got["%s/%s" % (it, f)] = open(os.path.join(out, it, f)).read()
    assert sorted(got) == sorted(gold)
    for k in gold:
        if k.endswith("pre_clusters.csv"):
            assert got[k] == gold[k], k
        else:
            a = [l.split("\t") for l in got[k].splitlines()]; b = [l.split("\t") for l in gold[k].splitlines()]
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert x[:5] == y[:5]
                assert abs(float(x[5]) - float(y[5])) <= 16 * np.spacing(float(y[5]))


def test_list_positions_kept_up_to_date_round_by_round():
    """parallelize.ListPositions (one pass over the reads per ROUND) == parallelize.list_positions (replay of all joins) == the plain replay of the reference's list moves, on
    random merge trees: rounds of disjoint calls in which surviving representatives join earlier ones"""
    rng = np.random.default_rng(12)
    for N, rounds, calls in ((40, 3, 2), (500, 4, 4), (3000, 5, 8)):
        alive = np.arange(N); joins = []; lp = parallelize.ListPositions(N)
        for _ in range(rounds):
            parts = np.array_split(rng.permutation(alive), calls); n0 = len(joins); gone = []
            for p in parts:
                p = np.sort(p)
                if len(p) < 2: continue
                k = max(1, len(p) // 3); keep = p[:k]; mov = p[k:][rng.random(len(p) - k) < 0.6]
                if len(mov) == 0: continue
                joins.append((mov, keep[rng.integers(0, k, len(mov))])); gone.append(mov)
            pos = lp.apply(joins[n0:]).copy()
            if gone: alive = np.setdiff1d(alive, np.concatenate(gone))
            assert np.array_equal(pos, parallelize.list_positions(N, joins))
            lists = parallelize.cluster_lists_from_joins(N, joins)
            for rep, members in lists.items():
                assert [int(pos[m]) for m in members] == list(range(len(members))) and all(int(lp.root[m]) == rep for m in members)
