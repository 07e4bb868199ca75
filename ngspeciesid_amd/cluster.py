"""Reference-shaped clustering interface (modules/cluster.py) on top of ngsid_cluster_greedy.

Same call signature, argument mutation and return structure as cluster.reads_to_clusters (cluster.py:207-353) so that a
caller of the reference (NGSpeciesID:29, parallelize.py:19,146,156) can switch modules.  The per-read Python loop, the
dict/set minimizer database and parasail are replaced by one C-ABI call; this file only moves lists between dicts.
"""
from __future__ import annotations
import itertools
import logging
import numpy as np
from . import runtime
from ._capi import ReadSet, cluster_params, ST_SHORT, ST_SEEDED
from .hostutil import acc_rank
from .ptable import dict_to_table


def _hpc(seq):
    return "".join(ch for ch, _ in itertools.groupby(seq))


def get_kmer_minimizers(seq, k_size, w_size, api=None):
    """cluster.get_kmer_minimizers(seq,k,w) -> [(kmer, pos)] for an (already homopolymer-compressed) string, computed on the GPU."""
    api = api or runtime.get_api()
    rs = ReadSet.from_strings([seq], ["I" * len(seq)])
    moff, codes, pos, hl, he = api.hpc_minimizers(rs, k_size, w_size)
    dec = {0: "", 1: "A", 2: "C", 3: "G", 4: "N", 5: "T"}
    out = []
    for c, p in zip(codes.tolist(), pos.tolist()):
        out.append(("".join(dec[(c >> (3 * (k_size - 1 - i))) & 7] for i in range(k_size)), p))
    return out


def reads_to_clusters(clusters, representatives, sorted_reads, p_emp_probs, minimizer_database, new_batch_index, args, api=None):
    api = api or runtime.get_api()
    n = len(sorted_reads)
    if n == 0:
        return {new_batch_index: (clusters, representatives, minimizer_database, new_batch_index)}
    ids = [r[0] for r in sorted_reads]
    rs = ReadSet.from_strings([r[3] for r in sorted_reads], [r[4] for r in sorted_reads])
    prev = np.array([r[1] for r in sorted_reads], dtype=np.int32)
    known = np.array([representatives[i][6] if len(representatives[i]) == 8 else np.nan for i in ids], dtype=np.float64)
    prm = cluster_params(k=args.k, w=args.w, min_shared=args.min_shared, min_fraction=args.min_fraction, mapped_threshold=args.mapped_threshold,
                         aligned_threshold=args.aligned_threshold, min_prob_no_hits=args.min_prob_no_hits,
                         symmetric=bool(getattr(args, "symmetric_map_align_thresholds", False)), p_shared=dict_to_table(p_emp_probs))
    rep, herr, st, cnt = api.cluster_greedy(rs, prm, acc_rank=acc_rank([r[2] for r in sorted_reads]),
                                            prev_batch=prev if prev.any() else None, known_err=known if prev.any() else None)
    cluster_to_new = {}
    for j, (read_cl_id, b_i, acc, seq, qual, score) in enumerate(sorted_reads):
        if st[j] == ST_SHORT:
            continue                                                           # cluster.py:266-268
        if st[j] == ST_SEEDED or len(representatives[read_cl_id]) == 8:
            t = list(representatives[read_cl_id]); t[1] = new_batch_index; representatives[read_cl_id] = tuple(t)     # :244-247,274-277
        else:
            representatives[read_cl_id] = (read_cl_id, new_batch_index, acc, seq, qual, score, float(herr[j]), _hpc(seq))   # :292
        if rep[j] != j:
            cluster_to_new[read_cl_id] = ids[rep[j]]                           # :324
    for read_cl_id, new_cl_id in cluster_to_new.items():                       # :338-345
        clusters[new_cl_id].extend(clusters[read_cl_id])
        del clusters[read_cl_id]
        del representatives[read_cl_id]
    logging.debug("Total number of reads iterated through:{0}".format(n))
    logging.debug("Passed mapping criteria:{0}".format(int(cnt[0])))
    logging.debug("Passed alignment criteria in this process:{0}".format(int(cnt[1])))
    logging.debug("Total calls to alignment module in this process:{0}".format(int(cnt[2])))
    return {new_batch_index: (clusters, representatives, minimizer_database, new_batch_index)}


def p_shared_minimizer_empirical(error_rate_read, error_rate_center, p_emp_probs):
    e1 = min(max(round(error_rate_read, 2), 0.01), 0.15)
    e2 = min(max(round(error_rate_center, 2), 0.01), 0.15)
    return p_emp_probs[(e1, e2)]
