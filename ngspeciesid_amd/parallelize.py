"""Host parallelism of the clustering stage: contiguous batches + pairwise tree merge (reference modules/parallelize.py).

`--t N` of the reference = N score-ordered batches clustered independently, then merged pairwise in a binary tree where
the lower batch's representatives keep their database and the higher batch's representatives are re-clustered against
it (parallelize.py:107-217, cluster.py:221-223,243-248).  The same schedule is what shards the path over N GPUs: batch g
lives on GPU g, the only exchange is the all-gather of the surviving representatives before the merge rounds
(`distributed.py`).  Everything here is index-array bookkeeping; the clustering itself is ngsid_cluster_greedy.
"""
from __future__ import annotations
import logging
import numpy as np


def batch_list_total_nt(lens, nr_cores):
    """batch_list(lst, nr_cores, 'total_nt') (parallelize.py:54-67) on read lengths -> list of (start, end) slices.
    Quirks kept: can yield fewer batches than cores, and a trailing EMPTY batch when the last read fills a chunk."""
    lens = np.asarray(lens, dtype=np.int64)
    tot = int(lens.sum())
    chunk = int(tot / nr_cores) + 1
    cs = np.cumsum(lens)
    out, start, base = [], 0, 0
    while start < len(lens):                                   # one searchsorted per batch instead of a Python loop over the reads
        i = int(np.searchsorted(cs, base + chunk, side="left"))    # first read at which the running sum since `start` reaches the chunk
        if i >= len(lens):
            break
        out.append((start, i + 1)); start = i + 1; base = int(cs[i])
    out.append((start, len(lens)))
    return out


def batch_list_nr_reads(n, nr_cores):
    """batch_list(..., 'nr_reads') (parallelize.py:47-52)."""
    chunk = int(n / nr_cores) + 1
    return [(a, min(a + chunk, n)) for a in range(0, n, chunk)]


def batch_merge_consecutive(prev_idx):
    """batch_list(..., merge_consecutive=True) (parallelize.py:34-45) on the previous batch indices of the score-ordered
    representatives -> list of index lists."""
    batch_id, batch, out = 2, [], []
    for i, b in enumerate(prev_idx):
        if b <= batch_id:
            batch.append(i)
        else:
            out.append(batch); batch_id += 2; batch = [i]
    out.append(batch)
    return out


def tree_cluster(cluster_fn, lens, score, nr_cores, batch_type="total_nt", state=None, on_round=None, track=None, cluster_fn_many=None):
    """The whole parallel_clustering schedule on index arrays.

    cluster_fn(read_idx, prev_batch, known_err) -> (rep_local, herr, status, counters): clusters the reads `read_idx`
      (global indices, processing order) and returns for each the LOCAL index of its representative.
    state: optional (bidx, herr) = the situation AFTER round 1 (every index is a surviving representative of batch
      bidx[i] >= 1 with HPC error rate herr[i]); the multi-GPU path runs round 1 on the ranks, all-gathers the
      representatives and enters here for the merge rounds only.
    Returns (rep_of [N] global representative per read, herr [N] (NaN where unknown), joins)
    cluster_fn_many([(read_idx, prev_batch, known_err), ...]) -> [results]: optional; the calls of ONE round (disjoint batches: none reads what another writes) handed over together,
      so that the caller may run them side by side (fastpath: two contexts of the device); results in the order of the calls.
    track: optional dict; on return track["pos"] = list_positions(N, joins), kept up to date round by round (ListPositions) instead of replayed from the joins afterwards.
    on_round(it, reps, root, herr, joins, pos): called after every round but the last (where the reference writes its per-round dumps, parallelize.py:193) with the surviving
      representatives in the order of the merged cluster dictionaries (batch by batch, input order within a batch), every read's representative and list position as they stand
      (ListPositions) and the joins so far.
    where joins lists, per cluster_fn call, the arrays (joining_reps, new_reps) in the order the reference moves read lists
    (cluster.py:338-345); within one call nothing joins a read that itself joined (a joined read is no representative any more).
    """
    N = len(lens)
    rep_of = np.arange(N, dtype=np.int64)
    joins = []
    if state is None:
        herr = np.full(N, np.nan)
        bidx = np.zeros(N, dtype=np.int64)
        if batch_type == "nr_reads":
            batches = batch_list_nr_reads(N, nr_cores)
        else:
            batches = batch_list_total_nt(lens, nr_cores)
        cur_batches = [np.arange(a, b, dtype=np.int64) for a, b in batches]
        prev = None                                # previous batch index per read in cur_batches (None = all 0)
        num_batches = nr_cores
    else:
        bidx = np.asarray(state[0], dtype=np.int64).copy(); herr = np.asarray(state[1], dtype=np.float64).copy()
        # parallelize.py:184 sorts by score; batches hold contiguous score ranges there, so (batch, score) is the same order
        # and stays well defined when the shards of a multi-GPU run were scored independently
        reps = np.lexsort((-np.asarray(score), bidx))
        groups = batch_merge_consecutive(bidx[reps])
        cur_batches = [reps[np.asarray(g, dtype=np.int64)] if len(g) else np.zeros(0, dtype=np.int64) for g in groups]
        prev = True
        num_batches = len(cur_batches)
        if nr_cores == 1:
            return rep_of, herr, joins
    # ---- round structure of parallelize.py:136-217
    it = 1
    lp = ListPositions(N) if (track is not None or on_round is not None) else None
    while True:
        single = len(cur_batches) == 1
        alive_next = []; n_before = len(joins)
        todo = [(bi, idx, None if prev is None else bidx[idx].astype(np.int32), None if prev is None else herr[idx]) for bi, idx in enumerate(cur_batches) if len(idx)]
        done = cluster_fn_many([c[1:] for c in todo]) if (cluster_fn_many is not None and len(todo) > 1) else None
        for x, (bi, idx, pb, ke) in enumerate(todo):
            new_index = 1 if single else bi + 1
            rep_local, he, st, _ = done[x] if done is not None else cluster_fn(idx, pb, ke)
            rep_g = idx[np.asarray(rep_local, dtype=np.int64)]
            moved = rep_g != idx
            if moved.any():
                joins.append((idx[moved].copy(), rep_g[moved].copy()))      # one entry per call, in processing order = cluster_to_new_cluster_id order
            rep_of[idx[moved]] = rep_g[moved]
            surv = idx[~moved]
            known = ~np.isnan(np.asarray(he)[~moved])
            herr[surv[known]] = np.asarray(he)[~moved][known]
            # batch index update (cluster.py:245,275,292): every surviving processed / seeded read gets new_index,
            # except reads skipped for HPC length < k (status 3) which keep their old index
            keep_old = np.asarray(st)[~moved] == 3
            bidx[surv[~keep_old]] = new_index
            alive_next.append(surv)
        if lp is not None: lp.apply(joins[n_before:])
        if single or num_batches == 1:
            break
        reps = np.concatenate(alive_next) if alive_next else np.zeros(0, dtype=np.int64)
        if on_round is not None:
            on_round(it, reps, lp.root, herr, joins, lp.pos)
        # sorted(all_representatives, key=score, reverse=True): stable; dict order = batch order then read order
        order = np.lexsort((-score[reps], bidx[reps]))                 # = sort by score on contiguous batches (see above)
        reps = reps[order]
        it += 1
        groups = batch_merge_consecutive(bidx[reps])
        cur_batches = [reps[np.asarray(g, dtype=np.int64)] if len(g) else np.zeros(0, dtype=np.int64) for g in groups]
        num_batches = len(cur_batches)
        prev = True
        logging.debug("Batches after pairwise consecutive merge: %d", num_batches)
    # path compression: a representative that joined later drags its reads (cluster.py:338-345)
    for _ in range(64):
        nxt = rep_of[rep_of]
        if np.array_equal(nxt, rep_of):
            break
        rep_of = nxt
    if track is not None: track["pos"] = lp.pos
    return rep_of, herr, joins


def cluster_lists_from_joins(N, joins):
    """Replay of clusters[new].append(...) / del clusters[old] (cluster.py:338-345) -> {rep: [member indices in the reference's list order]}.
    Plain dict / list replay (small inputs, tests); list_positions() is the array form of the same order."""
    clusters = {i: [i] for i in range(N)}
    for aa, bb in joins:
        for a, b in zip(np.asarray(aa).tolist(), np.asarray(bb).tolist()):
            clusters[b].extend(clusters[a]); del clusters[a]
    return clusters


class ListPositions:
    """list_positions() kept up to date round by round: root[x] = the representative whose list read x is in, pos[x] = its place there.  A round's joins (one entry per
    cluster_fn call; the calls of a round work on disjoint representatives, and within a call nothing joins a read that itself joined) move every list behind the list it joins:
    all reads of a moved list shift by the same offset, so a round costs a few passes over the reads instead of a replay of all rounds."""
    def __init__(self, N):
        self.N = N; self.root = np.arange(N, dtype=np.int64); self.pos = np.zeros(N, dtype=np.int64); self.size = np.ones(N, dtype=np.int64)

    def apply(self, round_joins):
        N = self.N; size = self.size
        off = np.zeros(N, dtype=np.int64); new = np.arange(N, dtype=np.int64); any_ = False
        for aa, bb in round_joins:
            a = np.asarray(aa, dtype=np.int64); b = np.asarray(bb, dtype=np.int64)
            if len(a) == 0: continue
            any_ = True
            sa = size[a]
            o = np.argsort(b, kind="stable"); bs = b[o]; sas = sa[o]
            csum = np.cumsum(sas) - sas
            first = np.ones(len(bs), dtype=bool); first[1:] = bs[1:] != bs[:-1]
            gstart = csum[first][np.cumsum(first) - 1]
            off[a[o]] = size[bs] + (csum - gstart)
            new[a] = b
            np.add.at(size, b, sa)
        if any_:
            self.pos += off[self.root]; self.root = new[self.root]
        return self.pos


def list_positions(N, joins):
    """Position of every read in the read list of its final cluster (the order cluster.py:338-345 builds: a joining representative's whole list
    is appended to the list of the representative it joins), without replaying lists: the joins form a forest, the final list is its
    pre-order with children in join order, so  pos(x) = pos(parent) + [size of the parent's list before x's call] + [sizes of the children that
    joined the same parent earlier in that call]; the sums along the ancestor chains are taken by pointer jumping."""
    size = np.ones(N, dtype=np.int64); rel = np.zeros(N, dtype=np.int64); parent = np.arange(N, dtype=np.int64)
    for aa, bb in joins:
        a = np.asarray(aa, dtype=np.int64); b = np.asarray(bb, dtype=np.int64)
        if len(a) == 0:
            continue
        sa = size[a]
        o = np.argsort(b, kind="stable"); bs = b[o]; sas = sa[o]
        csum = np.cumsum(sas) - sas
        first = np.ones(len(bs), dtype=bool); first[1:] = bs[1:] != bs[:-1]
        gstart = csum[first][np.cumsum(first) - 1]
        rel[a[o]] = size[bs] + (csum - gstart)
        parent[a] = b
        np.add.at(size, b, sa)
    pos = rel.copy(); p = parent.copy()
    while True:
        gp = p[p]
        pos = pos + np.where(p != np.arange(N), pos[p], 0)
        if np.array_equal(gp, p):
            break
        p = gp
    return pos


def write_round_dumps(clusters, representatives, args, it):
    """<outfolder>/<it>/pre_clusters.csv and cluster_origins.csv after a round of parallel_clustering (parallelize.py:85-104,193): the clusters by size, largest first (ties in
    dict order), members in list order with the score suffix cut off; one origin line per cluster.  Write-only in the reference as well."""
    import os
    from .help_functions import mkdir_p
    folder = os.path.join(args.outfolder, str(it))
    mkdir_p(folder)
    by_size = sorted(clusters, key=lambda c: len(clusters[c]), reverse=True)
    with open(os.path.join(folder, "pre_clusters.csv"), "w") as out:
        for c in by_size:
            out.writelines("{0}\t{1}\n".format(c, acc.rsplit("_", 1)[0]) for acc in clusters[c])
    logging.debug("Nr clusters larger than 1: %d", sum(1 for c in by_size if len(clusters[c]) > 1))
    logging.debug("Nr clusters (all):  %d", len(clusters))
    with open(os.path.join(folder, "cluster_origins.csv"), "w") as out:
        for c in by_size:
            t = representatives[c]
            out.write("{0}\t{1}\t{2}\t{3}\t{4}\t{5}\n".format(t[0], t[2], t[3], t[4], t[5], t[6] if len(t) > 6 else ""))     # (a read whose compressed form is shorter than k has none)


def parallel_clustering(read_array, p_emp_probs, args, api=None):
    """parallelize.parallel_clustering(read_array, p_emp_probs, args) -> (clusters, representatives)  (parallelize.py:107-217).
    Same batches, same merge rounds; batches of a round run one after the other on this process's GPU (or are spread over
    the ranks of a torch.distributed job by distributed.py)."""
    from . import cluster as _cluster
    import math
    nr = args.nr_cores
    lens = [len(r[3]) for r in read_array]
    if getattr(args, "batch_type", "total_nt") == "nr_reads":
        sl = batch_list_nr_reads(len(read_array), nr)
    else:
        sl = batch_list_total_nt(lens, nr)
    read_batches = [read_array[a:b] for a, b in sl]
    cluster_batches, origin_batches, dbs = [], [], []
    for batch in read_batches:
        cluster_batches.append({i: [acc] for i, b_i, acc, seq, qual, score in batch})
        origin_batches.append({i: (i, b_i, acc, seq, qual, score) for i, b_i, acc, seq, qual, score in batch})
        dbs.append({})
    num_batches = nr
    it = 1
    while True:
        if len(read_batches) == 1:
            res = _cluster.reads_to_clusters(cluster_batches[0], origin_batches[0], read_batches[0], p_emp_probs, dbs[0], 1, args, api=api)
            c, o, _, _ = res[1]
            return c, o
        all_cl, all_repr, all_db = {}, {}, {}
        for i in range(len(read_batches)):
            res = _cluster.reads_to_clusters(cluster_batches[i], origin_batches[i], read_batches[i], p_emp_probs, dbs[i], i + 1, args, api=api)
            c, o, db, bi = res[i + 1]
            all_cl.update(c); all_repr.update(o); all_db[bi] = db
        read_array = [(i, b_index, acc, seq, qual, score) for i, (i, b_index, acc, seq, qual, score, *_rest) in
                      sorted(all_repr.items(), key=lambda x: x[1][5], reverse=True)]
        if num_batches == 1:
            return all_cl, all_repr
        if getattr(args, "outfolder", None):
            write_round_dumps(all_cl, all_repr, args, it)                     # parallelize.py:193
        it += 1
        groups = batch_merge_consecutive([r[1] for r in read_array])
        read_batches = [[read_array[j] for j in g] for g in groups]
        num_batches = len(read_batches)
        cluster_batches, origin_batches, dbs = [], [], []
        for batch in read_batches:
            cluster_batches.append({i: all_cl[i] for i, *_ in batch})
            origin_batches.append({i: all_repr[i] for i, *_ in batch})
            dbs.append({})
