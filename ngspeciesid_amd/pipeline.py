"""The hot path end to end on one GPU: cluster -> draft consensus (spoa-style) -> rc merge -> polish (racon-style).

Array-level driver used by bench.py and by the reference-shaped host layer (cluster.py / consensus.py of this
package).  Everything heavy happens behind the C-ABI; this module only does the bookkeeping the reference does in
Python dicts, with numpy on index arrays.
"""
from __future__ import annotations
import time
import numpy as np
from ._capi import Api, ReadSet, cluster_params, poa_params, polish_params, POA_LOCAL

# Draft consensus: coverage-trim the ends of every tile consensus (ngsid_poa_params_t.trim).  spoa itself completes the heaviest bundle to
# a sink, so its consensus can end in the unsupported tail of a single read, and racon cannot shorten or extend a backbone end; with the
# trim the drafts of the noisy synthetic sets equal their amplicons before polishing.
# Reads per exact-order POA tile (draft and polishing windows).  With coverage-trimmed tile consensuses the depth does not matter for the accuracy on deep
# clusters (exact from 5.6 % to 14.3 % read error at depths 8 / 6 / 5 / 4: 0 of 250 polished sequences wrong per depth, 40 000 reads per cluster), and on shallow,
# noisy ones the SMALLER tile is the better one (100 reads per cluster at 14.3 % error: 12 of 100 polished sequences wrong at depth 4, 25 at depth 6; 200 reads and
# more: none at either) - profiles/r05_tile_depth_sweep.txt.  Round 5: 4 (was 6 since round 3): the graphs of a tile stay smaller (fewer rows per alignment), k_poa_tile
# 396 -> 363 ms per C3 step, the step 780 -> 748 ms; depth 3 loses the majority inside a tile (2 edits per amplicon on the bench workload) and is slower again.
TILE_DEPTH = 4
# Round 6: a unit (a cluster in the draft, a window in the polisher) with FEWER sequences than this is aligned as ONE graph in file order - spoa's / racon's own order
# (consensus.py:257-266,87) - instead of being tiled (ngsid_poa_params_t.single_below).  Tiling is a throughput device for deep clusters; profiles/r06_tile_depth_sweep.txt.
SINGLE_BELOW = 64
import os as _os
_TOUCH = bool(_os.environ.get("NGSID_TOUCH"))          # dev probe (round 5): one trivial device operation in the middle of the host work between clustering and consensus
DRAFT_TRIM = 1

_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def revcomp_str(s: str) -> str:
    return _COMP[np.frombuffer(s.encode(), dtype=np.uint8)[::-1]].tobytes().decode()


def clusters_from_rep(rep_of: np.ndarray):
    """-> (reps, order, grp_off): clusters keyed by representative read; order lists each cluster's reads with the
    representative first and the members in processing order (cluster.py:338-345: the representative was processed
    before any read that joined it, and reads are processed in index order)."""
    from . import fastio
    return fastio.group_by_rep(rep_of)                               # three counting passes in the host library (NumPy: compare, cumsum, gather, radix argsort, bincount - 7 ms per 10^6 reads)


def select_centers(reps, counts, score, abundance_cutoff):
    """clusters sorted by (size, representative score) descending, size >= cutoff   (consensus.py:254-256)."""
    idx = np.lexsort((-score[reps], -counts))          # primary: size desc, secondary: score desc (stable)
    return [int(i) for i in idx if counts[i] >= abundance_cutoff]


def detect_reverse_complements(api: Api, centers, rc_identity_threshold):
    """consensus.detect_reverse_complements (consensus.py:148-183): centers = [n_reads, c_id, seq, groups(list of cluster ids)].
    Identity = matching columns / alignment columns of the semi-global alignment (open 3, ext 1, +2/-2), max over fw / rc."""
    n = len(centers)
    if n <= 1:
        return [[c[0], c[1], c[2], list(c[3])] for c in centers]
    seqs = [c[2] for c in centers]
    rcs = [revcomp_str(s) for s in seqs]
    q = ReadSet.from_strings(seqs); t = ReadSet.from_strings(seqs + rcs)
    qi, ti = [], []
    for i in range(n):
        for j in range(i + 1, n):
            qi += [i, i]; ti += [j, n + j]
    score, ncols, nmatch, _ = api.sg_align_batch(q, t, qi, ti, 3, 1, 2, -2, 13, None)
    ident = {}
    p = 0
    for i in range(n):
        for j in range(i + 1, n):
            fw = nmatch[p] / float(ncols[p]); rc = nmatch[p + 1] / float(ncols[p + 1]); p += 2
            ident[(i, j)] = max(fw, rc)
    out, removed = [], set()
    for i in range(n):
        nr, cid, seq, groups = centers[i]
        if cid in removed:
            continue
        merged_n, allg = nr, list(groups)
        if i < n - 1:
            for j in range(i + 1, n):
                if ident[(i, j)] >= rc_identity_threshold:          # NB the reference also re-merges already removed centres
                    merged_n += centers[j][0]; removed.add(centers[j][1]); allg += list(centers[j][3])
        out.append([merged_n, cid, seq, allg])
    return out


def pooled_read_lists(merged, group_reads):
    """reads polished against every merged centre: the pooled files of consensus.py:208-215.  The reference re-merges centres that were removed
    already (detect_reverse_complements above), so one cluster can be pooled under two centres; the polisher keeps strand and layers per read, so
    a read stays with the FIRST centre that lists it (a deviation in that rare case, logged)."""
    seen = set(); out = []; dup = 0
    for m in merged:
        parts = []
        for ci in m[3]:
            if ci in seen:
                dup += 1; continue
            seen.add(ci); parts.append(group_reads(ci))
        out.append(np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32))
    if dup:
        import logging
        logging.warning("%d cluster(s) were merged into more than one centre: their reads polish the first of them only", dup)
    return out


def run_hot_path(api: Api, rs: ReadSet, score: np.ndarray, acc_rank=None, k=13, w=20, abundance_ratio=0.1,
                 rc_identity_threshold=0.9, max_seqs_for_consensus=-1, racon_iter=3, tile_depth=None, band=0, node_cap=0,
                 p_shared=None, cluster_kwargs=None, do_consensus=True, do_polish=True, timings=None, polish_trim=2, polish_aln_mode=2, polish_stop_when_stable=True,
                 strand_aware=False, draft_trim=None, single_below=None):
    """Returns dict(rep_of, status, counters, hpc_err, centers=[(n_reads, c_id, draft, polished, groups)]); with strand_aware (extension, off by
    default: strand.py) also flip [n] = reads that were reverse-complemented for the consensus stages, and rep_of is the merged membership."""
    tile_depth = TILE_DEPTH if tile_depth is None else tile_depth
    single_below = SINGLE_BELOW if single_below is None else single_below
    T = timings if timings is not None else {}
    t0 = time.perf_counter()
    prm = cluster_params(k=k, w=w, p_shared=p_shared, **(cluster_kwargs or {}))
    rep_of, herr, status, counters = api.cluster_greedy(rs, prm, acc_rank=acc_rank)
    T["cluster"] = T.get("cluster", 0.0) + time.perf_counter() - t0
    res = dict(rep_of=rep_of, status=status, counters=counters, hpc_err=herr, centers=[])
    n = rs.n
    if strand_aware:
        from . import strand
        t0 = time.perf_counter()
        rep_of, flip, _, sinfo = strand.strand_merge(api, rs, rep_of, score, prm, min_size=max(2, int(abundance_ratio * n) // 2))
        if flip.any():
            rs = strand.orient_reads(rs, flip)                                   # the consensus stages see every cluster in one orientation
        res.update(rep_of=rep_of, flip=flip, strand_info=sinfo)
        T["strand_merge"] = T.get("strand_merge", 0.0) + time.perf_counter() - t0
    if not do_consensus:
        return res
    t0 = time.perf_counter()
    reps, order, grp_off, counts = clusters_from_rep(rep_of)
    if _TOUCH and hasattr(api, "ctx"):
        import ctypes as _C
        api.lib.ngsid_ctx_option(api.ctx, b"touch", _C.c_int64(1))
    cutoff = int(abundance_ratio * n)                                           # NGSpeciesID:65
    sel = select_centers(reps, counts, score, cutoff)
    T["host_group"] = T.get("host_group", 0.0) + time.perf_counter() - t0
    if not sel:
        return res
    t0 = time.perf_counter()
    sub_order, sub_off = [], [0]
    for ci in sel:
        a, b = int(grp_off[ci]), int(grp_off[ci + 1])
        if max_seqs_for_consensus >= 0:
            b = min(b, a + max_seqs_for_consensus)                              # consensus.py:260
        sub_order.append(order[a:b]); sub_off.append(sub_off[-1] + (b - a))
    sub_order = np.concatenate(sub_order) if sub_order else np.zeros(0, np.uint32)
    drafts = api.poa_consensus(rs, sub_off, poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=tile_depth, band=band, node_cap=node_cap, trim=DRAFT_TRIM if draft_trim is None else draft_trim, single_below=single_below),
                               read_order=sub_order)
    T["consensus"] = T.get("consensus", 0.0) + time.perf_counter() - t0
    t0 = time.perf_counter()
    centers = [[int(counts[ci]), int(reps[ci]), drafts[x], [ci]] for x, ci in enumerate(sel)]
    merged = detect_reverse_complements(api, centers, rc_identity_threshold)
    T["rc_merge"] = T.get("rc_merge", 0.0) + time.perf_counter() - t0
    polished = [m[2] for m in merged]
    if do_polish and racon_iter > 0:
        t0 = time.perf_counter()
        def group_reads(ci):                                                    # pooled reads of the merged clusters (consensus.py:208-215)
            a, b = int(grp_off[ci]), int(grp_off[ci + 1])
            if max_seqs_for_consensus >= 0:
                b = min(b, a + max_seqs_for_consensus)                          # the pooled file is built from the truncated reads_c_id files
            return order[a:b]
        lists = pooled_read_lists(merged, group_reads)
        pprm = polish_params(iters=racon_iter, k=k, w=w, tile_depth=tile_depth, band=band, node_cap=node_cap, trim=polish_trim, aln_mode=polish_aln_mode, stop_when_stable=polish_stop_when_stable, single_below=single_below)
        p_off = np.concatenate(([0], np.cumsum([len(x) for x in lists])))
        polished, used = api.polish(ReadSet.from_strings([m[2] for m in merged]), rs, p_off, pprm, read_order=np.concatenate(lists))      # (dealt to two contexts when it pays: _capi.Api lanes)
        T["polish"] = T.get("polish", 0.0) + time.perf_counter() - t0
    res["centers"] = [(m[0], m[1], m[2], polished[i], [int(reps[ci]) for ci in m[3]]) for i, m in enumerate(merged)]
    return res
