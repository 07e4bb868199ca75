"""The hot path sharded over N GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Shard g (rank g) = batch g+1 of the reference's `--t N` schedule.  Per rank, no collective: round-1 clustering of the
shard, tile consensuses and window consensuses of the shard's own reads.  Collectives (all small, latency bound):
  1. all-gather of the surviving REPRESENTATIVES (sequence, qualities, score, HPC error rate) after round 1 - the
     reference's only exchange (parallelize.py:169-184) - then every rank replays the pairwise tree merge on them;
  2. all-reduce of cluster sizes (which clusters reach the abundance cutoff);
  3. all-gather of the per-shard partial consensuses (one string per selected cluster and rank), merged by a POA whose
     per-base weights are proportional to the reads each partial stands for - once for the draft and once after the
     polishing iterations (which every shard runs locally against the common merged draft).
Reads never move between GPUs.
"""
from __future__ import annotations
import numpy as np
import torch
import torch.distributed as dist
from . import parallelize, pipeline
from ._capi import ReadSet, cluster_params, poa_params, polish_params, POA_LOCAL, ST_SHORT
from .hostutil import subset_reads


def _pack(obj: dict) -> np.ndarray:
    """dict of numpy arrays / scalars / lists of strings -> one uint8 buffer: a small JSON header (names, dtypes, shapes) + the raw array bytes"""
    import json
    head, chunks = [], []
    for k, v in obj.items():
        if isinstance(v, (list, tuple)) and (len(v) == 0 or isinstance(v[0], str)):
            b = [x.encode() for x in v]; lens = np.array([len(x) for x in b], dtype=np.int64)
            head.append((k, "strlist", [len(b)])); chunks += [lens.tobytes(), b"".join(b)]
        elif v is None:
            head.append((k, "none", []))
        else:
            a = np.asarray(v)
            if a.ndim: a = np.ascontiguousarray(a)           # (ascontiguousarray would turn a scalar into a 1-element vector)
            head.append((k, a.dtype.str, list(a.shape))); chunks.append(a.tobytes())
    h = json.dumps(head).encode()
    return np.frombuffer(np.array([len(h)], dtype=np.int64).tobytes() + h + b"".join(chunks), dtype=np.uint8)


def _unpack(buf: np.ndarray) -> dict:
    import json
    hl = int(buf[:8].view(np.int64)[0]); head = json.loads(buf[8:8 + hl].tobytes().decode()); o = 8 + hl; out = {}
    for k, dt, shape in head:
        if dt == "none":
            out[k] = None
        elif dt == "strlist":
            n = shape[0]; lens = buf[o:o + 8 * n].view(np.int64); o += 8 * n
            strs = []
            for l in lens.tolist():
                strs.append(buf[o:o + l].tobytes().decode()); o += l
            out[k] = strs
        else:
            cnt = int(np.prod(shape)) if shape else 1; nb = cnt * np.dtype(dt).itemsize
            a = buf[o:o + nb].view(np.dtype(dt)).reshape(shape).copy(); o += nb
            out[k] = a if shape else a[()]
    return out


def all_gather_obj(obj: dict, device):
    """variable-length all-gather of a dict of arrays: the sizes first, then ONE padded uint8 all_gather into a single [world, max] tensor
    (RCCL on the GPU, gloo on CPU) and one copy of that tensor to the host - the payloads (representatives, partial consensuses) are consumed
    by host-side schedule code.  No pickling: arrays travel as raw bytes behind a JSON header."""
    world = dist.get_world_size()
    blob = _pack(obj)
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, n)
    sizes = sizes.cpu().numpy()
    mx = max(int(sizes.max()), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    buf[:len(blob)] = torch.from_numpy(blob.copy()).to(device)
    out = torch.empty(world * mx, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, buf)
    host = out.cpu().numpy().reshape(world, mx)
    return [_unpack(host[r, :int(sizes[r])]) for r in range(world)]


class TorchComm:
    """the collectives of the sharded path over torch.distributed (backend nccl = RCCL over xGMI on the GPUs, gloo on CPU)"""

    def __init__(self, device=None):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))

    def all_gather_obj(self, obj):
        return all_gather_obj(obj, self.device)

    def all_reduce_sum(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        dist.all_reduce(t)
        return t.cpu().numpy()


class LocalComm:
    """N VIRTUAL ranks = N threads of one process (each with its own ngsid context on the same GPU): the composed N-shard path on a one-GPU box.
    Same payloads as TorchComm (they go through _pack / _unpack); the exchange is a slot list behind a threading.Barrier.  Eight real processes
    sharing one MI355X stall in the runtime (seen with torch's own generator kernels, before any library call), so the one-GPU emulation of an
    8-GPU run uses threads; torch.distributed itself is covered by the gloo tests on CPU and the 2- / 4-process runs on one GPU."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world = world; self.barrier = threading.Barrier(world); self.slots = [None] * world

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, item):
        sh = self.shared
        sh.slots[self.rank] = item
        sh.barrier.wait()
        out = list(sh.slots)
        sh.barrier.wait()                      # nobody overwrites a slot before everyone has read
        return out

    def all_gather_obj(self, obj):
        return [_unpack(b) for b in self._exchange(_pack(obj))]

    def all_reduce_sum(self, a):
        parts = self._exchange(np.asarray(a).copy())
        return np.sum(parts, axis=0)


def run_virtual_ranks(world, fn):
    """fn(comm) on one thread per virtual rank, comm = LocalComm of that rank (ctypes releases the GIL during the library calls, so the shards'
    GPU work overlaps); returns the results in rank order and re-raises the first real exception (a failed rank aborts the barrier so that
    the others do not wait for ever)"""
    import threading
    shared = LocalComm.Shared(world)
    res = [None] * world; err = [None] * world
    def work(r):
        try:
            res[r] = fn(LocalComm(shared, r))
        except BaseException as e:                  # noqa
            err[r] = e; shared.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real: raise real[0]
    if any(e is not None for e in err): raise [e for e in err if e is not None][0]
    return res


def _weighted_merge_all(api, partials, counts, band):
    """For every selected cluster: POA of its per-shard partial consensuses, each weighted by the NUMBER OF READS it stands for (ngsid_poa_consensus_weighted;
    until round 3 the count was squeezed into a quality character, 1 .. 93: a shard holding under 1 % of a cluster's reads - the common case for the rare
    species of a skewed sample - then weighed 1/93 instead of its share).  partials[c] / counts[c] = one entry per shard.  All clusters go through ONE
    library call (one group each)."""
    out = [""] * len(partials)
    seqs, weights, grp, which = [], [], [0], []
    for c, (ps, cs) in enumerate(zip(partials, counts)):
        items = [(s, int(n)) for s, n in zip(ps, cs) if s and n > 0]
        if not items:
            continue
        if len(items) == 1:
            out[c] = items[0][0]; continue
        for s, n in items:
            seqs.append(s); weights.append(n)
        grp.append(len(seqs)); which.append(c)
    if which:
        res = api.poa_consensus_weighted(ReadSet.from_strings(seqs), grp, poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=0, band=band), np.asarray(weights, dtype=np.uint32))
        for c, r in zip(which, res):
            out[c] = r
    return out


def representative_payload(rs_local, rep_local, herr, score_local, acc_rank_local=None):
    """What a shard contributes to the all-gather: its surviving representatives (a few KB each).  -> (mine, payload)"""
    n_local = rs_local.n
    mine = np.nonzero(rep_local == np.arange(n_local))[0]
    if rs_local.mem == 0:
        sub = subset_reads(rs_local, mine)
        payload = dict(idx=mine.astype(np.int64), seq=sub.seq, qual=sub.qual, off=sub.off, score=score_local[mine], herr=herr[mine],
                       rank_key=None if acc_rank_local is None else np.asarray(acc_rank_local)[mine], n=n_local)
    else:   # device resident shard: pull only the representatives to the host
        seq_t, qual_t, off_t = rs_local.keep["seq"], rs_local.keep["qual"], rs_local.keep["off"]
        offs = off_t.cpu().numpy()
        lens = offs[mine + 1] - offs[mine]
        noff = np.zeros(len(mine) + 1, dtype=np.uint64); noff[1:] = np.cumsum(lens)
        gi = torch.from_numpy(np.repeat(offs[mine] - noff[:-1].astype(np.int64), lens) + np.arange(int(noff[-1]), dtype=np.int64)).to(seq_t.device)
        payload = dict(idx=mine.astype(np.int64), seq=seq_t[gi].cpu().numpy(), qual=qual_t[gi].cpu().numpy(), off=noff, score=score_local[mine], herr=herr[mine],
                       rank_key=None if acc_rank_local is None else np.asarray(acc_rank_local)[mine], n=n_local)
    return mine, payload


def merge_representatives(api, gathered, prm, world):
    """Replay of the reference's pairwise tree merge (parallelize.py:169-215) on the all-gathered representatives of `world` shards;
    identical on every rank.  gathered[b] = dict(idx, seq, qual, off, score, herr, rank_key, n) of shard b.
    -> (rep_of_rep, r_owner, r_lidx, r_score, n_total): for every gathered representative the global id of its final representative."""
    n_total = sum(g["n"] for g in gathered)
    seqs = np.concatenate([g["seq"] for g in gathered]); quals = np.concatenate([g["qual"] for g in gathered])
    lens_all = np.concatenate([np.diff(g["off"].astype(np.int64)) for g in gathered])
    off_all = np.zeros(len(lens_all) + 1, dtype=np.uint64); off_all[1:] = np.cumsum(lens_all)
    reps_rs = ReadSet(seqs, quals, off_all)
    r_score = np.concatenate([g["score"] for g in gathered]); r_herr = np.concatenate([g["herr"] for g in gathered])
    r_batch = np.concatenate([np.full(len(g["idx"]), b + 1, dtype=np.int64) for b, g in enumerate(gathered)])
    r_owner = np.concatenate([np.full(len(g["idx"]), b, dtype=np.int64) for b, g in enumerate(gathered)])
    r_lidx = np.concatenate([g["idx"] for g in gathered])
    if gathered[0]["rank_key"] is not None:
        # accession order across shards: (rank key within the shard, shard) - a total order consistent on every rank
        key = np.concatenate([g["rank_key"].astype(np.int64) * world + b for b, g in enumerate(gathered)])
        r_accrank = np.argsort(np.argsort(key, kind="stable"), kind="stable").astype(np.uint32)
    else:
        r_accrank = np.arange(len(r_lidx), dtype=np.uint32)

    rep_of_rep = api.merge_representatives(reps_rs, prm, r_score, r_herr, r_batch, world, acc_rank=r_accrank).astype(np.int64)      # ngsid_merge_representatives
    return rep_of_rep, r_owner, r_lidx, r_score, n_total


def sharded_hot_path(api, rs_local: ReadSet, score_local, acc_rank_local=None, k=13, w=20, abundance_ratio=0.1, rc_identity_threshold=0.9,
                     racon_iter=3, tile_depth=pipeline.TILE_DEPTH, band=0, p_shared=None, cluster_kwargs=None, do_consensus=True, polish_trim=2, device=None, timings=None, polish_stop_when_stable=True, comm=None):
    """Runs on every rank; returns dict(final_rep=(rank, local idx) per local read as two arrays, centers=[(n, key, draft, polished)])."""
    import time
    T = timings if timings is not None else {}
    comm = comm or TorchComm(device)
    world, rank = comm.world, comm.rank
    prm = cluster_params(k=k, w=w, p_shared=p_shared, **(cluster_kwargs or {}))
    n_local = rs_local.n
    score_local = np.asarray(score_local, dtype=np.float64)
    # ---- 1. round 1 on the shard (no communication)
    t0 = time.perf_counter()
    rep_local, herr, st, cnt = api.cluster_greedy(rs_local, prm, acc_rank=acc_rank_local)
    T["cluster_local"] = T.get("cluster_local", 0.0) + time.perf_counter() - t0
    # ---- 2. all-gather the representatives, replay the tree merge everywhere
    t0 = time.perf_counter()
    mine, payload = representative_payload(rs_local, rep_local, herr, score_local, acc_rank_local)
    gathered = comm.all_gather_obj(payload)
    rep_of_rep, r_owner, r_lidx, r_score, n_total = merge_representatives(api, gathered, prm, world)
    base = np.concatenate(([0], np.cumsum([len(g["idx"]) for g in gathered])))
    my_gid = np.full(n_local, -1, dtype=np.int64); my_gid[mine] = base[rank] + np.arange(len(mine))
    final_gid = rep_of_rep[my_gid[rep_local]]                                 # global representative id of every local read
    T["merge"] = T.get("merge", 0.0) + time.perf_counter() - t0
    res = dict(final_owner=r_owner[final_gid], final_lidx=r_lidx[final_gid], final_gid=final_gid, counters=cnt, n_total=n_total, centers=[])
    if not do_consensus:
        return res
    # ---- 3. cluster sizes over all shards
    t0 = time.perf_counter()
    sizes = comm.all_reduce_sum(np.bincount(final_gid, minlength=len(r_lidx)).astype(np.int64))
    cutoff = int(abundance_ratio * n_total)
    cand = np.nonzero((sizes >= cutoff) & (sizes > 0))[0]
    cand = cand[np.lexsort((-r_score[cand], -sizes[cand]))]
    # ---- 4. per-shard partial consensus of every selected cluster, all-gather, weighted merge
    # (few distinct keys: a 16-bit key makes numpy's stable sort a radix sort)
    order = np.argsort(final_gid.astype(np.uint16 if len(r_lidx) < 65536 else np.int64), kind="stable").astype(np.uint32)
    sorted_gid = final_gid[order]
    lo = np.searchsorted(sorted_gid, cand, side="left"); hi = np.searchsorted(sorted_gid, cand, side="right")
    sub_order = np.concatenate([order[a:b] for a, b in zip(lo, hi)]) if len(cand) else np.zeros(0, np.uint32)
    sub_off = np.concatenate(([0], np.cumsum(hi - lo))).astype(np.uint64)
    partial = api.poa_consensus(rs_local, sub_off, poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=tile_depth, band=band, trim=pipeline.DRAFT_TRIM), read_order=sub_order) if len(cand) else []
    allp = comm.all_gather_obj(dict(cons=list(partial), cnt=np.asarray(hi - lo, dtype=np.int64)))
    drafts = _weighted_merge_all(api, [[p["cons"][c] for p in allp] for c in range(len(cand))], [[p["cnt"][c] for p in allp] for c in range(len(cand))], band)
    T["consensus"] = T.get("consensus", 0.0) + time.perf_counter() - t0
    # ---- 5. reverse-complement merge (identical on every rank), then polish: per iteration local window consensus + weighted merge
    t0 = time.perf_counter()
    centers = [[int(sizes[g]), int(g), drafts[c], [c]] for c, g in enumerate(cand)]
    merged = pipeline.detect_reverse_complements(api, centers, rc_identity_threshold)
    polished = [m[2] for m in merged]
    if racon_iter > 0 and len(merged):
        # every shard polishes the (identical) merged drafts with its own reads, all iterations locally - the orientation, the alignments and
        # the window graphs never leave the GPU - then ONE all-gather of the polished strings and a weighted merge
        lists = pipeline.pooled_read_lists(merged, lambda c: order[lo[c]:hi[c]])
        p_off = np.concatenate(([0], np.cumsum([len(x) for x in lists])))
        p_order = np.concatenate(lists) if lists else np.zeros(0, np.uint32)
        bb = ReadSet.from_strings(polished)
        loc, used = api.polish(bb, rs_local, p_off, polish_params(iters=racon_iter, k=k, w=w, tile_depth=tile_depth, band=band, trim=polish_trim, stop_when_stable=polish_stop_when_stable), read_order=p_order)
        allq = comm.all_gather_obj(dict(cons=list(loc), cnt=np.asarray(used, dtype=np.int64)))
        mg = _weighted_merge_all(api, [[q["cons"][c] for q in allq] for c in range(len(merged))], [[q["cnt"][c] for q in allq] for c in range(len(merged))], band)
        polished = [mg[c] or polished[c] for c in range(len(merged))]
    T["polish"] = T.get("polish", 0.0) + time.perf_counter() - t0
    res["centers"] = [(m[0], m[1], m[2], polished[i]) for i, m in enumerate(merged)]
    return res
