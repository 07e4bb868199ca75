"""(f3) Strand-aware clustering as an OPTION (`--strand_aware`; off by default, so the default output stays the reference's).

The reference clusters strand-unaware (cluster.py:16-39 takes the minimizers of the read as it is), so the forward and the reverse-complement
reads of one amplicon end in two clusters, get two draft consensus sequences, and are only joined after the drafts exist, by aligning the
drafts pairwise in both orientations (consensus.detect_reverse_complements, consensus.py:148-183).  With this option the join happens at the
clustering level, before any consensus work:

  1. the greedy clustering runs unchanged;
  2. the representatives of the clusters (largest first) are put through ONE more clustering call in which they seed the database as they are
     and their REVERSE COMPLEMENTS are clustered against them with the merge-round semantics of the reference (cluster.py:221-223,243-248:
     lower-batch representatives are fixed, higher-batch ones are re-clustered) - so "cluster j is the reverse complement of cluster t" is
     decided by exactly the criteria that decide every other membership (shared minimizers + mapped fraction, else block alignment);
  3. clusters joined that way become one cluster (the representative that comes first in the processing order stays), the reads of the
     flipped clusters are reverse-complemented for the consensus stages, and final_clusters.tsv shows ONE cluster per amplicon.

Everything goes through the same C-ABI calls as the rest of the path (ngsid_cluster_greedy), so the oracle backend reproduces it bit for bit.
The greedy pass itself stays strand-unaware: a reverse-complement read never joins a forward cluster directly, its cluster does right after.
"""
from __future__ import annotations
import numpy as np
from ._capi import ReadSet, MEM_HOST

_COMP = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def fetch_reads(rs: ReadSet, idx) -> ReadSet:
    """reads idx of a host OR device-resident read set as a host read set"""
    from .hostutil import subset_reads
    idx = np.asarray(idx, dtype=np.int64)
    if rs.mem == MEM_HOST:
        return subset_reads(rs, idx)
    k = rs.keep if isinstance(rs.keep, dict) else {}
    if k.get("host") is not None:                         # made by Api.upload_reads: the host copy is still there
        return subset_reads(k["host"], idx)
    import torch
    seq_t, qual_t, off_t = k["seq"], k.get("qual"), k["off"]
    offs = off_t.cpu().numpy().astype(np.int64)
    lens = offs[idx + 1] - offs[idx]
    noff = np.zeros(len(idx) + 1, dtype=np.uint64); noff[1:] = np.cumsum(lens)
    gi = torch.from_numpy(np.repeat(offs[idx] - noff[:-1].astype(np.int64), lens) + np.arange(int(noff[-1]), dtype=np.int64)).to(seq_t.device)
    return ReadSet(seq_t[gi].cpu().numpy(), None if qual_t is None else qual_t[gi].cpu().numpy(), noff)


def reverse_complement_reads(rs: ReadSet) -> ReadSet:
    """host read set -> every read reverse-complemented (qualities reversed), same offsets"""
    seq = np.empty_like(rs.seq); qual = None if rs.qual is None else np.empty_like(rs.qual)
    off = rs.off.astype(np.int64)
    for i in range(rs.n):
        a, b = int(off[i]), int(off[i + 1])
        seq[a:b] = _COMP[rs.seq[a:b][::-1]]
        if qual is not None: qual[a:b] = rs.qual[a:b][::-1]
    return ReadSet(seq, qual, rs.off)


def orient_reads(rs: ReadSet, flip) -> ReadSet:
    """reads with flip[i] != 0 reverse-complemented (qualities reversed), the others copied; host in -> host out, torch-backed device in -> device out.
    Chunked index arithmetic (no Python loop over the reads)."""
    import torch
    flip = np.asarray(flip).astype(bool)
    if not flip.any():
        return rs
    if rs.mem == MEM_HOST:
        seq_t = torch.from_numpy(rs.seq); qual_t = None if rs.qual is None else torch.from_numpy(rs.qual); off_t = torch.from_numpy(rs.off.astype(np.int64))
    else:
        k = rs.keep if isinstance(rs.keep, dict) else {}
        if k.get("seq") is None:
            # a read set made by Api.upload_reads keeps its host copy and the Api it was uploaded through: orient the host copy, upload the result
            if k.get("host") is not None and k.get("api") is not None:
                return k["api"].upload_reads(orient_reads(k["host"], flip))
            raise ValueError("orient_reads needs a host read set, a torch-backed device read set or one made by Api.upload_reads")
        seq_t, qual_t, off_t = k["seq"], k.get("qual"), k["off"]
    dev = seq_t.device
    comp = torch.from_numpy(_COMP).to(dev)
    out_s = seq_t.clone(); out_q = None if qual_t is None else qual_t.clone()
    fidx = torch.from_numpy(np.nonzero(flip)[0]).to(dev)
    CH = 1 << 15
    for c0 in range(0, len(fidx), CH):
        ids = fidx[c0:c0 + CH]
        a = off_t[ids]; b = off_t[ids + 1]; ln = b - a
        tot = int(ln.sum().item())
        if tot == 0: continue
        start = torch.cumsum(ln, 0) - ln
        rel = torch.arange(tot, device=dev) - torch.repeat_interleave(start, ln)
        dst = torch.repeat_interleave(a, ln) + rel
        src = torch.repeat_interleave(b - 1, ln) - rel
        out_s[dst] = comp[seq_t[src].long()]
        if out_q is not None: out_q[dst] = qual_t[src]
    if rs.mem == MEM_HOST:
        return ReadSet(out_s.numpy(), None if out_q is None else out_q.numpy(), rs.off)
    return ReadSet.from_torch(out_s, out_q, off_t)


class _ParityDSU:
    def __init__(self, n):
        self.p = list(range(n)); self.par = [0] * n

    def find(self, x):
        path = []
        while self.p[x] != x:
            path.append(x); x = self.p[x]
        root = x
        # compress: parity to the root = xor along the path
        acc = 0
        for y in reversed(path):
            acc ^= self.par[y]; self.par[y] = acc; self.p[y] = root
        return root

    def parity(self, x):
        self.find(x); return self.par[x] if self.p[x] != x else 0

    def union(self, a, b, rel, key):
        """orientation(a) = orientation(b) xor rel; the root with the smaller key stays the root"""
        ra, rb = self.find(a), self.find(b)
        if ra == rb:
            return False
        pa, pb = self.parity(a), self.parity(b)
        if key[rb] < key[ra]:
            ra, rb = rb, ra
        self.p[rb] = ra; self.par[rb] = pa ^ pb ^ rel
        return True


def strand_merge(api, rs: ReadSet, rep_of, score, prm, min_size=2, pos=None, max_clusters=4096):
    """-> (rep_of_new, flip [n] bool, pos_new or None, info).  rep_of: representative read of every read (itself for representatives);
    pos: optional position of every read in its cluster's read list (kept consistent: a joining cluster's list is appended, cluster.py:338-345)."""
    rep_of = np.asarray(rep_of, dtype=np.int64)
    n = len(rep_of)
    flip = np.zeros(n, dtype=bool)
    reps, sizes = np.unique(rep_of, return_counts=True)
    sel = sizes >= max(int(min_size), 1)
    cand, csz = reps[sel], sizes[sel]
    o = np.lexsort((cand, -np.asarray(score, dtype=np.float64)[cand], -csz))[:max_clusters]          # (size, score) descending = the order the consensus stage uses
    cand, csz = cand[o], csz[o]
    m = len(cand)
    info = dict(candidates=int(m), merged=0)
    if m < 2:
        return rep_of, flip, pos, info
    fw = fetch_reads(rs, cand); rc = reverse_complement_reads(fw)
    both = ReadSet(np.concatenate([fw.seq, rc.seq]), None if fw.qual is None else np.concatenate([fw.qual, rc.qual]),
                   np.concatenate([fw.off, fw.off[-1] + rc.off[1:]]).astype(np.uint64))
    prev = np.concatenate([np.full(m, 1, dtype=np.int32), np.full(m, 2, dtype=np.int32)])
    rep_l, _, _, _ = api.cluster_greedy(both, prm, acc_rank=np.arange(2 * m, dtype=np.uint32), prev_batch=prev)
    rep_l = np.asarray(rep_l, dtype=np.int64)
    dsu = _ParityDSU(m); key = cand.tolist()
    for j in range(m):
        t = int(rep_l[m + j])
        if t < m and t != j:                       # rc(representative j) joined the forward representative t
            if dsu.union(j, t, 1, key): info["merged"] += 1
    if info["merged"] == 0:
        return rep_of, flip, pos, info
    root_of = np.array([dsu.find(c) for c in range(m)], dtype=np.int64)
    par = np.array([dsu.parity(c) for c in range(m)], dtype=bool)
    # the lists of the joining clusters are appended to the root's list in candidate order (the larger cluster first)
    base = np.zeros(m, dtype=np.int64); grown = {}
    for c in range(m):
        r = int(root_of[c])
        if r == c: continue
        base[c] = grown.get(r, int(csz[r])); grown[r] = int(base[c]) + int(csz[c])
    cand_of_cluster = np.full(len(reps), -1, dtype=np.int64)
    cand_of_cluster[np.nonzero(sel)[0][o]] = np.arange(m)
    c_of_read = cand_of_cluster[np.searchsorted(reps, rep_of)]
    moved = (c_of_read >= 0) & (root_of[np.maximum(c_of_read, 0)] != np.maximum(c_of_read, 0))
    cm = c_of_read[moved]
    new_rep = rep_of.copy(); new_rep[moved] = cand[root_of[cm]]
    flip[moved] = par[cm]
    new_pos = None
    if pos is not None:
        new_pos = np.asarray(pos, dtype=np.int64).copy(); new_pos[moved] += base[cm]
    return new_rep, flip, new_pos, info
