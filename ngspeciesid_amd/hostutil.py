"""Small host helpers shared by the reference-shaped layer, the pipeline and the tests."""
from __future__ import annotations
import numpy as np
from ._capi import ReadSet


def acc_rank(accs):
    """Dense rank of the accession strings under byte-wise (= Python str) comparison, equal strings equal rank."""
    accs = list(accs)
    order = sorted(range(len(accs)), key=lambda i: accs[i])
    rank = np.zeros(len(accs), dtype=np.uint32)
    r = 0
    for j, i in enumerate(order):
        if j and accs[order[j - 1]] != accs[i]:
            r += 1
        rank[i] = r
    return rank


def subset_reads(rs: ReadSet, idx) -> ReadSet:
    """Gather reads `idx` of a HOST read set into a new contiguous host read set (record copies run in the library's multi-threaded host
    gather, csrc/host_io.hip; a million reads are memcpy-bound instead of building a 750 M-entry index array)."""
    import ctypes as C
    from . import runtime
    idx = np.asarray(idx, dtype=np.int64)
    off = rs.off.astype(np.int64)
    lens = off[idx + 1] - off[idx]
    noff = np.zeros(len(idx) + 1, dtype=np.uint64); noff[1:] = np.cumsum(lens)
    total = int(noff[-1])
    if len(idx) and np.all(np.diff(idx) == 1):           # contiguous slice: plain views
        a, b = int(off[idx[0]]), int(off[idx[-1] + 1])
        return ReadSet(rs.seq[a:b], None if rs.qual is None else rs.qual[a:b], noff)
    lib = runtime.load_library()
    so = np.ascontiguousarray(off[idx].astype(np.uint64)); ln = np.ascontiguousarray(lens.astype(np.uint32)); do = np.ascontiguousarray(noff[:-1])
    seq = np.empty(total, dtype=np.uint8)
    lib.ngsid_host_gather(rs.seq.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), C.c_uint64(len(idx)), seq.ctypes.data_as(C.c_void_p), do.ctypes.data_as(C.c_void_p))
    qual = None
    if rs.qual is not None:
        qual = np.empty(total, dtype=np.uint8)
        lib.ngsid_host_gather(rs.qual.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), C.c_uint64(len(idx)), qual.ctypes.data_as(C.c_void_p), do.ctypes.data_as(C.c_void_p))
    return ReadSet(seq, qual, noff)


def make_cluster_fn(api, rs: ReadSet, rank, prm):
    """cluster_fn for parallelize.tree_cluster backed by one C-ABI implementation (GPU library, or the oracle in tests)."""
    def fn(read_idx, prev_batch, known_err):
        sub = subset_reads(rs, read_idx)
        return api.cluster_greedy(sub, prm, acc_rank=rank[np.asarray(read_idx, dtype=np.int64)], prev_batch=prev_batch, known_err=known_err)
    return fn
