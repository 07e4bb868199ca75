"""(f2) FASTQ ingest and FASTQ / TSV egress on arrays: the whole file is one byte buffer, records are offset arrays, and the per-record work
(indexing, CSR gathers, record assembly) runs in the library's multi-threaded host helpers (csrc/host_io.hip).  Files that are not plain
4-line FASTQ (multi-line records, FASTA) go through the general reader of help_functions.readfq, which follows the reference's reader.

Replaces the per-record Python of help_functions.py:13-42, get_sorted_fastq_for_cluster.py:174-177, NGSpeciesID:99-120, consensus.py:203-215.
"""
from __future__ import annotations
import ctypes as C
import os
import numpy as np
from time import perf_counter as _perf
from . import runtime
from ._capi import ReadSet
from .help_functions import readfq


class Names:
    """accession strings as (byte buffer, offsets, lengths)"""

    def __init__(self, buf, off, length):
        self.buf = np.ascontiguousarray(buf, dtype=np.uint8); self.off = np.ascontiguousarray(off, dtype=np.uint64); self.len = np.ascontiguousarray(length, dtype=np.uint32)

    def __len__(self):
        return len(self.off)

    def get(self, i):
        a = int(self.off[i]); return self.buf[a:a + int(self.len[i])].tobytes().decode()

    def compact(self, idx):
        """the names idx, in that order, gathered into a buffer of their own"""
        idx = np.asarray(idx, dtype=np.int64)
        ln = np.ascontiguousarray(self.len[idx]); so = np.ascontiguousarray(self.off[idx])
        off = np.zeros(len(idx), dtype=np.uint64)
        if len(idx):
            off[1:] = np.cumsum(ln[:-1], dtype=np.uint64)
        buf = np.empty(int(ln.sum()) if len(idx) else 0, dtype=np.uint8)
        runtime.load_library().ngsid_host_gather(_p(self.buf), _p(so), _p(ln), C.c_uint64(len(idx)), _p(buf), _p(off))
        return Names(buf, off, ln)

    @staticmethod
    def from_list(strs):
        bs = [s.encode() for s in strs]
        ln = np.array([len(b) for b in bs], dtype=np.uint32)
        off = np.zeros(len(bs), dtype=np.uint64)
        if len(bs):
            off[1:] = np.cumsum(ln[:-1], dtype=np.uint64)
        return Names(np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8), off, ln)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def read_fastq(path):
    """-> (Names, ReadSet (host CSR), plain): plain = the file was a 4-line FASTQ handled by the array path."""
    lib = runtime.load_library()
    import os
    # the file is MAPPED, not copied (1.5 GB at C3: the copy alone took a third of the ingest).  Nothing keeps pointing into the mapping once this function returns
    # (the background writers used to read the names from it while `--fastq <outfolder>/sorted.fastq` was being truncated by those very writers): bases,
    # qualities AND names are gathered into owned arrays below and the mapping is dropped.  The input only has to stay unchanged for the duration of this call.
    # (round 5: mapped and unmapped by the library.  NumPy's memmap object unmapped the 1.5 GB of C3 on this thread with the interpreter lock held - 75 ms of the ingest's 130 ms -, and
    # a background unmap right after the ingest slows the upload that follows by as much (both want the address-space lock).  The mapping is therefore RELEASED LATER: by
    # release_mapped() - the CLI calls it when its run is over - or by the next read_fastq; until then nothing reads from it.)
    release_mapped()
    mptr = C.POINTER(C.c_ubyte)(); mlen = C.c_uint64(0)
    if lib.ngsid_host_map_file(os.fsencode(path), C.byref(mptr), C.byref(mlen)):
        raise OSError("cannot read %s" % path)
    buf = np.ctypeslib.as_array(mptr, shape=(mlen.value,)) if mlen.value else np.zeros(0, dtype=np.uint8)
    try:
        return _read_fastq_mapped(lib, path, buf)
    finally:
        del buf
        if mlen.value: _MAPPED.append((mptr, mlen))


_MAPPED = []


def release_mapped():
    """unmaps the input files read_fastq left mapped (a detached native thread does it: gigabytes take tens of milliseconds)"""
    lib = runtime.load_library()
    while _MAPPED:
        mptr, mlen = _MAPPED.pop()
        lib.ngsid_host_unmap_file(mptr, mlen, C.c_int32(1))


def _read_fastq_mapped(lib, path, buf):
    n = C.c_uint64(0)
    rc = lib.ngsid_host_fastq_index(_p(buf), C.c_uint64(len(buf)), None, None, None, C.c_uint64(0), C.byref(n))
    if rc == 0 and n.value > 0:
        nr = n.value
        rec = np.zeros(4 * nr, dtype=np.uint64); nlen = np.zeros(nr, dtype=np.uint32); slen = np.zeros(nr, dtype=np.uint32)
        rc = lib.ngsid_host_fastq_index(_p(buf), C.c_uint64(len(buf)), _p(rec), _p(nlen), _p(slen), C.c_uint64(nr), C.byref(n))
        if rc == 0:
            off = np.zeros(nr + 1, dtype=np.uint64); off[1:] = np.cumsum(slen, dtype=np.uint64)
            total = int(off[-1])
            seq = np.empty(total, dtype=np.uint8); qual = np.empty(total, dtype=np.uint8)
            so = np.ascontiguousarray(rec[1::4]); qo = np.ascontiguousarray(rec[3::4]); do = np.ascontiguousarray(off[:-1])
            lib.ngsid_host_gather(_p(buf), _p(so), _p(slen), C.c_uint64(nr), _p(seq), _p(do))
            lib.ngsid_host_gather(_p(buf), _p(qo), _p(slen), C.c_uint64(nr), _p(qual), _p(do))
            no = np.ascontiguousarray(rec[0::4]); noff = np.zeros(nr, dtype=np.uint64)
            if nr: noff[1:] = np.cumsum(nlen[:-1], dtype=np.uint64)
            nbuf = np.empty(int(nlen.sum(dtype=np.uint64)), dtype=np.uint8)
            lib.ngsid_host_gather(_p(buf), _p(no), _p(nlen), C.c_uint64(nr), _p(nbuf), _p(noff))      # (15 MB per million reads: 2 ms)
            return Names(nbuf, noff, nlen), ReadSet(seq, qual, off), True
    # general reader (multi-line FASTQ / FASTA / empty file)
    accs, seqs, quals = [], [], []
    with open(path) as f:
        for acc, (s, q) in readfq(f):
            accs.append(acc); seqs.append(s); quals.append(q if q is not None else "")
    if any(len(s) != len(q) for s, q in zip(seqs, quals)):
        raise ValueError("%s: records without (full-length) qualities - the clustering path needs FASTQ" % path)
    return Names.from_list(accs), ReadSet.from_strings(seqs, quals), False


def normalize_bases(seq: np.ndarray) -> int:
    """upper-case, non-ACGTN -> N, in place; returns the number of bytes changed"""
    lib = runtime.load_library()
    ch = C.c_uint64(0)
    lib.ngsid_host_normalize_bases(_p(seq), C.c_uint64(len(seq)), C.byref(ch))
    return int(ch.value)


def count_foreign_bases(seq: np.ndarray) -> int:
    ch = C.c_uint64(0)
    runtime.load_library().ngsid_host_count_foreign_bases(_p(seq), C.c_uint64(len(seq)), C.byref(ch))
    return int(ch.value)


def repr_doubles(values, prefix=""):
    """CSR (bytes, offsets) of CPython's repr(float) of every value, each preceded by `prefix` (one character or empty)"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    buf = np.empty(32 * len(v) + 32, dtype=np.uint8); off = np.zeros(len(v) + 1, dtype=np.uint64); need = C.c_uint64(0)
    rc = runtime.load_library().ngsid_host_repr_doubles(_p(v), C.c_uint64(len(v)), C.c_int32(ord(prefix) if prefix else 0), _p(buf), C.c_uint64(len(buf)), _p(off), C.byref(need))
    if rc:
        raise RuntimeError("ngsid_host_repr_doubles failed (%d)" % rc)
    return buf[:int(off[-1])].copy() if len(v) else np.zeros(1, np.uint8), off


def argsort_desc(values) -> np.ndarray:
    """stable argsort in descending order (the order of sorted.fastq: get_sorted_fastq_for_cluster.py:174) - ngsid_host_argsort_desc"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    order = np.empty(len(v), dtype=np.uint64)
    rc = runtime.load_library().ngsid_host_argsort_desc(_p(v), C.c_uint64(len(v)), _p(order))
    if rc != 0:
        raise RuntimeError("ngsid_host_argsort_desc failed (%d)" % rc)
    return order.astype(np.int64)


def list_positions(rep) -> np.ndarray:
    """pos[i] = number of earlier reads with the same representative - ngsid_host_list_positions"""
    r = np.ascontiguousarray(rep, dtype=np.int64)
    pos = np.empty(len(r), dtype=np.int64)
    rc = runtime.load_library().ngsid_host_list_positions(_p(r), C.c_uint64(len(r)), _p(pos))
    if rc != 0:
        raise ValueError("ngsid_host_list_positions: a representative index is out of range")
    return pos


import threading as _threading
_GBR_TLS = _threading.local()          # scratch of group_by_rep, reused between calls (fresh 8 MB arrays page-fault on every call while the GPU waits for the consensus stage); per thread
                                       # (the ctypes call releases the interpreter lock, virtual ranks are threads)


def group_by_rep(rep_of):
    """-> (reps, order, grp_off, counts) of a representative map - ngsid_host_group_by_rep / _rep32 (no widening copy for the int32 map of the clustering call)"""
    a = np.asarray(rep_of)
    r = np.ascontiguousarray(a) if a.dtype == np.int32 else np.ascontiguousarray(a, dtype=np.int64)
    n = len(r)
    _GBR = _GBR_TLS.__dict__
    if _GBR.get("n", -1) < n:
        _GBR.update(n=n, reps=np.empty(n, dtype=np.int64), counts=np.empty(n, dtype=np.int64), goff=np.empty(n + 1, dtype=np.uint64))
    reps, counts, goff = _GBR["reps"], _GBR["counts"], _GBR["goff"]
    order = np.empty(n, dtype=np.uint32)
    nr = C.c_uint64(0)
    lib = runtime.load_library()
    fn = lib.ngsid_host_group_by_rep32 if r.dtype == np.int32 and hasattr(lib, "ngsid_host_group_by_rep32") else None
    if fn is None:
        r = np.ascontiguousarray(r, dtype=np.int64); fn = lib.ngsid_host_group_by_rep
    rc = fn(_p(r), C.c_uint64(n), _p(reps), C.byref(nr), _p(order), _p(goff), _p(counts))
    if rc != 0:
        raise ValueError("ngsid_host_group_by_rep: not a representative map (an index out of range, or a representative that does not represent itself)")
    R = int(nr.value)
    return reps[:R].copy(), order, goff[:R + 1].copy(), counts[:R].copy()


def _csr(strs):
    bs = [s if isinstance(s, bytes) else s.encode() for s in strs]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return (np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(1, np.uint8)), off


class NativeJobs:
    """background record-writer jobs of the LIBRARY (ngsid_host_write_records_async): the job runs on native worker threads, this object keeps the arrays of every job alive
    until wait() has collected its result.  No interpreter thread is involved, so the thread that drives the GPU never meets a writer at the interpreter lock (round 5)."""
    def __init__(self):
        self.jobs = []                      # (job id, path, arrays kept alive)

    def wait(self):
        lib = runtime.load_library()
        jobs, self.jobs = self.jobs, []
        bad = []
        for jid, path, _ in jobs:                     # (waits for ALL of them before it reports)
            t0 = _perf()
            if lib.ngsid_host_async_wait(C.c_uint64(jid)) != 0: bad.append(path)
            if os.environ.get("NGSID_WRITE_TRACE"):
                import sys; sys.stderr.write("[wait] %s %.3f s\n" % (os.path.basename(path), _perf() - t0))
        # The arrays the jobs kept alive (among them the last reference to the 1.5 GB file-order copy of the reads at C3) are released LATER, by a helper thread: unmapping them
        # takes 0.1 - 0.17 s with the interpreter lock held, whichever thread does it (traced: every write had ended 0.13 s before the stages were done and the wait itself
        # took 0.000 s, yet the caller lost 0.165 s here; a helper thread that released them at once held the lock against the caller just as long).  Half a second later the
        # caller of a CLI run has returned (a process that ends before that never unmaps them at all).
        import threading, time as _time
        box = [jobs]; del jobs
        def _later(b=box):
            _time.sleep(0.5); b.clear()
        threading.Thread(target=_later, daemon=True).start(); del box
        if bad:
            raise OSError("cannot write %s" % ", ".join(bad))


def _write_records(path, append, kind, idx, names, first_token, sb, so, sfx_by_read, rs, jobs, threads=0):
    lib = runtime.load_library()
    seq, qual, off = (rs.seq, rs.qual, rs.off) if rs is not None else (None, None, None)
    args = (path.encode(), C.c_int32(int(append)), C.c_int32(kind), C.c_uint64(len(idx)), _p(idx), _p(names.buf), _p(names.off), _p(names.len),
            C.c_int32(int(first_token)), _p(sb), _p(so), C.c_int32(int(sfx_by_read)), _p(seq), _p(qual), _p(off))
    if jobs is None:
        if lib.ngsid_host_write_records(*args):
            raise OSError("cannot write %s" % path)
        return
    jid = C.c_uint64(0)
    if lib.ngsid_host_write_records_async(*args, C.c_int32(int(threads or 0)), C.byref(jid)):
        if lib.ngsid_host_write_records(*args):            # the queue could not take the job (no thread to be had): write it here
            raise OSError("cannot write %s" % path)
        return
    jobs.jobs.append((jid.value, path, (idx, names, sb, so, seq, qual, off, rs)))


def write_fastq(path, idx, names: Names, rs: ReadSet, suffixes=None, first_token=False, append=False, sfx_by_read=False, jobs: NativeJobs = None, threads=0):
    """FASTQ records of reads idx (in that order); a suffix is appended to each record's name (the '_score' of sorted.fastq): suffixes = list of
    strings or a CSR (bytes, offsets), one per OUTPUT record, or one per READ (indexed by idx[j]) with sfx_by_read.  jobs: run as a background job of the library."""
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    sb, so = (None, None) if suffixes is None else (suffixes if isinstance(suffixes, tuple) else _csr(suffixes))
    _write_records(path, append, 0, idx, names, first_token, sb, so, sfx_by_read, rs, jobs, threads)


def write_paf(path, idx, names: Names, rs_off, aln, tname, tlen, suffixes=None, jobs: NativeJobs = None, threads=0):
    """read_alignments_it_{i}.paf (consensus.py:112-121): one PAF line per read idx[j] with the record aln[j] = (strand, q_begin, q_end, t_begin, t_end, distance) of
    ngsid_polish_trace_aln (no line where strand < 0).  suffixes = per-READ CSR (bytes, offsets) appended to the name before it is cut at the first blank."""
    lib = runtime.load_library()
    idx = np.ascontiguousarray(idx, dtype=np.uint64); aln = np.ascontiguousarray(aln, dtype=np.int32).reshape(-1, 6); assert len(aln) == len(idx)
    off = np.ascontiguousarray(rs_off, dtype=np.uint64)
    sb, so = (None, None) if suffixes is None else suffixes
    args = (path.encode(), C.c_uint64(len(idx)), _p(idx), _p(names.buf), _p(names.off), _p(names.len), _p(sb), _p(so), _p(off), _p(aln), tname.encode(), C.c_uint32(int(tlen)), C.c_int32(int(threads or 0)))
    if jobs is not None:
        jid = C.c_uint64(0)
        if lib.ngsid_host_write_paf(*args, C.byref(jid)) == 0:
            jobs.jobs.append((jid.value, path, (idx, names, sb, so, off, aln))); return
    if lib.ngsid_host_write_paf(*args, None):
        raise OSError("cannot write %s" % path)


def write_tsv(path, idx, names: Names, prefixes, append=False, jobs: NativeJobs = None):
    """lines 'prefix<TAB>name' for reads idx; prefixes = (byte buffer, offsets) CSR or a list of strings, one per line."""
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    pb, po = prefixes if isinstance(prefixes, tuple) else _csr(prefixes)
    _write_records(path, append, 1, idx, names, False, pb, po, False, None, jobs)


def int_prefixes(values):
    """CSR of the decimal strings of an integer array (ngsid_host_int_prefixes: two passes in the host library; round 5 - the NumPy digit loops took 30 ms of the launch thread)."""
    values = np.ascontiguousarray(values, dtype=np.int64)
    if len(values) == 0:
        return np.zeros(1, np.uint8), np.zeros(1, dtype=np.uint64)
    lib = runtime.load_library()
    off = np.zeros(len(values) + 1, dtype=np.uint64); need = C.c_uint64(0)
    lib.ngsid_host_int_prefixes(_p(values), C.c_uint64(len(values)), None, C.c_uint64(0), _p(off), C.byref(need))
    buf = np.empty(max(int(need.value), 1), dtype=np.uint8)
    if lib.ngsid_host_int_prefixes(_p(values), C.c_uint64(len(values)), _p(buf), C.c_uint64(len(buf)), _p(off), C.byref(need)):
        raise ValueError("int_prefixes failed")
    return buf[:int(need.value)], off
