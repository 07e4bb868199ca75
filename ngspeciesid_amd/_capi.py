"""ctypes view of the C-ABI declared in include/ngsid.h (structs + call wrappers).

`Api` binds one shared library.  The product binds libngsid_hip.so (prefix ``ngsid_``, ctx first);
the tests bind the CPU oracle with the same wrapper (prefix ``ongsid_``, no ctx) - the oracle is
never bound from inside this package.
"""
from __future__ import annotations
import os
import ctypes as C
from time import perf_counter as _perf
import numpy as np

MEM_HOST, MEM_DEVICE = 0, 1
ERRORS = {-1: "NGSID_ERR_NO_DEVICE", -2: "NGSID_ERR_ARG", -3: "NGSID_ERR_ALPHABET", -4: "NGSID_ERR_CAPACITY",
          -5: "NGSID_ERR_HIP", -6: "NGSID_ERR_TOO_LONG", -7: "NGSID_ERR_NO_PTABLE"}
ST_NEWREP, ST_MAPPED, ST_ALIGNED, ST_SHORT, ST_SEEDED = 0, 1, 2, 3, 4
POA_LOCAL, POA_GLOBAL, POA_SEMI = 0, 1, 2


class NgsidError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "NGSID_ERR"), code, text))
        self.code = code


class Reads(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("qual", C.c_void_p), ("off", C.c_void_p), ("n", C.c_uint64), ("mem", C.c_uint32), ("_pad", C.c_uint32)]


class ClusterParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32), ("min_shared", C.c_int32), ("symmetric", C.c_int32),
                ("min_fraction", C.c_double), ("mapped_threshold", C.c_double), ("aligned_threshold", C.c_double),
                ("min_prob_no_hits", C.c_double), ("p_shared", C.c_double * 225)]


class PoaParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32),
                ("tile_depth", C.c_int32), ("band", C.c_int32), ("node_cap", C.c_int32), ("trim", C.c_int32), ("single_below", C.c_int32)]


class PolishParams(C.Structure):
    _fields_ = [("iters", C.c_int32), ("window", C.c_int32), ("quality_threshold", C.c_double), ("error_threshold", C.c_double),
                ("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32), ("k", C.c_int32), ("w", C.c_int32),
                ("tile_depth", C.c_int32), ("band", C.c_int32), ("node_cap", C.c_int32),
                ("aln_match", C.c_int32), ("aln_mismatch", C.c_int32), ("aln_open", C.c_int32), ("aln_ext", C.c_int32), ("trim", C.c_int32), ("aln_mode", C.c_int32), ("stop_when_stable", C.c_int32), ("single_below", C.c_int32)]


def cluster_params(k=13, w=20, min_shared=5, min_fraction=0.8, mapped_threshold=0.7, aligned_threshold=0.4,
                   min_prob_no_hits=0.1, symmetric=False, p_shared=None):
    p = ClusterParams()
    p.k, p.w, p.min_shared, p.symmetric = int(k), int(w), int(min_shared), int(bool(symmetric))
    p.min_fraction, p.mapped_threshold, p.aligned_threshold, p.min_prob_no_hits = min_fraction, mapped_threshold, aligned_threshold, min_prob_no_hits
    t = np.full(225, np.nan) if p_shared is None else np.asarray(p_shared, dtype=np.float64)
    for i in range(225):
        p.p_shared[i] = t[i]
    return p


def poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=0, band=0, node_cap=0, trim=0, single_below=0):
    """Defaults = `spoa -l 0 -r 0 -g -2` (consensus.py:87; spoa 4.0.x m=5 n=-4, linear because g>=e)."""
    return PoaParams(int(mode), int(match), int(mismatch), int(gap), int(tile_depth), int(band), int(node_cap), int(trim), int(single_below))


def polish_params(iters=2, window=500, quality_threshold=10.0, error_threshold=0.3, match=3, mismatch=-5, gap=-4,
                  k=13, w=20, tile_depth=0, band=0, node_cap=0, aln_match=2, aln_mismatch=-2, aln_open=3, aln_ext=1, trim=1, aln_mode=2, stop_when_stable=1, single_below=0):
    """Defaults = racon 1.4.x (-w 500 -q 10 -e 0.3 -m 3 -x -5 -g -4), iters = --racon_iter (NGSpeciesID:212)."""
    return PolishParams(int(iters), int(window), float(quality_threshold), float(error_threshold), int(match), int(mismatch), int(gap),
                        int(k), int(w), int(tile_depth), int(band), int(node_cap), int(aln_match), int(aln_mismatch), int(aln_open), int(aln_ext), int(trim), int(aln_mode), int(stop_when_stable), int(single_below))


class ReadSet:
    """CSR read set backed by numpy arrays (host) or by raw device pointers (torch tensors kept alive by the caller)."""

    def __init__(self, seq, qual, off, mem=MEM_HOST, keep=None):
        self.mem = mem
        self.keep = keep
        if mem == MEM_HOST:
            self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
            self.qual = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
            self.off = np.ascontiguousarray(off, dtype=np.uint64)
            self.n = len(self.off) - 1
            self.c = Reads(self.seq.ctypes.data, None if self.qual is None else self.qual.ctypes.data, self.off.ctypes.data, self.n, MEM_HOST, 0)
        else:
            self.seq, self.qual, self.off = seq, qual, off            # integers (device addresses)
            self.n = int(keep["n"])
            self.c = Reads(int(seq), None if qual is None else int(qual), int(off), self.n, MEM_DEVICE, 0)

    @staticmethod
    def from_strings(seqs, quals=None):
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        seq = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
        qual = None
        if quals is not None:
            qual = np.frombuffer("".join(quals).encode(), dtype=np.uint8).copy() if len(quals) else np.zeros(0, np.uint8)
        return ReadSet(seq, qual, off)

    @staticmethod
    def from_torch(seq_t, qual_t, off_t):
        """torch tensors on a cuda/hip device: uint8, uint8, int64 (n+1).  The library works on its own HIP stream, so torch's stream
        is drained first: whatever produced the tensors must have finished writing them before the hand-over."""
        import torch
        torch.cuda.current_stream(seq_t.device).synchronize()
        keep = dict(seq=seq_t, qual=qual_t, off=off_t, n=off_t.numel() - 1)
        return ReadSet(seq_t.data_ptr(), None if qual_t is None else qual_t.data_ptr(), off_t.data_ptr(), MEM_DEVICE, keep)

    def release(self):
        """device read set made by Api.upload_reads: give the buffers back (idempotent)"""
        k = self.keep if isinstance(self.keep, dict) else None
        if self.mem == MEM_DEVICE and k and k.get("api") is not None:
            api, k["api"] = k["api"], None
            api._call("reads_release", C.byref(self.c))

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def get(self, i):
        a, b = int(self.off[i]), int(self.off[i + 1])
        return self.seq[a:b].tobytes().decode(), (None if self.qual is None else self.qual[a:b].tobytes().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


LANES = int(os.environ.get("NGSID_LANES", "2"))           # contexts a consensus / polishing call is dealt to (Api.lanes); 1 = off
LANE_MIN_READS = 100_000                                   # below: one context (measured on the bench workload: 100 k reads 109.5 ms either way, 300 k reads 248 -> 242 ms, 1 M 719 -> 708 ms)
LANE_MAX_READS = 4_000_000                                 # above: one context (a second one doubles the grow-only scratch: 130 - 140 GB at 10 M reads)


def lane_deal(sizes, lanes):
    """groups -> lanes: largest first to the lightest lane; per lane the group numbers ascending"""
    load = [0] * lanes; deal = [[] for _ in range(lanes)]
    for g in sorted(range(len(sizes)), key=lambda g: (-int(sizes[g]), g)):
        x = int(np.argmin(load))
        deal[x].append(g); load[x] += int(sizes[g])
    return [sorted(d) for d in deal if d]


def _lane_lists(grp_off, read_order, gs):
    """(grp_off, read_order) of the groups gs of a call"""
    go = np.asarray(grp_off, dtype=np.int64)
    off = np.zeros(len(gs) + 1, dtype=np.uint64); off[1:] = np.cumsum([int(go[g + 1] - go[g]) for g in gs])
    if read_order is None:
        ro = np.concatenate([np.arange(go[g], go[g + 1], dtype=np.uint32) for g in gs])
    else:
        r = np.asarray(read_order); ro = np.concatenate([r[int(go[g]):int(go[g + 1])] for g in gs]).astype(np.uint32)
    return off, ro


def _lane_backbones(bb: "ReadSet", gs):
    o = bb.off.astype(np.int64)
    off = np.zeros(len(gs) + 1, dtype=np.uint64); off[1:] = np.cumsum([int(o[g + 1] - o[g]) for g in gs])
    seq = np.concatenate([bb.seq[int(o[g]):int(o[g + 1])] for g in gs]) if len(gs) else np.zeros(0, np.uint8)
    qual = None if bb.qual is None else np.concatenate([bb.qual[int(o[g]):int(o[g + 1])] for g in gs])
    return ReadSet(seq, qual, off)


class LaneRecords:
    """alignment records [it][x][6] of a polish_trace(aln=True) call that ran in lanes: rec[it][a:b] of ONE group's range (what the PAF writers take) is a view into that lane's
    array; anything else goes through the merged array, built on first use (72 MB per million reads and three iterations: not copied unless somebody asks)"""
    def __init__(self, go, iters, where):
        self.go, self.iters, self.where, self._all = go, iters, where, None
        self.shape = (iters, int(go[-1]), 6); self._start = {int(go[g]): g for g in range(len(go) - 1)}

    def __array__(self, dtype=None, copy=None):
        if self._all is None:
            self._all = np.empty(self.shape, dtype=np.int32)
            for g, (arr, pos, n) in enumerate(self.where): self._all[:, int(self.go[g]):int(self.go[g + 1])] = arr[:, pos:pos + n]
        return self._all if dtype is None else self._all.astype(dtype)

    def __len__(self):
        return self.iters

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)): return _LaneRecordsOfIteration(self, int(i))
        return np.asarray(self)[i]


class _LaneRecordsOfIteration:
    def __init__(self, parent, it): self.p, self.it = parent, it

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.p)[self.it]; return a if dtype is None else a.astype(dtype)

    def __getitem__(self, sl):
        p = self.p
        if isinstance(sl, slice) and sl.step in (None, 1) and sl.start is not None and sl.stop is not None:
            g = p._start.get(int(sl.start))
            if g is not None and int(p.go[g + 1]) == int(sl.stop):
                arr, pos, n = p.where[g]; return arr[self.it, pos:pos + n]
        return np.asarray(p)[self.it][sl]


class Api:
    def __init__(self, lib: C.CDLL, prefix: str, ctx=None):
        self.lib, self.prefix, self.ctx = lib, prefix, ctx
        self.has_ctx = ctx is not None

    def close(self):
        """ngsid_destroy: releases the context's stream, side streams, pinned staging and scratch buffers (and, with the last context of the
        process, the device-memory cache and the read sets uploaded through it).  Contexts from runtime.get_api() are shared and stay open;
        the ones from runtime.new_api() belong to their caller: close them (or use `with runtime.new_api() as api:`)."""
        for t, ex in self.__dict__.pop("_twin_list", []):
            ex.shutdown(wait=True); t.close()
        if self.has_ctx and self.ctx is not None:
            f = getattr(self.lib, self.prefix + "destroy"); f.restype = None
            f(self.ctx)
            self.ctx = None; self.has_ctx = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _fn(self, name):
        f = getattr(self.lib, self.prefix + name)
        f.restype = C.c_int32
        return f

    def _call(self, name, *args):
        f = self._fn(name)
        t0 = _perf()
        rc = f(self.ctx, *args) if self.has_ctx else f(*args)
        cs = self.__dict__.setdefault("call_s", {})        # wall time per entry point as THIS side sees it (ctypes call + the wait for the interpreter lock on return); the library's
        cs[name] = cs.get(name, 0.0) + _perf() - t0        # own clock for the same calls: "host_<name>" lines of ngsid_profile_read (bench.py config.cli.binding_overhead_s)
        return rc

    # ---- lanes (round 6): the groups of ONE consensus / polishing call are independent, so they are dealt to `lanes` contexts of the same device, each driven by its own long-lived
    # host thread on its own HIP stream.  One lane's kernels then run in the other's host gaps (the unit and tile lists between the aligner and the POA levels of every polishing
    # iteration: ~3 ms each, GPU idle) and under the latency-bound top levels of its hierarchy.  Same results group by group (tests run both ways).  lanes = 1: off.
    lanes = LANES

    def _lane_deal(self, rs, grp_off, backbones=None):
        """-> per lane the (ascending) group numbers, or None when the call runs in this context alone"""
        ng = len(grp_off) - 1
        if self.lanes < 2 or ng < 2 or not self.has_ctx or self.prefix != "ngsid_" or "device" not in self.__dict__ or rs.mem != MEM_DEVICE or (backbones is not None and backbones.mem != MEM_HOST):
            return None
        sizes = np.diff(np.asarray(grp_off, dtype=np.int64))
        if not (LANE_MIN_READS <= int(sizes.sum()) <= LANE_MAX_READS):
            return None
        deal = lane_deal(sizes, min(self.lanes, ng))
        try:
            self._twins(len(deal) - 1)
        except NgsidError:                # no second context to be had (memory): this one does the work, from now on
            self.lanes = 1
            return None
        return deal

    def set_option(self, name, value):
        """ngsid_ctx_option on this context and on its lane contexts (present and future)"""
        self.__dict__.setdefault("_options", {})[name] = int(value)
        for a in self.contexts():
            if a.lib.ngsid_ctx_option(a.ctx, name.encode(), C.c_int64(int(value))) != 0: a._err(-2)

    def _twins(self, n):
        tw = self.__dict__.setdefault("_twin_list", [])
        while len(tw) < n:
            from . import runtime
            from concurrent.futures import ThreadPoolExecutor
            # ONE long-lived thread per twin: the library keeps its pinned staging vectors per host thread; a fresh thread per call would allocate them again every time, and
            # releasing pinned memory waits for the whole device - i.e. for the other lane
            opts = dict(self.__dict__.get("_options", {}))
            opts.setdefault("scratch_budget_mb", 16384)          # half the default aligner-traceback budget (measured: same speed - the lanes share the device anyway -, C5 at 2 M reads 204 -> 177 GB with both contexts at 16 GB)
            tw.append((runtime.new_api(self.__dict__.get("device"), options=opts), ThreadPoolExecutor(max_workers=1, thread_name_prefix="ngsid-lane")))
            tw[-1][0].lanes = 1
        return tw[:n]

    def _lane_run(self, deal, fn):
        tw = self._twins(len(deal) - 1)
        futs = [tw[x - 1][1].submit(fn, tw[x - 1][0], deal[x]) for x in range(1, len(deal))]
        try:
            res = [fn(self, deal[0])]
        finally:
            excs = [f.exception() for f in futs]          # (waits for every lane, whatever happened here)
        for e in excs:
            if e is not None: raise e
        return res + [f.result() for f in futs]

    def _lanes_or_none(self, deal, fn):
        """the lanes' results, or None when a lane ran out of device memory (a second context's working set did not fit beside the first): the lane contexts give their
        memory back, this context does the call alone - now (the caller repeats it) and from now on"""
        try:
            return self._lane_run(deal, fn)
        except NgsidError as e:
            if "out of memory" not in str(e).lower(): raise
            self.lanes = 1
            for t, _ in self.__dict__.get("_twin_list", []):
                try: t.lib.ngsid_ctx_option(t.ctx, b"release_scratch", C.c_int64(1))
                except Exception: pass
            return None

    def contexts(self):
        """this context and the lane contexts it has made (profiling and statistics are per context)"""
        return [self] + [t[0] for t in self.__dict__.get("_twin_list", [])]

    def profile_enable(self, on: bool):
        """ngsid_profile_enable on this context and its lane contexts"""
        for a in self.contexts(): a.lib.ngsid_profile_enable(a.ctx, C.c_int32(1 if on else 0))

    def profile_read(self):
        """ngsid_profile_read of this context and its lane contexts -> ({name: (count, ms)} summed, [the same per context]).  Kernel lines are HIP-event brackets of every launch
        on its own stream: launches of two lanes that overlap on the device are both counted in full."""
        per = []
        for a in self.contexts():
            buf = C.create_string_buffer(1 << 16); a.lib.ngsid_profile_read(a.ctx, buf, C.c_uint64(len(buf)))
            d = {}
            for line in buf.value.decode().splitlines():
                nm, cnt, ms = line.split(); d[nm] = (int(cnt), float(ms))
            per.append(d)
        tot = {}
        for d in per:
            for nm, (c_, m_) in d.items():
                c0, m0 = tot.get(nm, (0, 0.0)); tot[nm] = (c0 + c_, m0 + m_)
        return tot, per

    def upload_reads(self, rs: "ReadSet") -> "ReadSet":
        """host read set -> device-resident read set (one PCIe copy for all the calls that follow); backends without the entry point
        (the test oracle) and read sets that are on the device already are returned as they are"""
        if rs.mem != MEM_HOST or not hasattr(self.lib, self.prefix + "reads_upload"):
            return rs
        out = Reads()
        rc = self._call("reads_upload", C.byref(rs.c), C.byref(out))
        if rc: self._err(rc)
        return ReadSet(out.seq, out.qual, out.off, MEM_DEVICE, dict(n=rs.n, api=self, host=rs))

    def reads_subset(self, dev: "ReadSet", idx):
        """reads idx (in that order) of a device-resident read set -> (new device-resident read set, number of bases outside ACGTN in it); None when the backend
        has no such entry point (the test oracle) or the set is on the host"""
        if dev.mem != MEM_DEVICE or not hasattr(self.lib, self.prefix + "reads_subset"):
            return None
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = Reads(); foreign = C.c_uint64(0)
        rc = self._call("reads_subset", C.byref(dev.c), _p(idx), C.c_uint64(len(idx)), C.byref(out), C.byref(foreign))
        if rc: self._err(rc)
        return ReadSet(out.seq, out.qual, out.off, MEM_DEVICE, dict(n=len(idx), api=self)), int(foreign.value)

    def _err(self, rc):
        if self.has_ctx:
            g = getattr(self.lib, self.prefix + "last_error"); g.restype = C.c_char_p
            txt = g(self.ctx)
        else:
            g = getattr(self.lib, self.prefix + "last_error"); g.restype = C.c_char_p
            txt = g()
        raise NgsidError(rc, (txt or b"").decode(errors="replace"))

    # ---- (f1)
    def score_reads(self, rs: ReadSet, k, q_threshold=7.0):
        n = rs.n
        score = np.zeros(n); err = np.zeros(n); keep = np.zeros(n, dtype=np.uint8)
        rc = self._call("score_reads", C.byref(rs.c), C.c_int32(k), C.c_double(q_threshold), _p(score), _p(err), _p(keep))
        if rc: self._err(rc)
        return score, err, keep

    # ---- (a1-a3)
    def hpc_minimizers(self, rs: ReadSet, k, w, cap=None):
        assert rs.mem == MEM_HOST, "host wrapper; device callers use the raw C-ABI"
        n = rs.n
        if cap is None:
            cap = max(16, int(len(rs.seq)))
        moff = np.zeros(n + 1, dtype=np.uint64); codes = np.zeros(cap, dtype=np.uint64); pos = np.zeros(cap, dtype=np.uint32)
        hl = np.zeros(n, dtype=np.uint32); he = np.zeros(n); needed = C.c_uint64(0)
        rc = self._call("hpc_minimizers", C.byref(rs.c), C.c_int32(k), C.c_int32(w), _p(moff), _p(codes), _p(pos), C.c_uint64(cap), C.byref(needed), _p(hl), _p(he))
        if rc: self._err(rc)
        t = int(moff[-1])
        return moff, codes[:t], pos[:t], hl, he

    # ---- (a4-a11)
    def cluster_greedy(self, rs: ReadSet, prm: ClusterParams, acc_rank=None, prev_batch=None, known_err=None):
        n = rs.n
        rep = np.zeros(n, dtype=np.int32); herr = np.zeros(n); st = np.zeros(n, dtype=np.uint8); cnt = np.zeros(4, dtype=np.uint64)
        ar = None if acc_rank is None else np.ascontiguousarray(acc_rank, dtype=np.uint32)
        pb = None if prev_batch is None else np.ascontiguousarray(prev_batch, dtype=np.int32)
        ke = None if known_err is None else np.ascontiguousarray(known_err, dtype=np.float64)
        rc = self._call("cluster_greedy", C.byref(rs.c), C.byref(prm), _p(ar), _p(pb), _p(ke), _p(rep), _p(herr), _p(st), _p(cnt))
        if rc: self._err(rc)
        return rep, herr, st, cnt

    def merge_representatives(self, reps: ReadSet, prm: ClusterParams, score, hpc_err, batch, n_batches, acc_rank=None):
        """merge rounds of parallel_clustering on gathered representatives -> rep_of [R] (index of the final representative)"""
        R = reps.n
        rep = np.zeros(R, dtype=np.int32)
        sc = np.ascontiguousarray(score, dtype=np.float64); he = np.ascontiguousarray(hpc_err, dtype=np.float64); bt = np.ascontiguousarray(batch, dtype=np.int32)
        ar = None if acc_rank is None else np.ascontiguousarray(acc_rank, dtype=np.uint32)
        rc = self._call("merge_representatives", C.byref(reps.c), C.byref(prm), _p(ar), _p(sc), _p(he), _p(bt), C.c_int32(int(n_batches)), _p(rep))
        if rc: self._err(rc)
        return rep

    # ---- (a10,a15)
    def sg_align_batch(self, q: ReadSet, t: ReadSet, q_idx, t_idx, open_, ext=1, match=2, mismatch=-2, k=13, match_id=None):
        q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32); t_idx = np.ascontiguousarray(t_idx, dtype=np.uint32)
        n = len(q_idx)
        open_ = np.ascontiguousarray(np.broadcast_to(np.asarray(open_, dtype=np.int32), (n,)))
        mid = np.ascontiguousarray(np.broadcast_to(np.asarray(k if match_id is None else match_id, dtype=np.int32), (n,)))
        score = np.zeros(n, dtype=np.int32); ncols = np.zeros(n, dtype=np.int32); nmatch = np.zeros(n, dtype=np.int32); region = np.zeros(n, dtype=np.int32)
        rc = self._call("sg_align_batch", C.byref(q.c), C.byref(t.c), _p(q_idx), _p(t_idx), C.c_uint64(n), C.c_int32(match), C.c_int32(mismatch),
                        _p(open_), C.c_int32(ext), C.c_int32(k), _p(mid), _p(score), _p(ncols), _p(nmatch), _p(region))
        if rc: self._err(rc)
        return score, ncols, nmatch, region

    def sg_align_cigar_batch(self, q: ReadSet, t: ReadSet, q_idx, t_idx, open_, ext=1, match=2, mismatch=-2):
        """-> (score [n], list of column strings over '=XID' in alignment order, one per pair): the alignment parasail returns as a CIGAR"""
        q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32); t_idx = np.ascontiguousarray(t_idx, dtype=np.uint32)
        n = len(q_idx)
        open_ = np.ascontiguousarray(np.broadcast_to(np.asarray(open_, dtype=np.int32), (n,)))
        ql = np.diff(q.off.astype(np.int64)); tl = np.diff(t.off.astype(np.int64))
        cap = int((ql[q_idx] + tl[t_idx]).sum()) + 1 if n else 1
        score = np.zeros(n, dtype=np.int32); off = np.zeros(n + 1, dtype=np.uint64); ops = np.zeros(cap, dtype=np.uint8); needed = C.c_uint64(0)
        rc = self._call("sg_align_cigar_batch", C.byref(q.c), C.byref(t.c), _p(q_idx), _p(t_idx), C.c_uint64(n), C.c_int32(match), C.c_int32(mismatch),
                        _p(open_), C.c_int32(ext), _p(score), _p(off), _p(ops), C.c_uint64(cap), C.byref(needed))
        if rc: self._err(rc)
        return score, [ops[int(off[p]):int(off[p + 1])].tobytes().decode() for p in range(n)]

    def ed_align_batch(self, q: ReadSet, t: ReadSet, q_idx, t_idx, window=500, bp_windows=0):
        """edit-distance (read inside backbone) alignment of the polisher: distance, span[n,4], bp[n,bp_windows,4]"""
        q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32); t_idx = np.ascontiguousarray(t_idx, dtype=np.uint32)
        n = len(q_idx)
        dist = np.zeros(n, dtype=np.int32); span = np.zeros((n, 4), dtype=np.int32); bp = np.zeros((n, max(bp_windows, 1), 4), dtype=np.int32)
        rc = self._call("ed_align_batch", C.byref(q.c), C.byref(t.c), _p(q_idx), _p(t_idx), C.c_uint64(n), C.c_int32(window), C.c_int32(bp_windows),
                        _p(dist), _p(span), _p(bp) if bp_windows else None)
        if rc: self._err(rc)
        return dist, span, bp[:, :bp_windows]

    # ---- (a13,a14)
    def poa_consensus(self, rs: ReadSet, grp_off, prm: PoaParams, cap=None, read_order=None):
        deal = self._lane_deal(rs, grp_off) if cap is None else None
        if deal is None:
            return self._poa_consensus1(rs, grp_off, prm, cap, read_order)
        def one(a, gs):
            off, ro = _lane_lists(grp_off, read_order, gs)
            return a._poa_consensus1(rs, off, prm, None, ro)
        parts = self._lanes_or_none(deal, one)
        if parts is None: return self._poa_consensus1(rs, grp_off, prm, cap, read_order)
        out = [None] * (len(grp_off) - 1)
        for gs, r in zip(deal, parts):
            for x, g in enumerate(gs): out[g] = r[x]
        return out

    def _poa_consensus1(self, rs: ReadSet, grp_off, prm: PoaParams, cap=None, read_order=None):
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        ro = None if read_order is None else np.ascontiguousarray(read_order, dtype=np.uint32)
        ng = len(grp_off) - 1
        if cap is None:
            lens = np.diff(rs.off.astype(np.int64)) if rs.mem == MEM_HOST else None
            cap = int(4 * (lens.max() if lens is not None and len(lens) else 16384) * max(ng, 1) + 1024)
        coff = np.zeros(ng + 1, dtype=np.uint64); cons = np.zeros(cap, dtype=np.uint8); needed = C.c_uint64(0)
        rc = self._call("poa_consensus", C.byref(rs.c), _p(ro), _p(grp_off), C.c_uint64(ng), C.byref(prm), _p(coff), _p(cons), C.c_uint64(cap), C.byref(needed))
        if rc: self._err(rc)
        return [cons[int(coff[g]):int(coff[g + 1])].tobytes().decode() for g in range(ng)]

    def poa_consensus_weighted(self, rs: ReadSet, grp_off, prm: PoaParams, weight, cap=None, read_order=None):
        """ngsid_poa_consensus_weighted: sequence i stands for weight[i] reads (the merge of per-shard partial consensuses)"""
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        ro = None if read_order is None else np.ascontiguousarray(read_order, dtype=np.uint32)
        wv = np.ascontiguousarray(weight, dtype=np.uint32); assert len(wv) == rs.n
        ng = len(grp_off) - 1
        if cap is None:
            lens = np.diff(rs.off.astype(np.int64)) if rs.mem == MEM_HOST else None
            cap = int(4 * (lens.max() if lens is not None and len(lens) else 16384) * max(ng, 1) + 1024)
        coff = np.zeros(ng + 1, dtype=np.uint64); cons = np.zeros(cap, dtype=np.uint8); needed = C.c_uint64(0)
        rc = self._call("poa_consensus_weighted", C.byref(rs.c), _p(ro), _p(grp_off), C.c_uint64(ng), C.byref(prm), _p(wv), _p(coff), _p(cons), C.c_uint64(cap), C.byref(needed))
        if rc: self._err(rc)
        return [cons[int(coff[g]):int(coff[g + 1])].tobytes().decode() for g in range(ng)]

    def poa_consensus_cov(self, rs: ReadSet, grp_off, prm: PoaParams, cap=None, read_order=None):
        """-> [(consensus string, uint32 coverage array)] per group"""
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        ro = None if read_order is None else np.ascontiguousarray(read_order, dtype=np.uint32)
        ng = len(grp_off) - 1
        if cap is None:
            lens = np.diff(rs.off.astype(np.int64)) if rs.mem == MEM_HOST else None
            cap = int(4 * (lens.max() if lens is not None and len(lens) else 16384) * max(ng, 1) + 1024)
        coff = np.zeros(ng + 1, dtype=np.uint64); cons = np.zeros(cap, dtype=np.uint8); cov = np.zeros(cap, dtype=np.uint32); needed = C.c_uint64(0)
        rc = self._call("poa_consensus_cov", C.byref(rs.c), _p(ro), _p(grp_off), C.c_uint64(ng), C.byref(prm), _p(coff), _p(cons), _p(cov), C.c_uint64(cap), C.byref(needed))
        if rc: self._err(rc)
        return [(cons[int(coff[g]):int(coff[g + 1])].tobytes().decode(), cov[int(coff[g]):int(coff[g + 1])].copy()) for g in range(ng)]

    # ---- (a16,a17)
    def polish(self, backbones: ReadSet, rs: ReadSet, grp_off, prm: PolishParams, cap=None, read_order=None):
        deal = self._lane_deal(rs, grp_off, backbones) if cap is None else None
        if deal is None:
            return self._polish1(backbones, rs, grp_off, prm, cap, read_order)
        def one(a, gs):
            off, ro = _lane_lists(grp_off, read_order, gs)
            return a._polish1(_lane_backbones(backbones, gs), rs, off, prm, None, ro)
        parts = self._lanes_or_none(deal, one)
        if parts is None: return self._polish1(backbones, rs, grp_off, prm, cap, read_order)
        ng = len(grp_off) - 1; seqs = [None] * ng; used = np.zeros(ng, dtype=np.uint64)
        for gs, (sq, us) in zip(deal, parts):
            for x, g in enumerate(gs): seqs[g] = sq[x]; used[g] = us[x]
        return seqs, used

    def _polish1(self, backbones: ReadSet, rs: ReadSet, grp_off, prm: PolishParams, cap=None, read_order=None):
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        ro = None if read_order is None else np.ascontiguousarray(read_order, dtype=np.uint32)
        ng = len(grp_off) - 1
        if cap is None:
            cap = int(4 * len(backbones.seq) + 4096) if backbones.mem == MEM_HOST else 1 << 24
        ooff = np.zeros(ng + 1, dtype=np.uint64); out = np.zeros(cap, dtype=np.uint8); needed = C.c_uint64(0); used = np.zeros(max(ng, 1), dtype=np.uint64)
        rc = self._call("polish", C.byref(backbones.c), C.byref(rs.c), _p(ro), _p(grp_off), C.c_uint64(ng), C.byref(prm), _p(ooff), _p(out), C.c_uint64(cap), C.byref(needed), _p(used))
        if rc: self._err(rc)
        return [out[int(ooff[g]):int(ooff[g + 1])].tobytes().decode() for g in range(ng)], used[:ng]

    def polish_trace(self, backbones: ReadSet, rs: ReadSet, grp_off, prm: PolishParams, cap=None, read_order=None, aln=False):
        """ngsid_polish_trace: -> (seqs[it][g], used[it][g]): every group's sequence after every iteration (the last one is what polish() returns).
        aln=True (ngsid_polish_trace_aln): -> (seqs, used, aln[it][x] = (strand, q_begin, q_end, t_begin, t_end, distance) of listed read x) - the PAF records of every iteration."""
        deal = self._lane_deal(rs, grp_off, backbones) if cap is None else None
        if deal is None:
            return self._polish_trace1(backbones, rs, grp_off, prm, cap, read_order, aln)
        def one(a, gs):
            off, ro = _lane_lists(grp_off, read_order, gs)
            return a._polish_trace1(_lane_backbones(backbones, gs), rs, off, prm, None, ro, aln)
        go = np.asarray(grp_off, dtype=np.int64); ng = len(go) - 1; iters = int(prm.iters)
        parts = self._lanes_or_none(deal, one)
        if parts is None: return self._polish_trace1(backbones, rs, grp_off, prm, cap, read_order, aln)
        seqs = [[None] * ng for _ in range(iters)]; used = np.zeros((iters, ng), dtype=np.uint64); where = [None] * ng
        for gs, r in zip(deal, parts):
            pos = 0
            for x, g in enumerate(gs):
                n = int(go[g + 1] - go[g])
                for it in range(iters): seqs[it][g] = r[0][it][x]
                used[:, g] = r[1][:, x]
                if aln: where[g] = (r[2], pos, n)
                pos += n
        return (seqs, used, LaneRecords(go, iters, where)) if aln else (seqs, used)

    def _polish_trace1(self, backbones: ReadSet, rs: ReadSet, grp_off, prm: PolishParams, cap=None, read_order=None, aln=False):
        grp_off = np.ascontiguousarray(grp_off, dtype=np.uint64)
        ro = None if read_order is None else np.ascontiguousarray(read_order, dtype=np.uint32)
        ng = len(grp_off) - 1; iters = int(prm.iters); n = iters * ng
        if cap is None:
            cap = (int(4 * len(backbones.seq) + 4096) if backbones.mem == MEM_HOST else 1 << 24) * max(iters, 1)
        ooff = np.zeros(n + 1, dtype=np.uint64); out = np.zeros(cap, dtype=np.uint8); needed = C.c_uint64(0); used = np.zeros(max(n, 1), dtype=np.uint64)
        if aln:
            nl = int(grp_off[-1]); rec = np.empty((iters, nl, 6), dtype=np.int32)
            rc = self._call("polish_trace_aln", C.byref(backbones.c), C.byref(rs.c), _p(ro), _p(grp_off), C.c_uint64(ng), C.byref(prm), _p(ooff), _p(out), C.c_uint64(cap), C.byref(needed), _p(used), _p(rec))
        else:
            rc = self._call("polish_trace", C.byref(backbones.c), C.byref(rs.c), _p(ro), _p(grp_off), C.c_uint64(ng), C.byref(prm), _p(ooff), _p(out), C.c_uint64(cap), C.byref(needed), _p(used))
        if rc: self._err(rc)
        seqs = [[out[int(ooff[it * ng + g]):int(ooff[it * ng + g + 1])].tobytes().decode() for g in range(ng)] for it in range(iters)]
        u = used[:n].reshape(iters, ng) if n else used[:0].reshape(0, ng)
        return (seqs, u, rec) if aln else (seqs, u)

