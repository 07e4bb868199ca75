"""The drop-in CLI path on arrays: FASTQ in -> the reference's output files out, with ONE clustering schedule, ONE draft-consensus call and ONE
polishing call on a resident read set (no per-read Python, no temporary per-cluster files that are read back).

Same steps, same file contracts and the same results as the reference's main (NGSpeciesID:36-152) and as the reference-shaped dict layer of
this package (cluster.reads_to_clusters, consensus.run_spoa / run_racon, ... - kept for callers of the reference's modules, driven by tests/dict_layer.py):
  get_sorted_fastq_for_cluster.main (:124-191)  -> score_and_sort()     sorted.fastq, logfile.txt
  length filter / sub-sampling (NGSpeciesID:54-63)
  single_clustering / parallel_clustering      -> cluster()            (parallelize.tree_cluster for --t N: same batches, same merge rounds)
  writers (NGSpeciesID:99-120)                 -> write_cluster_files() final_clusters.tsv, final_cluster_origins.tsv
  form_draft_consensus / detect_reverse_complements / polish_sequences (consensus.py:249-278,148-183,186-246) -> consensus_and_polish()
                                                                       consensus_reference_*.fasta, reads_to_consensus_*.fastq, racon_cl_id_*/consensus.fasta
"""
from __future__ import annotations
import glob
import logging
import os
import random
import shutil
from time import time
import numpy as np
from . import runtime, fastio, parallelize, pipeline
from . import consensus as consensus_mod
from ._capi import ReadSet, cluster_params, poa_params, polish_params, POA_LOCAL
from .hostutil import subset_reads
from .ptable import select_p_table

MAX_READ_LEN = 65535          # NGSID_MAX_READ_LEN (include/ngsid.h): scoring, minimizers, clustering
MAX_CONSENSUS_LEN = 13107     # local-mode POA at the reference's match = 5 (5 x length < 65 536, include/ngsid.h NGSID_MAX_CONSENSUS_LEN)


def _repr_floats(x):
    return [repr(v) for v in np.asarray(x, dtype=np.float64).tolist()]


def _string_ranks(names: fastio.Names, sfx, idx):
    """dense rank of the strings name[i] + suffix[i], i in idx, under byte-wise comparison (= Python str order for UTF-8), equal strings equal rank:
    the accession tie-break of cluster.py:79,174.  Fixed-width byte rows + one numpy sort instead of a Python sort of a million strings."""
    import ctypes as C
    n = len(idx)
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    sb, so = sfx
    slen = (so[1:] - so[:-1]).astype(np.uint32)[idx]
    nlen = names.len[idx]
    W = int(nlen.max()) + int(slen.max()) + 1
    buf = np.zeros(n * W, dtype=np.uint8)
    lib = runtime.load_library()
    row = (np.arange(n, dtype=np.uint64) * np.uint64(W))
    lib.ngsid_host_gather(fastio._p(names.buf), fastio._p(np.ascontiguousarray(names.off[idx])), fastio._p(np.ascontiguousarray(nlen)), C.c_uint64(n), fastio._p(buf), fastio._p(row))
    lib.ngsid_host_gather(fastio._p(sb), fastio._p(np.ascontiguousarray(so[:-1][idx])), fastio._p(np.ascontiguousarray(slen)), C.c_uint64(n), fastio._p(buf), fastio._p(row + nlen.astype(np.uint64)))
    rows = buf.view("S%d" % W)
    o = np.argsort(rows, kind="stable")
    srt = rows[o]
    new = np.ones(n, dtype=bool); new[1:] = srt[1:] != srt[:-1]
    rank = np.empty(n, dtype=np.uint32); rank[o] = (np.cumsum(new) - 1).astype(np.uint32)
    return rank


class BackgroundWriters:
    """Output files are written by worker threads while the GPU stages run (the native writers release the GIL).  join() waits for all of them
    and re-raises the first failure; main() joins before it returns, whatever happens, so the files are complete whenever the call is over."""
    def __init__(self, workers=None):
        from concurrent.futures import ThreadPoolExecutor
        workers = int(workers or os.environ.get("NGSID_CLI_WRITERS", "8"))
        self.pool = ThreadPoolExecutor(max_workers=workers); self.futures = []
        self.native = fastio.NativeJobs()          # round 5: the record writers run as background jobs of the LIBRARY (native threads, no interpreter lock); the pool is left for the few jobs that are Python

    def submit(self, fn, *a, **kw):
        threads = kw.pop("_threads", None)
        def job():
            # a writer takes at most 8 helper threads (two writers at a time fill a 16-CPU quota; the big writes run while the GPU does the long consensus launches);
            # the writer of sorted.fastq runs beside the clustering call, whose driver thread needs its core back after every launch: 4
            import ctypes as C
            if os.environ.get("NGSID_CLI_WRITER_NICE"):
                import threading
                try: os.setpriority(os.PRIO_PROCESS, threading.get_native_id(), int(os.environ["NGSID_CLI_WRITER_NICE"]))      # dev switch: the helper threads inherit it
                except Exception: pass
            try: runtime.load_library().ngsid_host_thread_cap(C.c_int32(int(threads or os.environ.get("NGSID_CLI_WRITER_THREADS", "8"))))
            except Exception: pass
            return fn(*a, **kw)
        self.futures.append(self.pool.submit(job))

    def join(self):
        futures, self.futures = self.futures, []
        err = None; _t0 = time()
        for f in futures:
            try:
                f.result()
            except BaseException as e:          # keep waiting for the others, report the first
                err = err or e
        _t1 = time()
        try:
            self.native.wait()
        except BaseException as e:
            err = err or e
        if os.environ.get("NGSID_WRITE_TRACE"):
            import sys; sys.stderr.write("[join] python jobs %.3f s, native jobs %.3f s\n" % (_t1 - _t0, time() - _t1))
        if err is not None:
            raise err


_NATIVE = (fastio.write_fastq, fastio.write_tsv, fastio.write_paf)       # + _write_pooled (below): their work is ONE call of the library's record writer


def _single_below(args):
    v = getattr(args, "poa_single_below", None)
    return pipeline.SINGLE_BELOW if v is None else int(v)


def _write(args, fn, *a, **kw):
    bg = getattr(args, "_writers", None)
    if bg is None:
        kw.pop("_threads", None)
        fn(*a, **kw)
    elif (fn in _NATIVE or fn is _write_pooled) and os.environ.get("NGSID_CLI_PY_WRITERS") != "1":
        th = kw.pop("_threads", None)
        if th and fn is fastio.write_fastq: kw["threads"] = int(th)
        fn(*a, jobs=bg.native, **kw)           # prepared here, written by the library's own background threads
    else:
        bg.submit(fn, *a, **kw)


def _write_logfile(logf, err_kept):
    """logfile.txt (get_sorted_fastq_for_cluster.py:184-188): error rates of the kept reads - lowest, highest, median, and the mean as the reference forms it (Python's
    sum over the ascending list, so the digits are the reference's)"""
    er = np.sort(err_kept)
    with open(logf, "w") as lf:
        if len(er):
            lf.write("Lowest read error rate:{0}\n".format(float(er[0])))
            lf.write("Highest read error rate:{0}\n".format(float(er[-1])))
            lf.write("Median read error rate:{0}\n".format(float(er[int(len(er) / 2)])))
            lf.write("Mean read error rate:{0}\n".format(float(sum(er.tolist()) / len(er))))
        lf.write("\n")


class SortedReads:
    """the content of sorted.fastq in memory: reads in score order, original names + '_score' suffixes (CSR bytes, one per read)"""

    def __init__(self, names, rs, suffixes, score, err=None, lens=None):
        self.names, self._rs, self.score, self.err = names, rs, np.asarray(score, dtype=np.float64), err
        self.sfx = suffixes if isinstance(suffixes, tuple) else fastio._csr(suffixes)
        self.n = len(self.score)
        self.lens = np.asarray(lens, dtype=np.int64) if lens is not None else np.diff(rs.off.astype(np.int64))      # read lengths (known before the host copy exists)
        self.dev = None; self.foreign = None          # round 5: the same reads resident on the device in score order + bases outside ACGTN among them (score_and_sort)

    @property
    def rs(self):
        """the host copy of the sorted reads; round 5: gathered by a worker thread while the device clusters (the device got its own copy by a gather in HBM)"""
        if not isinstance(self._rs, ReadSet):
            self._rs = self._rs.result()
        return self._rs

    def suffix(self, i):
        b, o = self.sfx; return b[int(o[i]):int(o[i + 1])].tobytes().decode()

    def name_has_blank(self):
        """per read: does the name contain white space (str.split() would cut it)?  computed once on the compact name buffer"""
        # (round 5: ONE computation, under a lock, started by the worker that gathers the host copy while the device clusters - the five early writers of the pooled files used to
        # find it missing at the same moment and each ran these array passes beside the main thread: a convoy on the interpreter lock, 90 ms in the sampled stacks)
        import threading
        lock = self.__dict__.setdefault("_blank_lock", threading.Lock())
        with lock:
            if getattr(self, "_blank", None) is None:
                nb = self.names
                spaces = np.flatnonzero((nb.buf == 32) | ((nb.buf >= 9) & (nb.buf <= 13)))
                a = nb.off.astype(np.int64); b = a + nb.len.astype(np.int64)
                self._blank = (np.searchsorted(spaces, b, "left") > np.searchsorted(spaces, a, "left")) if len(spaces) else np.zeros(len(nb), dtype=bool)
        return self._blank


def score_and_sort(args, api, T=None):
    """get_sorted_fastq_for_cluster.main: score, filter, stable sort by score, write sorted.fastq + logfile.txt -> SortedReads"""
    T = T if T is not None else {}
    t0 = time()
    out_path = args.outfile
    logf = os.path.join(args.outfolder, "logfile.txt")
    if os.path.isfile(out_path) and getattr(args, "use_old_sorted_file", False):
        open(logf, "w").close()                                           # the reference truncates the log file before it returns (:186)
        logging.warning("Using already existing sorted file in specified directory, in not intended, specify different outfolder or delete the current file.")
        names, rs, _ = fastio.read_fastq(out_path)
        # names carry the '_score' suffix already: split it off so that both entry routes hand over the same structure
        accs = [names.get(i) for i in range(len(names))]
        base = [a.rsplit("_", 1)[0] for a in accs]; sfx = ["_" + a.rsplit("_", 1)[1] for a in accs]
        return SortedReads(fastio.Names.from_list(base), rs, sfx, [float(s[1:]) for s in sfx])
    names, rs, _ = fastio.read_fastq(args.fastq)
    T["read_fastq"] = time() - t0; t0 = time()
    lens = np.diff(rs.off.astype(np.int64))
    # round 5: the reads cross PCIe ONCE, in file order; they are scored there, and the score order is a gather on the device (ngsid_reads_subset) - the host copy in
    # score order (the writers' and the TSVs' source) is gathered by a worker thread while the device clusters.  Backends without the entry point (the test oracle) and
    # empty inputs take the host route of round 4.
    dev_raw = api.upload_reads(rs) if (rs.n and hasattr(api, "reads_subset") and hasattr(api.lib, api.prefix + "reads_subset") and hasattr(api.lib, api.prefix + "reads_upload")) else None
    score, err, keep = api.score_reads(dev_raw if dev_raw is not None else rs, args.k, args.quality_threshold) if rs.n else (np.zeros(0), np.zeros(0), np.zeros(0, np.uint8))
    T["score"] = time() - t0; t0 = time()
    idx = np.nonzero(keep)[0]
    order = idx[fastio.argsort_desc(score[idx])]                          # read_array.sort(key=score, reverse=True) is stable (multi-threaded radix sort: numpy's stable float sort took 0.09 s per million reads)
    sfx = fastio.repr_doubles(score[order], prefix="_")                   # "_" + repr(score), as "{0}".format(score) prints a float
    T["sort"] = time() - t0; t0 = time()
    # sorted.fastq (1.5 GB at C3) is written by a worker thread while the reads are clustered (ngsid_host_write_records: every helper thread writes its records
    # with pwrite() at their own offset).  Dev switch NGSID_CLI_DEFER_SORTED_WRITE=1 starts the write after the clustering call instead (the serial writer of
    # round 3 cost that call 0.1 s of host interference; measured both ways in round 4)
    getattr(args, "_deferred", []).append((fastio.write_fastq, (out_path, order, names, rs), dict(suffixes=sfx))) if hasattr(args, "_deferred") else _write(args, fastio.write_fastq, out_path, order, names, rs, suffixes=sfx, _threads=int(os.environ.get("NGSID_CLI_SORTED_WRITER_THREADS", "4")))
    T["write_sorted_fastq"] = time() - t0; t0 = time()
    logging.debug(f"{len(order)} reads passed quality critera (avg phred Q val over {args.quality_threshold} and length > 2*k) and will be clustered.")
    _write(args, _write_logfile, logf, err[idx])      # (a sort of a million doubles and a sequential Python sum: 0.06 s that nothing waits for - with the other writers)
    nm = names.compact(order)                       # names of the kept reads in their own small buffer (the file buffer can go)
    # the accession ranks of ALL sorted reads (the clustering's third sort key) are computed by a worker thread while the reads are gathered, normalised and
    # uploaded (numpy's sort releases the GIL); cluster() takes them when it clusters every read, which is the usual call
    from concurrent.futures import ThreadPoolExecutor
    sfx_csr = sfx if isinstance(sfx, tuple) else fastio._csr(sfx)
    def _bg(fn, *a):           # a worker beside the launch thread: its helper calls count as background (csrc/host_io.hip: bounded process-wide)
        import ctypes as C
        try: runtime.load_library().ngsid_host_thread_cap(C.c_int32(8))
        except Exception: pass
        return fn(*a)
    pool = ThreadPoolExecutor(max_workers=1)
    fut = pool.submit(_bg, _string_ranks, nm, sfx_csr, np.arange(len(order), dtype=np.int64)); pool.shutdown(wait=False)
    lens_sorted = lens[order]
    if dev_raw is not None:
        t1 = time()
        got = api.reads_subset(dev_raw, order)
        dev_raw.release()
        T["device_gather_sorted"] = time() - t1
        pool2 = ThreadPoolExecutor(max_workers=1)
        sub = pool2.submit(_bg, subset_reads, rs, order); pool2.shutdown(wait=False)
    else:
        got = None
        sub = subset_reads(rs, order)
    T["gather_sorted"] = time() - t0
    sr = SortedReads(nm, sub, sfx_csr, score[order], err[order], lens=lens_sorted)
    if got is not None: sr.dev, sr.foreign = got
    sr.rank_all = fut
    if dev_raw is not None:
        pool3 = ThreadPoolExecutor(max_workers=1); pool3.submit(sr.name_has_blank); pool3.shutdown(wait=False)      # ready before the pooled writers ask for it (they run beside the draft stage)
    return sr


def normalized_reads(sr: SortedReads):
    """the read set the kernels see: upper case, everything outside ACGTN as N (the reference compares raw characters).  The copy
    is dropped when nothing had to change."""
    if fastio.count_foreign_bases(sr.rs.seq) == 0:
        return sr.rs
    seq = sr.rs.seq.copy()
    changed = fastio.normalize_bases(seq)
    if changed:
        logging.warning("%d bases outside A/C/G/T/N (lower case, IUPAC codes, U ...) are clustered as upper case / N; the output files keep the original letters", changed)
        return ReadSet(seq, sr.rs.qual, sr.rs.off)
    return sr.rs


def cluster(sr: SortedReads, work: ReadSet, sel, args, api, work_dev=None, T=None):
    """clusters the reads sel (indices into the sorted set, ascending = processing order).
    -> rep_of [n] (sorted index of the final representative, self for reads outside sel), herr [n], pos [n] (position in the cluster's read list),
       counters, acc_id [n] (dense rank of the accession 'name_score', -1 outside sel; None when all accessions are distinct)"""
    n = sr.n
    rep_of = np.arange(n, dtype=np.int64); herr = np.full(n, np.nan); pos = np.zeros(n, dtype=np.int64)
    if len(sel) == 0:
        return rep_of, herr, pos, np.zeros(4, dtype=np.uint64), None
    lens_all = sr.lens
    too_long = lens_all[sel] > MAX_READ_LEN
    if too_long.any():
        logging.warning("%d reads are longer than %d bases: they are not clustered and stay singletons (use --m / --s to filter by length)", int(too_long.sum()), MAX_READ_LEN)
        sel = sel[~too_long]
    whole = work_dev is not None and len(sel) == n and args.nr_cores <= 1       # every read in one call: the device-resident set as it is
    work = work_dev if whole else subset_reads(work if work is not None else sr.rs, sel)
    prm = cluster_params(k=args.k, w=args.w, min_shared=args.min_shared, min_fraction=args.min_fraction, mapped_threshold=args.mapped_threshold,
                         aligned_threshold=args.aligned_threshold, min_prob_no_hits=args.min_prob_no_hits,
                         symmetric=bool(getattr(args, "symmetric_map_align_thresholds", False)), p_shared=select_p_table(args.k, args.w))
    if np.isnan(select_p_table(args.k, args.w)).all():
        raise KeyError("no rows in the shared-minimizer table for k=%d, w=%d (NGSpeciesID:72-77)" % (args.k, args.w))
    T = T if T is not None else {}
    t1 = time()
    fut = getattr(sr, "rank_all", None)
    rank = fut.result() if (fut is not None and len(sel) == n) else _string_ranks(sr.names, sr.sfx, sel)      # (a subset needs its own dense ranks: equal strings, equal rank)
    T["cluster_wait_for_accession_ranks"] = time() - t1
    lens = lens_all[sel]; score = sr.score[sel]
    counters = np.zeros(4, dtype=np.uint64)
    if args.nr_cores > 1:
        on_dev = work_dev is not None and getattr(api, "has_ctx", False) and hasattr(api.lib, api.prefix + "reads_subset")
        def one_call(a, read_idx, prev_batch, known_err):
            # a batch = a gather on the DEVICE out of the resident read set (the host route - the test oracle - uploads the batch again)
            read_idx = np.asarray(read_idx, dtype=np.int64)
            got = a.reads_subset(work_dev, sel[read_idx]) if on_dev else None
            sub_ = got[0] if got is not None else subset_reads(work, read_idx)
            try:
                return a.cluster_greedy(sub_, prm, acc_rank=rank[read_idx], prev_batch=prev_batch, known_err=known_err)
            finally:
                if got is not None: sub_.release()
        def fn(read_idx, prev_batch, known_err):
            t3 = time(); r = one_call(api, read_idx, prev_batch, known_err)
            counters[:] += r[3]
            T.setdefault("cluster_calls", []).append((len(read_idx), round(time() - t3, 4)))
            return r
        def fn_many(calls):
            # the batches of a round side by side in two contexts of the device (_capi.Api lanes): the last batch of the first round holds the worst reads - hundreds of
            # representatives, restart round after restart round of a few pairs each (0.27 s of latency for 125 k reads at C3) - and the other seven fit beside it
            from ._capi import LANE_MIN_READS, NgsidError
            tw = None
            if on_dev and getattr(api, "lanes", 1) >= 2 and "device" in api.__dict__ and sum(len(c[0]) for c in calls) >= LANE_MIN_READS:
                try: tw = api._twins(1)[0]
                except NgsidError: tw = None
            if tw is None:
                return [fn(*c) for c in calls]
            import threading
            order = [len(calls) - 1] + list(range(len(calls) - 1)); nxt = [0]; lock = threading.Lock(); res = [None] * len(calls); tms = [None] * len(calls)
            def worker(a):
                while True:
                    with lock:
                        if nxt[0] >= len(order): return
                        x = order[nxt[0]]; nxt[0] += 1
                    t3 = time(); res[x] = one_call(a, *calls[x]); tms[x] = round(time() - t3, 4)
            f = tw[1].submit(worker, tw[0])
            try: worker(api)
            finally: e = f.exception()
            if e is not None: raise e
            for x, r in enumerate(res):
                counters[:] += r[3]; T.setdefault("cluster_calls", []).append((len(calls[x][0]), tms[x]))
            return res
        def dump(it, reps, rep_now, herr_now, joins_now, pos_now):
            if os.environ.get("NGSID_CLI_DUMP_TIMES"): args._dump_t = T.setdefault("round_dump_parts", {})
            t2 = time(); write_round_dump(args, sr, sel, it, reps, rep_now, herr_now, pos_now); T["cluster_round_dumps"] = T.get("cluster_round_dumps", 0.0) + time() - t2
        trk = {}
        rep_l, herr_l, joins = parallelize.tree_cluster(fn, lens, score, args.nr_cores, getattr(args, "batch_type", "total_nt"), on_round=dump if getattr(args, "outfolder", None) else None, track=trk, cluster_fn_many=fn_many)
        pos_l = trk["pos"]                     # (= parallelize.list_positions(len(sel), joins), kept up to date round by round)
    else:
        t1 = time()
        rep_l, herr_l, st, cnt = api.cluster_greedy(work, prm, acc_rank=rank)
        T["cluster_library_call"] = time() - t1; t1 = time()
        counters[:] = cnt
        rep_l = rep_l.astype(np.int64)
        is_rep = rep_l == np.arange(len(sel))
        herr_l = np.where(is_rep, herr_l, np.nan)
        # single pass: the list of a cluster is its representative followed by the joining reads in processing order, so a read's position is the number of
        # earlier reads of its cluster: one counting pass (ngsid_host_list_positions; the numpy sort-based form took 0.1 s per million reads)
        pos_l = fastio.list_positions(rep_l)
        T["cluster_list_positions"] = time() - t1
    rep_of[sel] = sel[rep_l]; herr[sel] = herr_l; pos[sel] = pos_l
    logging.debug("Passed mapping criteria:{0}".format(int(counters[0])))
    logging.debug("Passed alignment criteria in this process:{0}".format(int(counters[1])))
    logging.debug("Total calls to alignment module in this process:{0}".format(int(counters[2])))
    acc_id = None
    if len(rank) and int(rank.max()) + 1 < len(rank):                  # duplicate accessions exist (same name AND same score)
        acc_id = np.full(n, -1, dtype=np.int64); acc_id[sel] = rank
    return rep_of, herr, pos, counters, acc_id


def write_round_dump(args, sr, sel, it, reps, r, herr, pos_):
    """<outfolder>/<it>/pre_clusters.csv and cluster_origins.csv of a round of the --t > 1 schedule (parallelize.py:85-104,193): the clusters by size, largest first, ties in the
    order of the merged dictionaries (= reps); members in list order, names without the score suffix; cluster ids are positions in the sorted read file (sel maps the clustered reads
    to them).  The member order is a scatter (a read's place = start of its cluster + its position in the cluster's list: no sort over the reads), the million-line file goes to the
    background writers like the final ones (0.25 s of the --t 8 CLI at C3 when both were done the plain way)."""
    n = len(sel); _t = [time()]; _T = getattr(args, "_dump_t", None)
    def _mark(k):
        if _T is not None: _T[k] = _T.get(k, 0.0) + time() - _t[0]
        _t[0] = time()
    sizes = np.bincount(r, minlength=n)
    by_size = reps[np.argsort(-sizes[reps], kind="stable")]
    start = np.zeros(n, dtype=np.int64); start[by_size] = np.concatenate(([0], np.cumsum(sizes[by_size])[:-1]))
    _mark("sizes")
    members = np.empty(n, dtype=np.int64); members[start[r] + pos_] = np.arange(n, dtype=np.int64)
    _mark("scatter")
    folder = os.path.join(args.outfolder, str(it))
    os.makedirs(folder, exist_ok=True)
    _write(args, fastio.write_tsv, os.path.join(folder, "pre_clusters.csv"), sel[members], sr.names, fastio.int_prefixes(sel[r[members]]))
    _mark("prefixes+queue")
    with open(os.path.join(folder, "cluster_origins.csv"), "w") as f:
        for c in by_size.tolist():
            g = int(sel[c]); seq, qual = sr.rs.get(g); e = herr[c]
            f.write("{0}\t{1}\t{2}\t{3}\t{4}\t{5}\n".format(g, sr.names.get(g) + sr.suffix(g), seq, qual, float(sr.score[g]), "" if np.isnan(e) else float(e)))
    _mark("origins")
    logging.debug("Nr clusters larger than 1: %d", int((sizes[reps] > 1).sum()))
    logging.debug("Nr clusters (all):  %d", len(reps))


def cluster_table(sr, sel, rep_of, pos, single_pass=False):
    """clusters of the clustered reads: (reps in OUTPUT order [(size, score) descending, read index ascending], sizes, member index lists as CSR in the
    reference's list order, and in file order (score descending, stable)).  Representatives are read indices, so sizes come from a bincount and the list order
    from a scatter (pos = position in the cluster's list, a permutation of 0 .. size-1 per cluster): no sort over the million reads."""
    n = sr.n
    r = rep_of[sel]
    sizes_all = np.bincount(r, minlength=n)
    reps = np.flatnonzero(sizes_all); sizes = sizes_all[reps]
    idmap = np.empty(n, dtype=np.int64); idmap[reps] = np.arange(len(reps)); inv = idmap[r]
    out_order = np.lexsort((reps, -sr.score[reps], -sizes))               # sorted(clusters.items(), key=(len, score), reverse=True): ties keep dict order = read order
    out_rank = np.empty(len(reps), dtype=np.int64); out_rank[out_order] = np.arange(len(reps))
    cl = out_rank[inv]                                                     # output id of every clustered read
    goff = np.zeros(len(reps) + 1, dtype=np.int64); goff[1:] = np.cumsum(sizes[out_order])
    list_order = np.full(len(sel), -1, dtype=np.int64)
    tgt = goff[cl] + pos[sel]
    ok = len(sel) == 0 or (int(tgt.min()) >= 0 and int(tgt.max()) < len(sel))
    if ok: list_order[tgt] = sel
    if not ok or (list_order < 0).any():                                  # (not a permutation per cluster: never seen; the sort is the definition)
        list_order = sel[np.lexsort((pos[sel], cl))]                      # cluster by cluster, the reference's list order (spoa input order)
    if single_pass:
        # one clustering pass over score-sorted reads: a cluster's list is its representative (the smallest index) followed by the members in processing
        # order = ascending index = descending score with ties in list order: the file order IS the list order
        file_order = list_order
    else:
        file_order = sel[np.lexsort((pos[sel], -sr.score[sel], cl))]      # sorted(all_read_acc, key=score, reverse=True) is stable w.r.t. the list order
    return reps[out_order], sizes[out_order], goff, list_order, file_order, np.repeat(np.arange(len(reps), dtype=np.int64), sizes[out_order])


def write_cluster_files(args, sr, reps, sizes, herr, file_order, cl_sorted):
    _write(args, fastio.write_tsv, os.path.join(args.outfolder, "final_clusters.tsv"), file_order, sr.names, fastio.int_prefixes(cl_sorted))
    with open(os.path.join(args.outfolder, "final_cluster_origins.tsv"), "w") as f:
        for out_id, r in enumerate(reps.tolist()):
            seq, qual = sr.rs.get(r)
            e = herr[r]
            f.write("{0}\t{1}\t{2}\t{3}\t{4}\t{5}\n".format(out_id, sr.names.get(r), seq, qual, float(sr.score[r]), "" if np.isnan(e) else float(e)))
    return int((sizes > 1).sum())


def consensus_and_polish(args, sr, work, reps, sizes, goff, list_order, abundance_cutoff, api, acc_id=None, T=None):
    """form_draft_consensus + detect_reverse_complements + polish_sequences on the resident read set; writes the reference's files"""
    nsel = int((sizes >= abundance_cutoff).sum())                          # clusters are in (size, score) order already: the selected ones are a prefix
    singles = int((sizes == 1).sum()) if abundance_cutoff > 1 else 0
    disc = sizes[(sizes < abundance_cutoff) & (sizes > 1)]
    logging.debug(f"{singles} singletons were discarded")
    logging.debug(f"{len(disc)} clusters were discarded due to not passing the abundance_cutoff: a total of {int(disc.sum())} reads were discarded. "
                  f"Highest abundance among them: {int(disc.max()) if len(disc) else 0} reads.")
    for folder in glob.glob(os.path.join(args.outfolder, "racon_cl_id_*")):
        shutil.rmtree(folder)
    for file in glob.glob(os.path.join(args.outfolder, "consensus_reference_*")):
        os.remove(file)
    if nsel == 0:
        return []
    T = T if T is not None else {}
    t0 = time()
    mx = args.max_seqs_for_consensus
    groups = []                                                            # per selected cluster: its reads in list order, truncated (consensus.py:260)
    for c in range(nsel):
        a, b = int(goff[c]), int(goff[c + 1])
        if mx >= 0:
            b = min(b, a + mx)
        groups.append(list_order[a:b])
    # The pooled read file of a centre that absorbs no other cluster is that cluster's read list (consensus.py:208-215), known NOW: its writer starts before the draft
    # consensus instead of after the reverse-complement merge, so the 1.5 GB of reads_to_consensus_*.fastq at C3 have the draft AND the polishing stage to reach the disk.
    # A centre that does absorb others (rare) has its file rewritten after the merge, when these writers are done.  (Not with duplicate accessions: the pooled file of
    # the reference de-duplicates by header, handled in _merge_and_polish.)
    # (round 5: everything the main thread computes with arrays comes BEFORE the writers start - with six threads in array code every operation of this thread waited its turn for the
    # interpreter lock: 90 ms for the three lines below in the sampled stacks of a stalled run)
    gmax = int(sr.lens[np.concatenate(groups)].max())
    cgroups = groups                                                       # the reads the consensus stages see
    args._overlong = False
    if gmax > MAX_CONSENSUS_LEN:
        # one over-long read (a concatemer) in an abundant cluster must not end the run.  Such reads stay in the TSVs and in the pooled read files; the draft and
        # the polisher leave them out, with a warning.  Only a cluster left without any usable read is an error.
        cgroups = [g[sr.lens[g] <= MAX_CONSENSUS_LEN] for g in groups]
        ndrop = sum(len(g) - len(cg) for g, cg in zip(groups, cgroups))
        empty = [int(reps[c]) for c in range(nsel) if len(cgroups[c]) == 0]
        if empty:
            raise ValueError("clusters %s hold only reads longer than %d bases (up to %d): this build's POA engine forms consensus of reads up to %d bases (clustering itself handles %d); "
                             "filter by length (--m / --s) or raise --abundance_ratio so that these clusters are not polished" % (empty[:10], MAX_CONSENSUS_LEN, gmax, MAX_CONSENSUS_LEN, MAX_READ_LEN))
        logging.warning("%d read(s) longer than %d bases (up to %d) are left out of the consensus and polishing of their clusters (they stay in final_clusters.tsv and in the pooled read files)", ndrop, MAX_CONSENSUS_LEN, gmax)
        args._overlong = True
        gmax = int(max(sr.lens[cg].max() for cg in cgroups))
    sub_off = np.concatenate(([0], np.cumsum([len(g) for g in cgroups]))).astype(np.uint64)
    read_order = np.concatenate(cgroups); read_order32 = read_order.astype(np.uint32)
    args._merge_passes = 0                                                 # (a reused args namespace must not look like a later pass)
    args._pooled_early = {}
    if acc_id is None and getattr(args, "_writers", None) is not None and os.environ.get("NGSID_CLI_EARLY_POOLED", "1") == "1":
        for c in range(nsel):
            path = os.path.join(args.outfolder, "reads_to_consensus_{0}.fastq".format(int(reps[c])))
            _write(args, _write_pooled, path, groups[c], sr)
            args._pooled_early[int(reps[c])] = (c,)
    long_reads = gmax > 1000
    node_cap = 22 if long_reads else 0
    drafts = api.poa_consensus(work, sub_off, poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=getattr(args, "poa_tile_depth", pipeline.TILE_DEPTH), band=getattr(args, "poa_band", 0), node_cap=node_cap, trim=pipeline.DRAFT_TRIM, single_below=_single_below(args)),
                               read_order=read_order32)
    T["draft_consensus"] = time() - t0
    centers = [[int(sizes[c]), int(reps[c]), drafts[c], [c]] for c in range(nsel)]
    barcodes = None
    if getattr(args, "primer_file", "") or getattr(args, "remove_universal_tails", False):          # NGSpeciesID:134-142
        from . import barcode_trimmer
        barcodes = barcode_trimmer.get_universal_tails() if args.remove_universal_tails else barcode_trimmer.read_barcodes(args.primer_file)
        logging.debug("Detecting and removing universal tails" if args.remove_universal_tails else "Detecting and removing primers")
        # The reference's order (round 5; NGSpeciesID:134-152): trim the drafts, merge, polish the TRIMMED sequences - minimap2 keeps the primer ends of the reads out of
        # racon's alignment, here the polisher's overlap-span clipping does (aln_mode 3: of a read only the columns between its first and last run of 15 equal columns
        # count) together with trim 3 (a window keeps the ends of its backbone where no layer reaches them, as racon's NGS windows do) - then look for primers again and,
        # only if one was removed, run detect_reverse_complements + polish_sequences a second time.  (Rounds 2 - 4 polished the UNTRIMMED drafts and trimmed afterwards,
        # because the whole-read aligner carried the primers back into a trimmed backbone.)
        barcode_trimmer.remove_barcodes(centers, barcodes, args)
        logging.debug("{0} centers formed".format(len(centers)))
        merged = _merge_and_polish(args, sr, work, centers, groups, node_cap, api, acc_id, T, clip=True)
        if barcode_trimmer.remove_barcodes(merged, barcodes, args):                                   # NGSpeciesID:147-152: a primer was still there after polishing
            merged = _merge_and_polish(args, sr, work, [list(m) for m in merged], groups, node_cap, api, acc_id, T, clip=True)
        return merged
    logging.debug("{0} centers formed".format(len(centers)))
    return _merge_and_polish(args, sr, work, centers, groups, node_cap, api, acc_id, T)


def _merge_and_polish(args, sr, work, centers, groups, node_cap, api, acc_id, T, polish_backbones=None, used_out=None, clip=False):
    """detect_reverse_complements + polish_sequences (consensus.py:148-183,186-246): centers = [n_reads, c_id, sequence, cluster indices]"""
    t0 = time()
    for folder in glob.glob(os.path.join(args.outfolder, "racon_cl_id_*")):
        shutil.rmtree(folder)
    for file in glob.glob(os.path.join(args.outfolder, "consensus_reference_*")):
        os.remove(file)
    T["rc_merge_cleanup"] = T.get("rc_merge_cleanup", 0.0) + time() - t0; t1 = time()
    merged = pipeline.detect_reverse_complements(api, centers, args.rc_identity_threshold)
    T["rc_merge_detect"] = T.get("rc_merge_detect", 0.0) + time() - t1
    logging.debug(f"{len(merged)} consensus formed.")
    pooled = []
    seen_clusters = set(); polish_lists = []          # the polisher takes a read under one centre only (pipeline.pooled_read_lists): first centre wins
    early = getattr(args, "_pooled_early", {})        # pooled files whose writers were started before the draft: {c_id: cluster tuple the file holds}
    first_pass = not getattr(args, "_merge_passes", 0); args._merge_passes = getattr(args, "_merge_passes", 0) + 1
    if not first_pass and getattr(args, "_writers", None) is not None:
        args._writers.join()          # a later pass may queue a writer for a pooled file a job of the pass before is still writing
    if early and any(early.get(int(c_id)) != tuple(cs) for _, c_id, _, cs in merged):
        # some centre absorbed other clusters: wait for the early writers, then rewrite the merged files below.  FIRST pass: the files of the absorbed centres go - the
        # reference writes a pooled file per MERGED centre only (consensus.py:208-215), the early writers had started one per selected cluster.  Later passes (merge
        # after trimming, NGSpeciesID:147-152): the reference's second polish_sequences removes the racon folders and the consensus_reference_* files
        # (consensus.py:195-200) but leaves the pooled file an absorbed centre got in the first pass - so does this build, in both writer modes.
        args._writers.join()
        keep = {int(m[1]) for m in merged}
        for cid in list(early):
            if cid not in keep:
                if first_pass:
                    try: os.remove(os.path.join(args.outfolder, "reads_to_consensus_{0}.fastq".format(cid)))
                    except OSError: pass
                del early[cid]
    for nr, c_id, center, cs in merged:
        with open(os.path.join(args.outfolder, "consensus_reference_{0}.fasta".format(c_id)), "w") as f:
            f.write(">{0}\n{1}\n".format("consensus_cl_id_{0}_total_supporting_reads_{1}".format(c_id, nr), center))
        parts = []
        for c in cs:                                       # every source file is de-duplicated by header (a dict, consensus.py:210-212): first position, last content
            g = groups[c]
            if acc_id is not None and len(g) > 1:
                key = acc_id[g]
                u, first = np.unique(key, return_index=True)
                if len(u) < len(g):
                    _, last_rev = np.unique(key[::-1], return_index=True)
                    last = len(g) - 1 - last_rev                   # u is sorted the same way in both calls
                    o = np.argsort(first, kind="stable")
                    g = g[last[o]]
            parts.append(g)
        ids = np.concatenate(parts)
        pooled.append(ids)
        fresh = [p for c, p in zip(cs, parts) if c not in seen_clusters]
        if len(fresh) < len(parts):
            logging.warning("centre %d: %d cluster(s) were merged into an earlier centre as well; their reads polish that one only", c_id, len(parts) - len(fresh))
        seen_clusters.update(cs)
        pl = np.concatenate(fresh) if fresh else np.zeros(0, dtype=ids.dtype)
        if getattr(args, "_overlong", False): pl = pl[sr.lens[pl] <= MAX_CONSENSUS_LEN]          # over-long reads stay in the pooled file, the polisher leaves them out
        polish_lists.append(pl)
        if early.get(int(c_id)) != tuple(cs):
            _write(args, _write_pooled, os.path.join(args.outfolder, "reads_to_consensus_{0}.fastq".format(c_id)), ids, sr)     # while the polisher runs
            if early: early[int(c_id)] = tuple(cs)
    T["rc_merge_write_pooled_reads"] = T.get("rc_merge_write_pooled_reads", 0.0) + time() - t0; t0 = time()
    if getattr(args, "racon", False) and args.racon_iter >= 0:
        p_off = np.concatenate(([0], np.cumsum([len(x) for x in polish_lists]))).astype(np.uint64)
        bb = ReadSet.from_strings([(polish_backbones or {}).get(m[1], m[2]) for m in merged])
        prm = polish_params(iters=args.racon_iter, k=args.k, w=args.w, tile_depth=(getattr(args, "poa_tile_depth", 0) if getattr(args, "poa_tile_depth", 0) > 0 else pipeline.TILE_DEPTH), band=getattr(args, "poa_band", 0), node_cap=node_cap, trim=3 if clip else 2, aln_mode=3 if clip else 2,
                            stop_when_stable=0 if getattr(args, "polish_all_iterations", False) else 1, single_below=_single_below(args))      # clip: backbones are primer-trimmed (include/ngsid.h: aln_mode 3, trim 3)
        ro = np.concatenate(polish_lists).astype(np.uint32)
        want_paf = not getattr(args, "skip_paf", False)
        its_aln = None
        if args.racon_iter >= 1:                     # every iteration's sequence: run_racon leaves racon_polished_it_{i}.fasta behind (consensus.py:112-120) - and minimap2's PAF
            if want_paf: its, its_used, its_aln = api.polish_trace(bb, work, p_off, prm, read_order=ro, aln=True)
            else: its, its_used = api.polish_trace(bb, work, p_off, prm, read_order=ro)
            polished, used = its[-1], its_used[-1]
        else:
            its, its_used = [], []
            polished, used = api.polish(bb, work, p_off, prm, read_order=ro)
        for x, (nr, c_id, center, cs) in enumerate(merged):
            logging.debug("running racon on spoa reference {0} using {1} reads for polishing.".format(c_id, len(pooled[x])))
            folder = os.path.join(args.outfolder, "racon_cl_id_{0}".format(c_id))
            os.makedirs(folder, exist_ok=True)
            open(os.path.join(folder, "stdout.txt"), "w").close()
            name = "consensus_cl_id_{0}_total_supporting_reads_{1}".format(c_id, nr)
            if its:
                consensus_mod.write_racon_iteration_files(folder, name, [it[x] for it in its], [int(u[x]) for u in its_used])
                if its_aln is not None:                  # read_alignments_it_{i}.paf (consensus.py:112-121): the alignments iteration i polished with, against the sequence it started from
                    a, b = int(p_off[x]), int(p_off[x + 1]); start = (polish_backbones or {}).get(c_id, center)
                    for i in range(len(its)):
                        _write(args, fastio.write_paf, os.path.join(folder, "read_alignments_it_{0}.paf".format(i)), polish_lists[x], sr.names, sr.rs.off, its_aln[i][a:b], name,
                               len(start if i == 0 else its[i - 1][x]), suffixes=sr.sfx)
            else:
                shutil.copyfile(os.path.join(args.outfolder, "consensus_reference_{0}.fasta".format(c_id)), os.path.join(folder, "consensus.fasta"))
            merged[x][2] = polished[x]
            if used_out is not None: used_out[c_id] = int(used[x])
        T["polish"] = T.get("polish", 0.0) + time() - t0
    return merged


def _write_pooled(path, ids, sr, jobs=None):
    # names in the pooled file = first token of the sorted-file accession "name_score" (consensus.py:213): the suffix belongs to the name, so the
    # cut is applied to name + suffix; names with blanks lose their suffix with everything behind the blank
    blank = sr.name_has_blank()[ids] if len(ids) else np.zeros(0, dtype=bool)
    if not blank.any():
        fastio.write_fastq(path, ids, sr.names, sr.rs, suffixes=sr.sfx, first_token=True, sfx_by_read=True, jobs=jobs)
    else:
        sfx = [("" if bl else sr.suffix(int(i))) for i, bl in zip(ids.tolist(), blank.tolist())]
        fastio.write_fastq(path, ids, sr.names, sr.rs, suffixes=sfx, first_token=True, jobs=jobs)


def main(args, api=None):
    api = api or runtime.get_api()
    args._writers = BackgroundWriters() if not os.environ.get("NGSID_CLI_SYNC_WRITES") else None
    import sys, gc
    _gc = gc.isenabled() and os.environ.get("NGSID_CLI_GC", "0") != "1"
    if _gc: gc.disable()            # no cyclic collection during the run: a full collection of the interpreter's heap (torch is loaded) holds the lock for tens of ms in whichever thread triggers it
    _swi = sys.getswitchinterval(); sys.setswitchinterval(5e-4)      # a thread that wants the interpreter lock asks for it after 0.5 ms instead of 5 ms: the launch thread comes back from every library call beside up to eight writers
    _fd = None
    if os.environ.get("NGSID_CLI_STACKS"):          # dev aid (round 5): the stacks of ALL threads every 5 ms, written by faulthandler's own watchdog thread (it needs no interpreter lock)
        import faulthandler
        _fd = open(os.environ["NGSID_CLI_STACKS"], "a"); faulthandler.dump_traceback_later(0.005, repeat=True, file=_fd)
    if os.environ.get("NGSID_WRITE_TRACE"):
        import time as _t; sys.stderr.write("[cli] start %.3f\n" % (_t.monotonic() % 1000))
    try:
        res = _main(args, api)
        if os.environ.get("NGSID_WRITE_TRACE"):
            import time as _t; sys.stderr.write("[cli] stages done %.3f %s\n" % (_t.monotonic() % 1000, {k: round(v, 3) for k, v in res["timings"].items()}))
    finally:
        sys.setswitchinterval(_swi)
        if _gc: gc.enable()
        fastio.release_mapped()            # the input file's mapping (read_fastq leaves it to the end of the run: fastio.py)
        if _fd is not None:
            import faulthandler
            faulthandler.cancel_dump_traceback_later(); _fd.close()
        if args._writers is not None:
            t0 = time()
            try:
                args._writers.join()
            finally:
                args._writers.pool.shutdown(wait=True); args._writers = None
            wait = time() - t0
    res["timings"]["wait_for_writers"] = wait if os.environ.get("NGSID_CLI_SYNC_WRITES") is None else 0.0
    logging.debug("stage seconds: %s" % {k: (round(v, 3) if isinstance(v, float) else v) for k, v in res["timings"].items()})
    return res


def _main(args, api):
    T = {}
    t0 = time()
    args.outfile = os.path.join(args.outfolder, "sorted.fastq")
    if os.environ.get("NGSID_CLI_DEFER_SORTED_WRITE", "0") == "1": args._deferred = []        # dev switch (A/B of the write schedule, round 4): with the parallel record writer the write is short enough to run beside the clustering call
    sr = score_and_sort(args, api, T)
    if sr.dev is not None and sr.foreign == 0:
        work, work_dev = None, sr.dev               # the device holds the sorted reads already (gathered in HBM) and nothing needs normalising: `work` = sr.rs, taken when a caller needs the host copy
        T["normalize"] = 0.0; T["upload"] = 0.0; t0 = time()
    else:
        if sr.dev is not None: sr.dev.release(); sr.dev = None
        work = normalized_reads(sr)
        T["normalize"] = time() - t0 - sum(T.values()); t0 = time()
        work_dev = api.upload_reads(work)               # ONE copy to HBM for the clustering, the draft consensus and the polishing calls
        T["upload"] = time() - t0; t0 = time()
    sel = np.arange(sr.n, dtype=np.int64)
    if args.target_length > 0 and args.target_deviation > 0:
        lens = sr.lens
        sel = sel[(lens >= args.target_length - args.target_deviation) & (lens <= args.target_length + args.target_deviation)]
        logging.debug("Number of reads with read length in interval [{0},{1}]: {2}".format(args.target_length - args.target_deviation, args.target_length + args.target_deviation, len(sel)))
    if args.top_reads:
        sel = sel[:args.sample_size]
    elif 0 < args.sample_size < len(sel):
        sel = sel[np.asarray(sorted(random.sample(range(len(sel)), args.sample_size)), dtype=np.int64)]
    abundance_cutoff = int(args.abundance_ratio * len(sel))
    logging.info(f"Starting Clustering: {len(sel)} reads")
    try:
        rep_of, herr, pos, counters, acc_id = cluster(sr, work, sel, args, api, work_dev, T)
    finally:                                        # sorted.fastq is written whatever the clustering call did (the reference has it on disk before it clusters)
        deferred, args._deferred = getattr(args, "_deferred", []), []
        for fn_, a_, kw_ in deferred: _write(args, fn_, *a_, **kw_)
    T["cluster"] = time() - t0; t0 = time()
    logging.debug(f"Time elapsed clustering: {T['cluster']}")
    if getattr(args, "strand_aware", False) and len(sel) == sr.n:
        # extension (strand.py): reverse-complement clusters are joined here, before the cluster files are written and before any consensus work
        from . import strand
        prm = cluster_params(k=args.k, w=args.w, min_shared=args.min_shared, min_fraction=args.min_fraction, mapped_threshold=args.mapped_threshold,
                             aligned_threshold=args.aligned_threshold, min_prob_no_hits=args.min_prob_no_hits,
                             symmetric=bool(getattr(args, "symmetric_map_align_thresholds", False)), p_shared=select_p_table(args.k, args.w))
        if work is None: work = sr.rs
        rep_of, flip, pos, sinfo = strand.strand_merge(api, work, rep_of, sr.score, prm, min_size=max(2, abundance_cutoff // 2), pos=pos)
        logging.debug("strand-aware merge: %d of %d candidate clusters joined their reverse complement" % (sinfo["merged"], sinfo["candidates"]))
        if flip.any():
            work = strand.orient_reads(work, flip)
            if work_dev is not None and work_dev.mem != work.mem: work_dev.release()
            work_dev = api.upload_reads(work)
        T["strand_merge"] = time() - t0; t0 = time()
    elif getattr(args, "strand_aware", False):
        logging.warning("--strand_aware needs all reads clustered (no --m / --s / --sample_size selection): ignored")
    reps, sizes, goff, list_order, file_order, cl_sorted = cluster_table(sr, sel, rep_of, pos, single_pass=(args.nr_cores <= 1 and not getattr(args, "strand_aware", False)))
    nontrivial = write_cluster_files(args, sr, reps, sizes, herr, file_order, cl_sorted)
    T["write_clusters"] = time() - t0; t0 = time()
    logging.debug(f"Nr clusters larger than 1: {nontrivial}")
    logging.debug(f"Nr clusters (all): {len(reps)}")
    logging.info(f"Finished Clustering: {nontrivial} clusters formed")
    merged = []
    if args.consensus:
        logging.info("Starting Consensus creation and polishing")
        logging.debug(f"Forming draft consensus with abundance_cutoff >= {abundance_cutoff} ({args.abundance_ratio * 100}% of {len(sel)} reads)")
        merged = consensus_and_polish(args, sr, work_dev, reps, sizes, goff, list_order, abundance_cutoff, api, acc_id, T)
        logging.info(f"Finished Consensus creation: {len(merged)} created")
    if work_dev is not work:
        work_dev.release()
    return dict(n_sorted=sr.n, n_clustered=len(sel), clusters=len(reps), centers=merged, timings=T, counters=counters)
