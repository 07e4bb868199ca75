#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON (build container only).

TEST INFRASTRUCTURE.  /root/reference is imported here (never copied); `parasail` is satisfied by
oracle/ref_shim/parasail.py (the oracle's aligner), `edlib` by a stub.  The fixtures hold inputs and
the reference's outputs only (data), and are what pins the oracle (tests/test_oracle_golden.py).
Run:  make -C oracle && python oracle/make_golden.py          (re-execs itself with PYTHONHASHSEED=0)
"""
import os, sys
if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

import argparse, itertools, tempfile, shutil, logging
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "ref_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
import parasail  # noqa: E402  (the shim)
from modules import cluster, parallelize, help_functions, get_sorted_fastq_for_cluster, p_minimizers_shared, consensus  # noqa: E402
from ngspeciesid_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def ref_args(**kw):
    a = argparse.Namespace(k=13, w=20, min_shared=5, mapped_threshold=0.7, aligned_threshold=0.4,
                           symmetric_map_align_thresholds=False, min_fraction=0.8, min_prob_no_hits=0.1,
                           print_output=10 ** 9, nr_cores=1, batch_type="total_nt", quality_threshold=7.0,
                           outfolder=None, fastq=None, use_old_sorted_file=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def p_table(k, w):
    d = {}
    for kk, ww, p, e1, e2 in p_minimizers_shared.read_empirical_p():       # NGSpeciesID:72-77
        if int(kk) == k and abs(int(ww) - w) <= 2:
            d[(float(e1), float(e2))] = float(p)
            d[(float(e2), float(e1))] = float(p)
    return d


def p_table_dense(d):
    t = np.full(225, np.nan)
    for (e1, e2), p in d.items():
        i, j = int(round(e1 * 100)), int(round(e2 * 100))
        t[(i - 1) * 15 + (j - 1)] = p
    return t


def csr(strings):
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    return np.frombuffer("".join(strings).encode(), dtype=np.uint8).copy(), off


def enc_code(kmer, k):
    m = {"A": 1, "C": 2, "G": 3, "N": 4, "T": 5}
    c = 0
    for i in range(k):
        c = (c << 3) | (m[kmer[i]] if i < len(kmer) else 0)
    return c


# ------------------------------------------------------------------------------------------------ A: minimizers / HPC
def golden_minimizers(reads, tag):
    phred = {chr(i): min(10 ** (-(ord(chr(i)) - 33) / 10.0), 0.79433) for i in range(128)}
    out = {}
    seqs = [s for _, s, _ in reads]; quals = [q for _, _, q in reads]
    out["seq"], out["off"] = csr(seqs)
    out["qual"], _ = csr(quals)
    hpcs, errs = [], []
    for s, q in zip(seqs, quals):
        h = "".join(ch for ch, _ in itertools.groupby(s))                   # cluster.py:265
        lens = [len(list(g)) for _, g in itertools.groupby(s)]               # cluster.py:279-286
        qc, st = [], 0
        for hl in lens:
            qc.append(min(q[st:st + hl], key=lambda x: phred[x])); st += hl
        qc = "".join(qc)
        errs.append(sum([qc.count(c) * phred[c] for c in set(qc)]) / float(len(qc)))   # :290-291
        hpcs.append(h)
    out["hpc"], out["hpc_off"] = csr(hpcs)
    out["hpc_err"] = np.array(errs)
    for (k, w) in ((13, 20), (15, 50), (10, 100), (21, 21)):
        codes, pos, moff = [], [], [0]
        for h in hpcs:
            if len(h) >= k:
                for kmer, p in cluster.get_kmer_minimizers(h, k, w):        # cluster.py:16-39
                    codes.append(enc_code(kmer, k)); pos.append(p)
            moff.append(len(codes))
        out["codes_%d_%d" % (k, w)] = np.array(codes, dtype=np.uint64)
        out["pos_%d_%d" % (k, w)] = np.array(pos, dtype=np.uint32)
        out["moff_%d_%d" % (k, w)] = np.array(moff, dtype=np.uint64)
    np.savez_compressed(os.path.join(GOLD, "minimizers_%s.npz" % tag), **out)
    print("minimizers", tag, len(reads), "reads")


# ------------------------------------------------------------------------------------------------ B/C: clustering
class SerialPool:
    def __init__(self, processes=None): pass
    def map_async(self, fn, data):
        res = [fn(d) for d in data]
        class R:
            def get(self, timeout=None): return res
        return R()
    def close(self): pass
    def join(self): pass
    def terminate(self): pass


def run_reference_clustering(read_array, k, w, t):
    """read_array: [(i, 0, acc_with_score, seq, qual, score)] like NGSpeciesID:58"""
    args = ref_args(k=k, w=w, nr_cores=t)
    pt = p_table(k, w)
    trace = {}
    calls = []
    orig_gbc, orig_gba = cluster.get_best_cluster, cluster.get_best_cluster_block_align

    def gbc(read_cl_id, *a, **kw):
        r = orig_gbc(read_cl_id, *a, **kw)
        trace[read_cl_id] = r
        return r

    def gba(read_cl_id, *a, **kw):
        r = orig_gba(read_cl_id, *a, **kw)
        calls.append((read_cl_id, r[0]))
        return r
    cluster.get_best_cluster, cluster.get_best_cluster_block_align = gbc, gba
    try:
        if t == 1:
            clusters, representatives = {}, {}
            for i, b_i, acc, seq, qual, score in read_array:               # single_clustering NGSpeciesID:20-33
                clusters[i] = [acc]; representatives[i] = (i, b_i, acc, seq, qual, score)
            res = cluster.reads_to_clusters(clusters, representatives, read_array, pt, {}, 1, args)
            clusters, representatives, _, _ = list(res.values())[0]
        else:
            tmp = tempfile.mkdtemp(); args.outfolder = tmp
            parallelize.Pool = SerialPool
            clusters, representatives = parallelize.parallel_clustering(list(read_array), pt, args)
            shutil.rmtree(tmp)
    finally:
        cluster.get_best_cluster, cluster.get_best_cluster_block_align = orig_gbc, orig_gba
    acc2i = {acc: i for i, _, acc, _, _, _ in read_array}
    n = len(read_array)
    rep_of = np.full(n, -1, dtype=np.int32)
    order, ooff, keys = [], [0], []
    for cid, accs in clusters.items():
        keys.append(cid)
        for a in accs:
            rep_of[acc2i[a]] = cid; order.append(acc2i[a])
        ooff.append(len(order))
    err = np.full(n, np.nan)
    for cid, tup in representatives.items():
        if len(tup) == 8: err[cid] = tup[6]
    out = dict(rep_of=rep_of, cl_keys=np.array(keys, dtype=np.int32), cl_order=np.array(order, dtype=np.int32),
               cl_off=np.array(ooff, dtype=np.int64), rep_err=err)
    if t == 1:
        bm = np.full(n, -2, dtype=np.int32); ns = np.zeros(n, dtype=np.int32); ra = np.zeros(n)
        for rid, (b, s, r) in trace.items():
            bm[rid], ns[rid], ra[rid] = b, s, r
        out.update(tr_best=bm, tr_nshared=ns, tr_ratio=ra,
                   aln_calls=np.array(calls, dtype=np.int32).reshape(-1, 2),
                   counters=np.array([sum(1 for v in trace.values() if v[0] >= 0), sum(1 for c in calls if c[1] >= 0), len(calls)], dtype=np.int64))
    return out


def golden_cluster(read_array, k, w, tag, ts=(1, 2, 4, 8)):
    out = {}
    out["acc"] = np.array([a for _, _, a, _, _, _ in read_array])
    out["seq"], out["off"] = csr([s for _, _, _, s, _, _ in read_array])
    out["qual"], _ = csr([q for _, _, _, _, q, _ in read_array])
    out["score"] = np.array([sc for *_, sc in read_array])
    out["k"], out["w"] = np.int32(k), np.int32(w)
    out["p_table"] = p_table_dense(p_table(k, w))
    for t in ts:
        r = run_reference_clustering(read_array, k, w, t)
        for kk, v in r.items():
            out["t%d_%s" % (t, kk)] = v
        print("cluster", tag, "t=%d" % t, "clusters", len(r["cl_keys"]), "sizes", sorted(np.diff(r["cl_off"]).tolist(), reverse=True)[:6],
              "counters", r.get("counters"))
    np.savez_compressed(os.path.join(GOLD, "cluster_%s.npz" % tag), **out)


def sorted_read_array(fastq, k, tmp):
    args = ref_args(k=k, nr_cores=1, fastq=fastq, outfolder=tmp)
    args.outfile = os.path.join(tmp, "sorted.fastq")
    path = get_sorted_fastq_for_cluster.main(args)                          # NGSpeciesID:49
    return [(i, 0, acc, seq, qual, float(acc.split("_")[-1])) for i, (acc, (seq, qual)) in enumerate(help_functions.readfq(open(path, "r")))]


def golden_scores(fastq, k, tag):
    reads = [(acc, seq, qual) for acc, (seq, qual) in help_functions.readfq(open(fastq, "r"))]
    D_no_min = get_sorted_fastq_for_cluster.D_no_min
    import math
    score, err, keep = [], [], []
    for acc, seq, qual in reads:                                            # fastq_single_core :124-155
        h = "".join(ch for ch, _ in itertools.groupby(seq))
        if len(seq) < 2 * k or len(h) < k:
            score.append(0.0); err.append(0.0); keep.append(0); continue
        ee = get_sorted_fastq_for_cluster.expected_number_of_erroneous_kmers(qual, k)
        p_no = 1.0 - ee / float((len(seq) - k + 1))
        score.append(p_no * (len(seq) - k + 1))
        er = sum([qual.count(c) * D_no_min[c] for c in set(qual)]) / float(len(qual))
        err.append(er); keep.append(0 if 10 * -math.log(er, 10) <= 7.0 else 1)
    out = dict(score=np.array(score), err=np.array(err), keep=np.array(keep, dtype=np.uint8), k=np.int32(k))
    out["seq"], out["off"] = csr([s for _, s, _ in reads]); out["qual"], _ = csr([q for _, _, q in reads])
    np.savez_compressed(os.path.join(GOLD, "scores_%s.npz" % tag), **out)
    print("scores", tag, len(reads), "kept", int(np.sum(keep)))


# ------------------------------------------------------------------------------------------------ D: batch_list
def golden_batches():
    out = {}
    cases = {"eq8_c4": ([100] * 8, 4), "eq8_c8": ([100] * 8, 8), "ragged_c3": ([50, 700, 20, 300, 300, 10, 900, 5, 5], 3),
             "fill_c2": ([10, 10, 10, 10], 2), "one_c4": ([500], 4), "n13_c8": (list(range(100, 113)), 8)}
    for name, (lens, cores) in cases.items():
        lst = [(i, 0, "a%d_1.0" % i, "A" * L, "I" * L, 1.0) for i, L in enumerate(lens)]
        b = [len(x) for x in parallelize.batch_list(lst, cores, "total_nt")]           # parallelize.py:54-67
        out[name + "_lens"] = np.array(lens, dtype=np.int64); out[name + "_cores"] = np.int32(cores)
        out[name + "_sizes"] = np.array(b, dtype=np.int64)
        print("batch_list", name, b)
    np.savez_compressed(os.path.join(GOLD, "batch_list.npz"), **out)


# ------------------------------------------------------------------------------------------------ F/G: window identity, rc identity
def golden_align(pairs, tag):
    out = {}
    s1 = [a for a, _ in pairs]; s2 = [b for _, b in pairs]
    out["q"], out["q_off"] = csr(s1); out["t"], out["t_off"] = csr(s2)
    ratios, tratios, opens, mids, scores, ncols, nmatch, ident = [], [], [], [], [], [], [], []
    rnd = np.random.default_rng(5)
    for a, b in pairs:
        op = int(rnd.integers(2, 6)); mid = int(rnd.integers(7, 13)); k = 13
        _, _, (a1, a2, r, tr) = cluster.parasail_block_alignment(a, b, k, mid, opening_penalty=op)   # cluster.py:130-169
        ratios.append(r); tratios.append(tr); opens.append(op); mids.append(mid)
        ncols.append(len(a1)); nmatch.append(sum(1 for x, y in zip(a1, a2) if x == y))
        res = parasail.sg_trace_scan_16(a, b, op, 1, (2, -2)); scores.append(res.score)
        ident.append(consensus.highest_aln_identity(a, b))                                            # consensus.py:129-145
    out.update(ratio=np.array(ratios), target_ratio=np.array(tratios), open=np.array(opens, dtype=np.int32),
               match_id=np.array(mids, dtype=np.int32), score=np.array(scores, dtype=np.int32),
               n_cols=np.array(ncols, dtype=np.int32), n_match=np.array(nmatch, dtype=np.int32), identity=np.array(ident))
    np.savez_compressed(os.path.join(GOLD, "align_%s.npz" % tag), **out)
    print("align", tag, len(pairs), "pairs; ratio range", min(ratios), max(ratios))


def main():
    logging.basicConfig(level=logging.WARNING)
    tmp = tempfile.mkdtemp()
    fq = "/root/reference/test/sample_h1.fastq"
    raw = [(acc, seq, qual) for acc, (seq, qual) in help_functions.readfq(open(fq, "r"))]
    # edge cases for the minimizer encoder
    edge = [("e_k", "ACGTACGTACGTA", "I" * 13), ("e_k1", "ACGTACGTACGTAC", "5" * 14), ("e_w1", "ACGTACGTACGTACGTACG", "+" * 19),
            ("e_w", "ACGTACGTACGTACGTACGT", "&" * 20), ("e_same", "AC" * 40, "?" * 80), ("e_n", "ACGTNACGTTGCANNACGTACGTTGACTGACTGATCGATGCATCGTAGCTAGCTAGCATCGA", "9" * 62),
            ("e_hp", "AAAAAACCCCCGGGGGTTTTTACGTACGTACGTTTTTGGGGACGTAGCTAGCTAGGGGGGGGGCTAGCATCGACTGACTGACTAGC", "".join(chr(33 + (i * 7) % 40) for i in range(86))),
            ("e_short", "ACGTAC", "IIIIII"), ("e_lowq", "ACGTTGCATGCATGCATCGATCGATCGATGCATGCATCGATCGATCG", "!\"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJKLMNO")]
    golden_minimizers(raw + edge, "sample_h1")
    sp = synth.make_species(5, 750, 0.15, seed=1)
    rd = synth.make_reads(sp, 200, mu=17.0, seed=3)
    s = rd["seq"].numpy(); q = rd["qual"].numpy(); off = rd["off"].numpy()
    golden_minimizers([("r%d" % i, s[off[i]:off[i + 1]].tobytes().decode(), q[off[i]:off[i + 1]].tobytes().decode()) for i in range(200)], "synth200")

    golden_scores(fq, 13, "sample_h1")
    ra = sorted_read_array(fq, 13, tmp)
    golden_cluster(ra, 13, 20, "sample_h1")

    # synthetic sets: 5 species, 15 % divergence (wide decision margins) and a 10 % "hard" set; one CCS-like set at k15/w50
    for tag, n, L, div, mu, k, w, nsp in (("synth2k_d15", 2000, 750, 0.15, 17.0, 13, 20, 5), ("synth600_d10_q14", 600, 750, 0.10, 14.0, 13, 20, 5),
                                          ("synth300_ccs", 300, 1200, 0.15, 30.0, 15, 50, 4)):
        spx = synth.make_species(nsp, L, div, seed=11)
        rdx = synth.make_reads(spx, n, mu=mu, seed=5)
        fqx = os.path.join(tmp, tag + ".fastq"); synth.reads_to_fastq(rdx, fqx)
        rax = sorted_read_array(fqx, k, tmp)
        golden_cluster(rax, k, w, tag, ts=(1, 2, 8) if n >= 1000 else (1, 4))
    golden_batches()

    pairs = []
    rnd = np.random.default_rng(9)
    seqs = [t[3] for t in ra]
    for _ in range(40):
        i, j = rnd.integers(0, len(seqs), 2); pairs.append((seqs[i], seqs[j]))
    pairs += [("ACGT", "ACGT"), ("A", "C"), ("ACGTACGTAC", "TTTTTTTTTTTTTT"), ("ACGTNNACGTACGTAGCTAGC", "ACGTACGTACGTAGCTAGCNN"), ("acgtacgtagctagctagcatcg", "ACGTACGTAGCTAGCTAGCATCG"),
              ("GATTACAGATTACAGATTACA", "GATTACAGATTACA"), ("GATTACAGATTACA", "CCCCGATTACAGATTACATTTT")]
    golden_align(pairs, "sample_h1")
    shutil.rmtree(tmp)




def golden_cli():
    """final_clusters.tsv / final_cluster_origins.tsv / sorted.fastq of the reference CLI (--t 1, clustering only) on test/sample_h1.fastq."""
    import importlib.machinery, importlib.util, hashlib
    loader = importlib.machinery.SourceFileLoader("ngs_ref_cli", "/root/reference/NGSpeciesID")
    spec = importlib.util.spec_from_loader("ngs_ref_cli", loader); mod = importlib.util.module_from_spec(spec)
    sys.modules["parasail"] = parasail
    loader.exec_module(mod)
    tmp = tempfile.mkdtemp()
    args = ref_args(k=13, w=20, nr_cores=1, fastq="/root/reference/test/sample_h1.fastq", outfolder=tmp, consensus=False, target_length=0, target_deviation=0,
                    top_reads=False, sample_size=0, abundance_ratio=0.1, primer_file="", remove_universal_tails=False, ont=True, isoseq=False)
    mod.main(args)
    for name in ("final_clusters.tsv", "final_cluster_origins.tsv"):
        shutil.copyfile(os.path.join(tmp, name), os.path.join(GOLD, "sample_h1_t1_" + name))
    with open(os.path.join(GOLD, "sample_h1_t1_sorted.fastq.md5"), "w") as f:
        f.write(hashlib.md5(open(os.path.join(tmp, "sorted.fastq"), "rb").read()).hexdigest() + "\n")
    shutil.rmtree(tmp)
    print("cli goldens written")


if __name__ == "__main__" and "--cli-only" in sys.argv:
    golden_cli()

if __name__ == "__main__" and "--cli-only" not in sys.argv:
    main()
    golden_cli()
