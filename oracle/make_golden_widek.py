#!/usr/bin/env python3
"""Golden vectors for k > 21 (the reference's table has rows for k = 10..30), produced by EXECUTING THE REFERENCE'S OWN PYTHON (build container
only; TEST INFRASTRUCTURE, see make_golden.py):
  tests/golden/minimizers_widek.npz   cluster.get_kmer_minimizers at (25,30) and (30,35) on sample_h1 + 200 synthetic reads: positions, and the
                                      k-mer strings as their dense rank among all minimizers of the set (what ngsid_hpc_minimizers returns for k > 21)
  tests/golden/cluster_synth2k_d15_k25.npz / _k30.npz   reads_to_clusters / parallel_clustering at k25/w30 and k30/w35 (--t 1 and --t 2)
Run:  make -C oracle && python oracle/make_golden_widek.py
"""
import os, sys
if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)
import itertools, tempfile, shutil
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # imports the reference's modules behind the parasail shim


def wide_minimizers():
    raw = [(acc, seq, qual) for acc, (seq, qual) in mg.help_functions.readfq(open("/root/reference/test/sample_h1.fastq", "r"))]
    sp = mg.synth.make_species(5, 750, 0.15, seed=1)
    rd = mg.synth.make_reads(sp, 200, mu=17.0, seed=3)
    s = rd["seq"].numpy(); q = rd["qual"].numpy(); off = rd["off"].numpy()
    syn = [("r%d" % i, s[off[i]:off[i + 1]].tobytes().decode(), q[off[i]:off[i + 1]].tobytes().decode()) for i in range(200)]
    edge = [("e_k", "ACGTACGTACGTAACGTTGCATGCATG", "I" * 27), ("e_short", "ACGTACGTACGTACGTACGTAC", "5" * 22), ("e_n", "ACGTNACGTTGCANNACGTACGTTGACTGACTGATCGATGCATCGTAGCTAGCTAGCATCGA", "9" * 62),
            ("e_w1", "ACGTACGTACGTACGTACGTACGTACGTA", "+" * 29), ("e_w", "ACGTACGTACGTACGTACGTACGTACGTAC", "&" * 30)]
    reads = raw + syn + edge
    out = {}
    seqs = [x[1] for x in reads]
    out["seq"], out["off"] = mg.csr(seqs); out["qual"], _ = mg.csr([x[2] for x in reads])
    hpcs = ["".join(ch for ch, _ in itertools.groupby(x)) for x in seqs]
    for (k, w) in ((25, 30), (30, 35), (22, 22), (32, 40)):
        kmers, pos, moff = [], [], [0]
        for h in hpcs:
            if len(h) >= k:
                for kmer, p in mg.cluster.get_kmer_minimizers(h, k, w):        # cluster.py:16-39
                    kmers.append(kmer); pos.append(p)
            moff.append(len(kmers))
        uniq = {km: i for i, km in enumerate(sorted(set(kmers)))}             # Python str order = the reference's min() order
        out["rank_%d_%d" % (k, w)] = np.array([uniq[km] for km in kmers], dtype=np.uint64)
        out["pos_%d_%d" % (k, w)] = np.array(pos, dtype=np.uint32)
        out["moff_%d_%d" % (k, w)] = np.array(moff, dtype=np.uint64)
        print("wide minimizers", k, w, len(kmers), "distinct", len(uniq))
    np.savez_compressed(os.path.join(mg.GOLD, "minimizers_widek.npz"), **out)


def wide_clusters():
    tmp = tempfile.mkdtemp()
    spx = mg.synth.make_species(5, 750, 0.15, seed=11)
    rdx = mg.synth.make_reads(spx, 1200, mu=17.0, seed=5)
    fqx = os.path.join(tmp, "w.fastq"); mg.synth.reads_to_fastq(rdx, fqx)
    for k, w in ((25, 30), (30, 35)):
        rax = mg.sorted_read_array(fqx, k, tmp)
        mg.golden_cluster(rax, k, w, "synth1200_k%d" % k, ts=(1, 2))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    wide_minimizers()
    wide_clusters()
