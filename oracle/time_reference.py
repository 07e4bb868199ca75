#!/usr/bin/env python3
"""Time the REFERENCE's own Python clustering (build container only; TEST / MEASUREMENT INFRASTRUCTURE).

/root/reference is imported (never copied); `parasail` is the shim backed by the oracle's C aligner, so the aligner cost is that of a
scalar C implementation (parasail's SIMD would be faster).  spoa / racon / minimap2 are absent, so only the clustering stage of the
reference can be timed here.  Output: reads/s of get_sorted_fastq_for_cluster and of reads_to_clusters (--t 1) on synthetic reads of the
bench profile.     python oracle/time_reference.py [n_reads]
"""
import os, sys, time, tempfile, shutil
if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"; os.execv(sys.executable, [sys.executable] + sys.argv)
import logging
logging.disable(logging.CRITICAL)
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as mg                      # sets up the shim and the reference import paths (its main() is not run on import)
from ngspeciesid_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
sp = synth.make_species(5, 750, 0.15, seed=1)
rd = synth.make_reads(sp, n, mu=17.0, seed=2)
tmp = tempfile.mkdtemp()
fq = os.path.join(tmp, "reads.fastq")
seq, qual, off = rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy()
with open(fq, "w") as f:
    for i in range(n):
        f.write("@r%d_sp%d\n%s\n+\n%s\n" % (i, int(rd["species"][i]), seq[off[i]:off[i + 1]].tobytes().decode(), qual[off[i]:off[i + 1]].tobytes().decode()))
t0 = time.perf_counter(); ra = mg.sorted_read_array(fq, 13, tmp); t1 = time.perf_counter()
args = mg.ref_args(k=13, w=20, nr_cores=1); pt = mg.p_table(13, 20)
clusters = {i: [acc] for i, b, acc, s, q, sc in ra}; reps = {i: (i, b, acc, s, q, sc) for i, b, acc, s, q, sc in ra}
ncalls0 = len(mg.parasail.CALLS)
t2 = time.perf_counter(); res = mg.cluster.reads_to_clusters(clusters, reps, ra, pt, {}, 1, args); t3 = time.perf_counter()
cl = list(res.values())[0][0]
sizes = sorted((len(v) for v in cl.values()), reverse=True)[:6]
print("reference Python, 1 core, %d synthetic 750 bp reads (5 species, mu=17): score+sort %.0f reads/s; reads_to_clusters %.0f reads/s (%d aligner calls through the C shim); largest clusters %s"
      % (n, n / (t1 - t0), len(ra) / (t3 - t2), len(mg.parasail.CALLS) - ncalls0, sizes))
shutil.rmtree(tmp)
