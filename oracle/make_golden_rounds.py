"""Build container only: the per-round dumps of the reference's parallel_clustering (parallelize.py:85-104,193: <outfolder>/<it>/pre_clusters.csv, cluster_origins.csv) on
test/sample_h1.fastq with --t 4, produced by IMPORTING AND RUNNING the reference (its Pool replaced by a serial one, parasail by oracle/ref_shim) -> tests/golden/sample_h1_t4_round_dumps.json.
    PYTHONHASHSEED=0 python oracle/make_golden_rounds.py"""
import json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as G                                  # the harness: shim path, ref_args, SerialPool, p_table, sorted_read_array
from modules import parallelize                           # /root/reference (make_golden put it on sys.path)

def main():
    tmp = tempfile.mkdtemp()
    read_array = G.sorted_read_array("/root/reference/test/sample_h1.fastq", 13, tmp)
    out = tempfile.mkdtemp()
    args = G.ref_args(k=13, w=20, nr_cores=4, outfolder=out)
    parallelize.Pool = G.SerialPool
    parallelize.parallel_clustering(list(read_array), G.p_table(13, 20), args)
    files = {}
    for root, _, fs in os.walk(out):
        for f in fs:
            files[os.path.relpath(os.path.join(root, f), out)] = open(os.path.join(root, f)).read()
    json.dump(files, open(os.path.join(ROOT, "tests", "golden", "sample_h1_t4_round_dumps.json"), "w"), indent=0, sort_keys=True)
    print(sorted(files), {k: len(v) for k, v in files.items()})

if __name__ == "__main__":
    main()
