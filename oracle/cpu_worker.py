"""bench.py `cpu_baseline` leg (test infrastructure): one worker process = one host core running the CPU oracle's whole path (cluster + draft
consensus + polish) on its batch of the sample, like one of the reference's `--t N` worker processes.  Usage: cpu_worker.py sample.npz a b out.json"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    path, a, b, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from oracle_lib import load_oracle
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.hostutil import subset_reads
    z = np.load(path)
    rs = subset_reads(ReadSet(z["seq"], z["qual"], z["off"]), np.arange(a, b))
    orc = load_oracle()
    kw = json.loads(str(z["kw"]))
    kw["p_shared"] = z["p_shared"]
    t = time.perf_counter()
    pipeline.run_hot_path(orc, rs, z["score"][a:b], acc_rank=z["acc_rank"][a:b], **kw)
    json.dump(dict(reads=b - a, seconds=time.perf_counter() - t), open(out, "w"))


if __name__ == "__main__":
    main()
