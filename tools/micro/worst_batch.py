"""GPU box: the batch of the WORST reads of the bench workload on its own (what the last batch of the first `--t 8` round is): time, representatives, and - with
NGSID_CLUSTER_TRACE=1 - the block trace of the clustering driver.   python tools/micro/worst_batch.py [reads] [batches]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
api = runtime.get_api(0); dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c3"]
sp, rd = bench.gen_sorted_reads(api, n, cfg["species"], cfg["length"], cfg["mu"], seed=7, device=dev, k=cfg["k"])
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
N = rs.n; a = N - N // nb
sub, _ = api.reads_subset(rs, np.arange(a, N, dtype=np.uint64))
prm = cluster_params(k=cfg["k"], w=cfg["w"], p_shared=select_p_table(cfg["k"], cfg["w"]))
rank = np.asarray(rd["orig"], dtype=np.uint32)[a:N]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rep_of, herr, st, cnt = api.cluster_greedy(sub, prm, acc_rank=rank)
    dt = time.perf_counter() - t0
    print("worst batch: %d reads, %.3f s, %d representatives, counters %s" % (sub.n, dt, int((rep_of == np.arange(sub.n)).sum()), cnt.tolist()))
