"""dev tool: one pass of the hot path on the bench workload with NGSID_HOST_TIMERS=1 - the library prints the host-side time of every stage
(job lists, uploads, kernels, downloads per hierarchy level) to stderr."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import numpy as np, torch, bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp, rd = bench.gen_sorted_reads(api, 1000000, 5, 750, 17.0, seed=7, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
for rep in range(2):
    if rep == 1: os.environ["NGSID_HOST_TIMERS"] = "1"
    res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3, tile_depth=8, band=0, p_shared=select_p_table(13, 20), polish_stop_when_stable=False)
