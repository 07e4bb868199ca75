"""dev tool: consensus accuracy on the noisy profile (mu=14, ~10 % read error): edit distance to the amplicon with and without end trimming"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from util_seq import edit_distance
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000; mu = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp, rd = bench.gen_sorted_reads(api, n, 5, 750, mu, seed=7, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3, tile_depth=8, band=128, p_shared=select_p_table(13, 20))
truths = [s.tobytes().decode() for s in sp]
for c in res["centers"]:
    best = None
    for ti, t in enumerate(truths):
        for a in range(0, 13, 2):
            for b in range(0, 13, 2):
                e = edit_distance(c[3][a:len(c[3]) - b if b else None], t)
                if best is None or e < best[0] or (e == best[0] and a + b < best[1] + best[2]): best = (e, a, b, ti)
    e0 = edit_distance(c[3], truths[best[3]])
    print("cluster of %d reads: consensus length %d (amplicon %d): edit distance %d untrimmed, %d after trimming %d / %d end bases; draft distance %d" %
          (c[0], len(c[3]), len(truths[best[3]]), e0, best[0], best[1], best[2], edit_distance(c[2], truths[best[3]])))
