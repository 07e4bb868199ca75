"""dev tool: the hot path on the long-read configurations (C5-like 2 kb CCS k15/w50, 5 kb ONT) - wall time per stage and consensus check.
    python tools/micro/time_longreads.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from util_seq import edit_distance
api = runtime.get_api(0); dev = torch.device("cuda", 0)
for (n, nsp, L, mu, k, w, ab) in ((200000, 20, 2000, 30.0, 15, 50, 0.002), (100000, 5, 5000, 17.0, 13, 20, 0.02)):
    sp, rd = bench.gen_sorted_reads(api, n, nsp, L, mu, seed=3, device=dev)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    for rep in range(2):
        T = {}; t0 = time.perf_counter()
        res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=k, w=w, abundance_ratio=ab, racon_iter=3, tile_depth=8, band=128,
                                    p_shared=select_p_table(k, w), timings=T, polish_stop_when_stable=False)
        dt = time.perf_counter() - t0
    truths = [s.tobytes().decode() for s in sp]
    eds = [min(min(edit_distance(c[3][a:len(c[3]) - b if b else None], t) for a in range(3) for b in range(3)) for t in truths) for c in res["centers"]]
    import hashlib
    print("n=%d L=%d k=%d w=%d mu=%g: %.2fs -> %.0f reads/s, stages %s, centers %d, max edit distance to the amplicons %d, f_aln %.3f, reads md5 %s, result md5 %s" % (n, L, k, w, mu, dt, n / dt, {a: round(b, 3) for a, b in T.items()}, len(res["centers"]), max(eds), float(res["counters"][2]) / n, hashlib.md5(rd["seq"].cpu().numpy().tobytes()).hexdigest()[:8], hashlib.md5("".join(c[3] for c in res["centers"]).encode()).hexdigest()[:8]), flush=True)
