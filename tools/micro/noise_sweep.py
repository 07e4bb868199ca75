"""dev tool: accuracy / speed of the whole path over the noise level and the POA tile depth
    python tools/micro/noise_sweep.py n_reads mu1,mu2,... depth1,depth2,..."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from util_seq import edit_distance
n = int(sys.argv[1]); mus = [float(x) for x in sys.argv[2].split(",")]; depths = [int(x) for x in sys.argv[3].split(",")]
api = runtime.get_api(0); dev = torch.device("cuda", 0)
for mu in mus:
    sp, rd = bench.gen_sorted_reads(api, n, 5, 750, mu, seed=7, device=dev)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    truths = [s.tobytes().decode() for s in sp]
    q = rd["qual"].cpu().numpy().astype(np.float64) - 33.0
    print("mu %.0f: %d of %d reads pass the filters, mean per-base error %.3f" % (mu, rs.n, n, float(np.mean(10.0 ** (-q[::97] / 10.0)))), flush=True)
    for D in depths:
        for rep in range(2):
            T = {}; t0 = time.perf_counter()
            res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3, tile_depth=D, band=0,
                                        p_shared=select_p_table(13, 20), polish_stop_when_stable=False, timings=T)
            dt = time.perf_counter() - t0
        eds = [min(edit_distance(c[3], t) for t in truths) for c in res["centers"]]; deds = [min(edit_distance(c[2], t) for t in truths) for c in res["centers"]]
        nrep = int((res["rep_of"] == np.arange(rs.n)).sum())
        print("   depth %2d: %.3f s (%s)  centres %d sizes %s  polished ed %s  draft ed %s  representatives %d  f_aln %.2f" %
              (D, dt, {a: round(b, 2) for a, b in T.items() if a in ("cluster", "consensus", "polish")}, len(res["centers"]), [c[0] for c in res["centers"]], eds, deds, nrep, float(res["counters"][2]) / rs.n), flush=True)
