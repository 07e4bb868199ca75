"""dev tool: accuracy of the whole path by POA tile depth on noisy reads - per (mu, depth): number of (seed, cluster) pairs whose polished consensus / draft
differs from its amplicon, over several seeds.  argv: n_reads seeds depths(comma) mus(comma)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import bench
from util_seq import edit_distance
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
api = runtime.get_api(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# a depth is D, "D:node_cap" (graph capacity in 1/16 of the first sequence: 0:160 = ONE graph per unit with room for ten times the first sequence = the restated spoa order) or
# "aT" = the adaptive rule of round 6 (units with fewer than T sequences run as one graph, larger ones at the default depth)
depths = [x for x in (sys.argv[3] if len(sys.argv) > 3 else "6,8").split(",")]
def dkw(d):
    if d.startswith("a"): return dict(tile_depth=pipeline.TILE_DEPTH, single_below=int(d[1:]))
    if ":" in d: return dict(tile_depth=int(d.split(":")[0]), node_cap=int(d.split(":")[1]), single_below=0)
    return dict(tile_depth=int(d), single_below=0)
mus = [float(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "14,13").split(",")]
extra = dict(kv.split("=") for kv in sys.argv[5:])
dev = torch.device("cuda", 0)
for mu in mus:
    tot = {d: [0, 0, 0.0] for d in depths}
    for seed in range(20, 20 + seeds):
        sp, rd = bench.gen_sorted_reads(api, n, 5, 750, mu, seed=seed, device=dev)
        rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
        truths = [s.tobytes().decode() for s in sp]
        for d in depths:
            t = time.perf_counter()
            res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3, band=0, **dkw(d),
                                        p_shared=select_p_table(13, 20), polish_stop_when_stable=False, **{k_: int(v) for k_, v in extra.items()})
            dt = time.perf_counter() - t
            bad_p = [min(edit_distance(c[3], t_) for t_ in truths) for c in res["centers"] if c[3] not in truths]
            bad_d = [min(edit_distance(c[2], t_) for t_ in truths) for c in res["centers"] if c[2] not in truths]
            tot[d][0] += len(bad_p) + abs(5 - len(res["centers"])); tot[d][1] += len(bad_d); tot[d][2] += dt
            if bad_p or bad_d: print("  mu %.0f seed %d depth %s: polished off by %s, drafts off by %s" % (mu, seed, d, bad_p, bad_d), flush=True)
    for d in depths:
        print("mu %.0f depth %6s: %d of %d polished wrong, %d drafts wrong, %.3f s per run" % (mu, d, tot[d][0], 5 * seeds, tot[d][1], tot[d][2] / seeds), flush=True)
