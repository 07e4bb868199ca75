"""dev tool: consensus + polish of the bench workload on ONE context vs on TWO contexts driven by two host threads (clusters split between
them; each context leaves room for the other's small launches).  argv: n_reads per_cu"""
import sys, os, time, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, cluster_params, POA_LOCAL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
per_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
api = runtime.get_api(0)
sp, rd = bench.gen_sorted_reads(api, n, 5, 750, 17.0, 7, dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"]); torch.cuda.synchronize()
from ngspeciesid_amd import ptable
rep_of, herr, status, counters = api.cluster_greedy(rs, cluster_params(k=13, w=20, p_shared=ptable.select_p_table(13, 20)))
reps, order, grp_off, counts = pipeline.clusters_from_rep(rep_of)
sel = pipeline.select_centers(reps, counts, rd["score"], int(0.02 * rs.n))
groups = [order[int(grp_off[ci]):int(grp_off[ci + 1])] for ci in sel]
print("groups", [len(g) for g in groups])
lens = np.diff(np.asarray(rd["off"].cpu().numpy(), dtype=np.int64))
band = 64 if lens.max() <= 1024 else 128

def run(a, gs):
    off = np.concatenate(([0], np.cumsum([len(g) for g in gs]))).astype(np.uint64)
    ro = np.concatenate(gs).astype(np.uint32)
    t0 = time.perf_counter()
    drafts = a.poa_consensus(rs, off, poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=8, band=band, trim=pipeline.DRAFT_TRIM), read_order=ro)
    t1 = time.perf_counter()
    pol, used = a.polish(ReadSet.from_strings(drafts), rs, off, polish_params(iters=3, k=13, w=20, tile_depth=8, band=band, trim=2, stop_when_stable=0), read_order=ro)
    return drafts, pol, t1 - t0, time.perf_counter() - t1

for it in range(2):
    t = time.perf_counter(); d1, p1, td, tp = run(api, groups); t1 = time.perf_counter() - t
    print("one context: %.3f s (draft %.3f polish %.3f)" % (t1, td, tp))
api.lib.ngsid_ctx_option(api.ctx, b"poa_tiles_per_cu", C.c_int64(per_cu))
api2 = runtime.new_api(0, {"poa_tiles_per_cu": per_cu})
# split: groups by size descending, alternating
idx = sorted(range(len(groups)), key=lambda i: -len(groups[i]))
sets = [idx[0::2], idx[1::2]]
for it in range(3):
    out = [None, None]
    def work(k):
        out[k] = run([api, api2][k], [groups[i] for i in sets[k]])
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
    t2 = time.perf_counter() - t
    d2 = [None] * len(groups); p2 = [None] * len(groups)
    for k in range(2):
        for j, i in enumerate(sets[k]):
            d2[i] = out[k][0][j]; p2[i] = out[k][1][j]
    print("two contexts (per_cu %d): %.3f s  [ctx0 draft %.3f polish %.3f | ctx1 draft %.3f polish %.3f]  same drafts %s same polished %s" % (per_cu, t2, out[0][2], out[0][3], out[1][2], out[1][3], d2 == d1, p2 == p1))
api2.close()
