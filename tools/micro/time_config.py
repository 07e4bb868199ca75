"""dev tool: the hot path on one synthetic configuration - stage times, kernel times and the consensus check.
    python tools/micro/time_config.py n_reads n_species length mu k w abundance_ratio [seed]
e.g. C4 per GPU: 1250000 50 750 17 13 20 0.005"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from util_seq import edit_distance
n, nsp, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); mu = float(sys.argv[4]); k, w = int(sys.argv[5]), int(sys.argv[6]); ab = float(sys.argv[7])
seed = int(sys.argv[8]) if len(sys.argv) > 8 else 3
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp, rd = bench.gen_sorted_reads(api, n, nsp, L, mu, seed=seed, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
for rep in range(2):
    if rep == 1: api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    T = {}; t0 = time.perf_counter()
    res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=k, w=w, abundance_ratio=ab, racon_iter=3, tile_depth=8, band=0,
                                p_shared=select_p_table(k, w), timings=T, polish_stop_when_stable=False)
    dt = time.perf_counter() - t0
buf = C.create_string_buffer(1 << 16); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(len(buf)))
kern = {l.split()[0]: round(float(l.split()[2]), 1) for l in buf.value.decode().splitlines()}
truths = [s.tobytes().decode() for s in sp]
eds = [min(min(edit_distance(c[3][a:len(c[3]) - b if b else None], t) for a in range(3) for b in range(3)) for t in truths) for c in res["centers"]]
n = rs.n; nrep = int((res["rep_of"] == np.arange(n)).sum())
print("n=%d species=%d L=%d: %.2fs -> %.0f reads/s; stages %s; kernels(ms) %s; representatives %d, centers %d, max edit distance %d, f_aln %.3f" %
      (n, nsp, L, dt, n / dt, {a: round(b, 3) for a, b in T.items()}, kern, nrep, len(res["centers"]), max(eds) if eds else -1, float(res["counters"][2]) / n))
