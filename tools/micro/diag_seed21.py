"""dev tool: the one depth-6 miss of the accuracy sweep (mu 14, seed 21, 200 k reads): where is the edit, which iteration makes it, which depths make it"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, difflib
import bench
from util_seq import edit_distance
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet, polish_params
from ngspeciesid_amd.ptable import select_p_table
api = runtime.get_api(0); dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 21
sp, rd = bench.gen_sorted_reads(api, 200000, 5, 750, 14.0, seed=seed, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
truths = [s.tobytes().decode() for s in sp]
res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3, tile_depth=6, band=0, p_shared=select_p_table(13, 20), polish_stop_when_stable=False)
rep = res["rep_of"]
for c in res["centers"]:
    if c[3] in truths: continue
    t = min(truths, key=lambda x: edit_distance(x, c[3]))
    sm = difflib.SequenceMatcher(None, t, c[3], autojunk=False)
    for op in sm.get_opcodes():
        if op[0] != "equal": print("cluster", c[1], "size", c[0], op, "truth ctx", t[max(0, op[1] - 12):op[2] + 12], "got ctx", c[3][max(0, op[3] - 12):op[4] + 12])
    ids = np.nonzero(np.isin(rep, c[4]))[0].astype(np.uint32)
    bb = ReadSet.from_strings([t])
    for depth in (6, 7, 8, 5, 12):
        for it in (1, 2, 3):
            pol, used = api.polish(bb, rs, [0, len(ids)], polish_params(iters=it, k=13, w=20, tile_depth=depth, band=0, trim=2, stop_when_stable=0), read_order=ids)
            print("  from the truth: depth %d iters %d -> ed %d (used %d)" % (depth, it, edit_distance(pol[0], t), int(used[0])))
    for nsub in (len(ids) // 2, len(ids) // 4, 5000):
        pol, used = api.polish(bb, rs, [0, nsub], polish_params(iters=1, k=13, w=20, tile_depth=6, band=0, trim=2, stop_when_stable=0), read_order=ids[:nsub])
        print("  first %d reads, depth 6: ed %d" % (nsub, edit_distance(pol[0], t)))
    for band in (128,):
        pol, used = api.polish(bb, rs, [0, len(ids)], polish_params(iters=1, k=13, w=20, tile_depth=6, band=band, trim=2, stop_when_stable=0), read_order=ids)
        print("  band %d depth 6: ed %d" % (band, edit_distance(pol[0], t)))
