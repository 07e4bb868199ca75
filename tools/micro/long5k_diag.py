"""dev tool: 5 kb ONT reads - where do the polished sequences differ from the amplicons?"""
import sys, os, time, ctypes as C, difflib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from util_seq import edit_distance
api = runtime.get_api(0); dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000; band = int(sys.argv[2]) if len(sys.argv) > 2 else 0; L = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
sp, rd = bench.gen_sorted_reads(api, n, 5, L, 17.0, seed=3, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
for it in (0, 1, 3):
    res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=it, tile_depth=8, band=band,
                                p_shared=select_p_table(13, 20), polish_stop_when_stable=False, do_polish=it > 0)
    buf = C.create_string_buffer(1 << 16); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(len(buf)))
    redo = [l for l in buf.value.decode().splitlines() if l.startswith("poa_band_redo")]
    truths = [s.tobytes().decode() for s in sp]
    print("iters", it, redo)
    for c in res["centers"]:
        seq = c[3] if it > 0 else c[2]
        ti = int(np.argmin([edit_distance(seq[:600], t[:600]) for t in truths])); t = truths[ti]
        e = edit_distance(seq, t)
        sm = difflib.SequenceMatcher(None, seq, t, autojunk=False)
        ops = [(tag, i1, i2, j1, j2) for tag, i1, i2, j1, j2 in sm.get_opcodes() if tag != "equal"]
        print("  centre of %d reads: len %d (amplicon %d) ed %d   diffs %s" % (c[0], len(seq), len(t), e, ops[:8]))
