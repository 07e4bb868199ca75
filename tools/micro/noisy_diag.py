"""dev tool: diagnostics on the failing mu=14 cluster (centre found by noisy_capture): per-iteration ends, parameter variations"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL
from ngspeciesid_amd.ptable import select_p_table
from ngspeciesid_amd.hostutil import subset_reads
from util_seq import edit_distance
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp, rd = bench.gen_sorted_reads(api, 1000000, 5, 750, 14.0, seed=7, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
prm = __import__("ngspeciesid_amd._capi", fromlist=["cluster_params"]).cluster_params(k=13, w=20, p_shared=select_p_table(13, 20))
rep_of, herr, status, counters = api.cluster_greedy(rs, prm, acc_rank=np.asarray(rd["orig"], dtype=np.uint32))
truths = [s.tobytes().decode() for s in sp]
hrs = ReadSet(rd["seq"].cpu().numpy(), rd["qual"].cpu().numpy(), rd["off"].cpu().numpy().astype(np.uint64))
members = np.nonzero(rep_of == 1)[0]
members = np.concatenate(([1], members[members != 1]))
sub = subset_reads(hrs, members)
n = sub.n
print("cluster of", n)


def go(D=8, band=128, trim=2, m=None, label=""):
    m = m or n
    s2 = subset_reads(sub, np.arange(m))
    draft = api.poa_consensus(s2, [0, m], poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=D, band=band))[0]
    ti = int(np.argmin([edit_distance(draft, t) for t in truths])); T = truths[ti]
    print("%s D=%d band=%d trim=%d m=%d: draft len %d ed %d" % (label, D, band, trim, m, len(draft), edit_distance(draft, T)))
    print("   truth end  ", T[-40:]); print("   draft end  ", draft[-48:]); print("   truth start", T[:40]); print("   draft start", draft[:48])
    bb = draft
    for it in range(3):
        pol, used = api.polish(ReadSet.from_strings([bb]), s2, [0, m], polish_params(iters=1, k=13, w=20, tile_depth=D, band=band, trim=trim, aln_mode=2, stop_when_stable=0))
        bb = pol[0]
        print("   iter %d len %d ed %d used %d   end %s" % (it, len(bb), edit_distance(bb, T), used[0], bb[-48:]))
    return bb, T


bb, T = go()
# where is the internal difference?
for i in range(min(len(bb), len(T))):
    if bb[i] != T[i]:
        print("first difference at", i, "polished", bb[max(0, i - 10):i + 12], "truth", T[max(0, i - 10):i + 12]); break
go(trim=1, label="trim1"); go(trim=0, label="trim0")
go(band=256, label="band256")
go(D=16, label="D16")
go(D=6, label="D6")
