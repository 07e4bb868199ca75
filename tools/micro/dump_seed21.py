"""dev tool: write the reads of the cluster that misses at depth 6 (mu 14, seed 21) to gpurun_out/seed21_cluster.npz for a replay on the CPU oracle"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, pipeline, strand
from ngspeciesid_amd._capi import ReadSet, polish_params
from ngspeciesid_amd.ptable import select_p_table
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp, rd = bench.gen_sorted_reads(api, 200000, 5, 750, 14.0, seed=21, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
truths = [s.tobytes().decode() for s in sp]
res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=0, tile_depth=6, band=0, p_shared=select_p_table(13, 20), do_polish=False)
rep = res["rep_of"]
c = [c for c in res["centers"] if c[1] == 0][0]
ids = np.nonzero(np.isin(rep, c[4]))[0].astype(np.uint32)
sub = strand.fetch_reads(rs, ids)
t = [t for t in truths if t.endswith("CACTCCTCAACCG")][0]
pol, used = api.polish(ReadSet.from_strings([t]), sub, [0, sub.n], polish_params(iters=1, k=13, w=20, tile_depth=6, band=0, trim=2, stop_when_stable=0))
print("hip from the truth:", len(t), len(pol[0]), pol[0][-20:], t[-20:])
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "seed21_cluster.npz"), seq=sub.seq, qual=sub.qual, off=sub.off, truth=np.frombuffer(t.encode(), dtype=np.uint8), hip=np.frombuffer(pol[0].encode(), dtype=np.uint8))
