"""Quick timing of the clustering stage on synthetic reads (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ngspeciesid_amd import runtime, synth
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nsp = int(sys.argv[2]) if len(sys.argv) > 2 else 5
api = runtime.get_api(0)
sp = synth.make_species(nsp, 750, 0.15, seed=1)
t = time.time(); rd = synth.make_reads(sp, n, mu=17.0, seed=7, device="cuda"); torch.cuda.synchronize(); print("gen %.2fs" % (time.time() - t))
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
prm = cluster_params(k=13, w=20, p_shared=select_p_table(13, 20))
for it in range(2):
    t = time.time(); rep, herr, st, cnt = api.cluster_greedy(rs, prm); dt = time.time() - t
    print("cluster %d reads: %.3fs -> %.0f reads/s; counters %s; clusters>1: %d" % (n, dt, n / dt, cnt.tolist(), int((np.bincount(rep) > 1).sum())))
spc = rd["species"].cpu().numpy()
pur = sum(np.bincount(spc[rep == r]).max() for r in np.unique(rep)) / n
print("purity %.5f nreps %d" % (pur, len(np.unique(rep))))
