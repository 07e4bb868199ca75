"""dev tool: polish a backbone that carries junk overhangs (the mu=14 failure shape) with CPU-generated reads of that species"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import runtime, synth
from ngspeciesid_amd._capi import ReadSet, polish_params
from ngspeciesid_amd.hostutil import subset_reads
from util_seq import edit_distance
api = runtime.get_api(0)
n = int(sys.argv[1]); mu = float(sys.argv[2]); seeds = [int(x) for x in sys.argv[3].split(",")]
sps = synth.make_species(5, 750, 0.15, seed=1)
sp = [s for s in sps if s.tobytes().decode().endswith("GTAACGG")]
truth = sp[0].tobytes().decode()
for seed in seeds:
    rd = synth.make_reads(sp, n, mu=mu, seed=seed + 100)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    for m in (n, n // 2, n // 4, n // 8, n // 16, n // 64):
        sub = subset_reads(rs, np.arange(m))
        bb = "CC" + truth + "GCCATAAATG"
        pol, used = api.polish(ReadSet.from_strings([bb]), sub, [0, m], polish_params(iters=1, k=13, w=20, tile_depth=8, band=128, trim=2, aln_mode=2, stop_when_stable=0))
        print("seed %d m %d: len %d (truth %d) ed %d end %s" % (seed, m, len(pol[0]), len(truth), edit_distance(pol[0], truth), pol[0][-20:]), flush=True)
