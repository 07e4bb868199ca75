"""dev tool: single-species noisy clusters (CPU-generated, so the same reads can be rebuilt in the build container): draft + 3 polishing
iterations on the GPU, edit distance to the amplicon; failing sets are bisected to the shortest failing prefix."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import runtime, synth
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL
from ngspeciesid_amd.hostutil import subset_reads
from util_seq import edit_distance
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000; mu = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
seeds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(1, 9))
D = int(sys.argv[4]) if len(sys.argv) > 4 else 8; band = int(sys.argv[5]) if len(sys.argv) > 5 else 128
api = runtime.get_api(0)


def run(rs, m, truth):
    sub = subset_reads(rs, np.arange(m))
    draft = api.poa_consensus(sub, [0, m], poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=D, band=band))[0]
    pol, used = api.polish(ReadSet.from_strings([draft]), sub, [0, m], polish_params(iters=3, k=13, w=20, tile_depth=D, band=band, trim=2, aln_mode=2, stop_when_stable=0))
    return edit_distance(draft, truth), edit_distance(pol[0], truth), len(pol[0]) - len(truth)


for seed in seeds:
    sp = synth.make_species(1, 750, 0.15, seed=seed)
    rd = synth.make_reads(sp, n, mu=mu, seed=seed + 100)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    truth = sp[0].tobytes().decode()
    d, p, dl = run(rs, n, truth)
    print("seed %d n %d: draft ed %d, polished ed %d (len diff %d)" % (seed, n, d, p, dl), flush=True)
    if p > 0:
        m = n
        while m > 64:
            h = m // 2
            d2, p2, dl2 = run(rs, h, truth)
            print("   prefix %d: draft ed %d polished ed %d (len diff %d)" % (h, d2, p2, dl2), flush=True)
            if p2 == 0:
                break
            m = h
        print("   smallest failing prefix (halving) %d" % m, flush=True)
