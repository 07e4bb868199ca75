"""dev tool: run the mu=14 million-read case, isolate the cluster whose polished consensus differs from its amplicon, shrink it by
prefix halving and dump the smallest failing read set (npz) for CPU-side debugging with the oracle."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000; mu = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "r2", "noisy_fail.npz")
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp, rd = bench.gen_sorted_reads(api, n, 5, 750, mu, seed=seed, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3, tile_depth=8, band=128, p_shared=select_p_table(13, 20), polish_stop_when_stable=False)
truths = [s.tobytes().decode() for s in sp]
hrs = ReadSet(rd["seq"].cpu().numpy(), rd["qual"].cpu().numpy(), rd["off"].cpu().numpy().astype(np.uint64))
rep_of = res["rep_of"]; spc = rd["species"].cpu().numpy()


def run(sub, m):
    s2 = subset_reads(sub, np.arange(m))
    draft = api.poa_consensus(s2, [0, m], poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=8, band=128))[0]
    pol, used = api.polish(ReadSet.from_strings([draft]), s2, [0, m], polish_params(iters=3, k=13, w=20, tile_depth=8, band=128, trim=2, aln_mode=2, stop_when_stable=0))
    return draft, pol[0]


for c in res["centers"]:
    ti = int(np.argmin([edit_distance(c[3], t) for t in truths])); e = edit_distance(c[3], truths[ti])
    members = np.nonzero(rep_of == c[1])[0]
    members = np.concatenate(([c[1]], members[members != c[1]]))
    print("centre %d: %d reads, polished ed %d, purity %.4f" % (c[1], len(members), e, float((spc[members] == ti).mean())), flush=True)
    if e == 0: continue
    sub = subset_reads(hrs, members)
    d, p = run(sub, sub.n)
    print("  isolated: draft ed %d polished ed %d (same as pipeline: %s)" % (edit_distance(d, truths[ti]), edit_distance(p, truths[ti]), p == c[3]), flush=True)
    m = sub.n
    while m > 64:
        h = m // 2
        d, p = run(sub, h); e2 = edit_distance(p, truths[ti])
        print("  prefix %d: draft ed %d polished ed %d len %d" % (h, edit_distance(d, truths[ti]), e2, len(p)), flush=True)
        if e2 == 0: break
        m = h
    # refine between m/2 (ok) and m (fails) with a few more steps
    lo, hi = m // 2, m
    for _ in range(6):
        mid = (lo + hi) // 2
        if mid == lo: break
        d, p = run(sub, mid)
        if edit_distance(p, truths[ti]) == 0: lo = mid
        else: hi = mid
    print("  smallest failing prefix found: %d" % hi, flush=True)
    s2 = subset_reads(sub, np.arange(hi))
    if len(s2.seq) < 30_000_000:
        np.savez_compressed(out, seq=s2.seq, qual=s2.qual, off=s2.off, truth=np.frombuffer(truths[ti].encode(), dtype=np.uint8))
        print("  saved %s (%d reads)" % (out, hi))
    break
