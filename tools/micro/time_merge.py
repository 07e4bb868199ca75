"""dev tool: cost of the multi-GPU representative merge (distributed.merge_representatives) emulated in ONE process.

    python tools/micro/time_merge.py [shards] [reads_per_shard] [species]

Clusters `shards` independent synthetic shards one after the other on cuda:0 (what each rank does locally), collects the
payloads the all-gather would deliver, and times the tree merge every rank replays, for world = 2, 4, ... shards.
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from ngspeciesid_amd import runtime, distributed
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
NSP = int(sys.argv[3]) if len(sys.argv) > 3 else 5          # species (C4: 50)
api = runtime.get_api(0); dev = torch.device("cuda", 0)
prm = cluster_params(k=13, w=20, p_shared=select_p_table(13, 20))
gathered = []
for r in range(W):
    sp, rd = bench.gen_sorted_reads(api, n, NSP, 750, 17.0, seed=7 + r, device=dev)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    acc = np.asarray(rd["orig"], dtype=np.uint32)
    t0 = time.perf_counter()
    rep, herr, st, cnt = api.cluster_greedy(rs, prm, acc_rank=acc)
    t1 = time.perf_counter()
    mine, payload = distributed.representative_payload(rs, rep, herr, np.asarray(rd["score"], dtype=np.float64), acc)
    t2 = time.perf_counter()
    print("shard %d: cluster %.3fs, %d representatives, payload %.3fs (%.1f KB)" % (r, t1 - t0, len(mine), t2 - t1, (payload["seq"].nbytes * 2) / 1e3), flush=True)
    gathered.append(payload)
    del rs, rd
w = 2
while w <= W:
    for rep_i in range(2):
        t0 = time.perf_counter()
        out = distributed.merge_representatives(api, gathered[:w], prm, w)
        dt = time.perf_counter() - t0
    print("world %d: merge %.3fs   (%d representatives -> %d)" % (w, dt, len(out[0]), len(np.unique(out[0]))), flush=True)
    w *= 2
