"""GPU box (round 5): BASELINE config C4 at its STATED size - 10 M x 750 bp, 50 species, abundance_ratio 0.005 - in ONE context on one
MI355X: pure and complete clusters, every polished consensus == its amplicon; wall time per stage and the library's device-memory high-water mark.
    python tools/r05_c4_full.py [reads] [config]      -> gpurun_out/r5/r05_full_<config>_<reads>.json
"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ctypes as C
import bench
from ngspeciesid_amd import runtime, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from test_gpu_fullsize import _check_clusters

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
name = sys.argv[2] if len(sys.argv) > 2 else "c4"
cfg = bench.CONFIGS[name]
api = runtime.get_api(0)
dev = torch.device("cuda", 0)
abundance = [cfg["geometric"] ** i for i in range(cfg["species"])] if cfg["geometric"] else None
t0 = time.perf_counter()
sp, rd = bench.gen_sorted_reads(api, n, cfg["species"], cfg["length"], cfg["mu"], seed=7, device=dev, abundance=abundance, k=cfg["k"])
torch.cuda.synchronize(); torch.cuda.empty_cache()
t_gen = time.perf_counter() - t0
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
free0, tot0 = torch.cuda.mem_get_info()
kw_ = dict(acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=cfg["k"], w=cfg["w"], abundance_ratio=cfg["abundance_ratio"], racon_iter=3, tile_depth=pipeline.TILE_DEPTH, band=0,
           p_shared=select_p_table(cfg["k"], cfg["w"]), polish_stop_when_stable=False)
t0 = time.perf_counter()
pipeline.run_hot_path(api, rs, rd["score"], **kw_)               # first pass: the context allocates its scratch (tens of GB of hipMalloc take seconds); timed separately
dt_cold = time.perf_counter() - t0
api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
T = {}
t0 = time.perf_counter()
res = pipeline.run_hot_path(api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=cfg["k"], w=cfg["w"], abundance_ratio=cfg["abundance_ratio"], racon_iter=3,
                            tile_depth=pipeline.TILE_DEPTH, band=0, p_shared=select_p_table(cfg["k"], cfg["w"]), polish_stop_when_stable=False, timings=T)
dt = time.perf_counter() - t0
buf = C.create_string_buffer(1 << 16)
api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(len(buf)))
api.lib.ngsid_profile_enable(api.ctx, C.c_int32(0))
kern = {}
for line in buf.value.decode().splitlines():
    nm, cnt, ms = line.split(); kern[nm] = (int(cnt), float(ms))
free1, _ = torch.cuda.mem_get_info()
_check_clusters(rd, res, cfg["species"], 0.995)
truths = sorted(s.tobytes().decode() for s in sp)
got = sorted(c[3] for c in res["centers"])
exact = got == truths
out = dict(config=name, reads=int(rs.n), bases=int(rd["off"][-1].item()), species=cfg["species"], one_context=True, wall_s_first_pass_incl_allocation=round(dt_cold, 3), wall_s=round(dt, 3), reads_per_s=round(rs.n / dt, 1), generation_s=round(t_gen, 1),
           stage_s={k_: round(v, 3) for k_, v in T.items()}, centres=len(got), every_consensus_equals_its_amplicon=bool(exact), clusters_pure_and_complete=True,
           library_hbm_peak_gb=round(kern.get("hbm_peak_bytes", (0, 0))[0] / 1e9, 2), library_hbm_held_after_gb=round(kern.get("hbm_live_bytes", (0, 0))[0] / 1e9, 2),
           device_free_before_gb=round(free0 / 1e9, 1), device_free_after_gb=round(free1 / 1e9, 1), device_total_gb=round(tot0 / 1e9, 1),
           read_set_gb=round(2 * int(rd["off"][-1].item()) / 1e9, 2),
           context_scratch_gb_by_purpose={k_[4:]: round(v[0] / 1e9, 2) for k_, v in kern.items() if k_.startswith("mem_")},
           kernel_ms={k_: round(v[1], 1) for k_, v in kern.items() if v[1] > 0}, poa_tiles_redone=kern.get("poa_band_redo_tiles", (0, 0))[0])
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5", "r05_full_%s_%d.json" % (name, n)), "w"), indent=1)
print(json.dumps(out))
assert exact, "%d of %d polished sequences differ from their amplicons" % (sum(1 for a, b in zip(got, truths) if a != b), len(truths))
