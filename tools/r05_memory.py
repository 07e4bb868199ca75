"""profiles/r05_memory.json: device memory of ONE context at BASELINE's configurations, from the runs of tools/r05_c4_full.py (C4 at 10 M reads, C5 at 2 M reads:
gpurun_out/r5/r05_full_*.json) and of bench.py (C3: config.hbm_gb of gpurun_out/r5/r05_bench_1m.json).  Numbers = bytes handed out by the library's own allocator
(ngsid_profile_read: hbm_peak_bytes, mem_* lines); the read set itself is torch's."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R5 = os.path.join(ROOT, "gpurun_out", "r5")
out = {"_what": __doc__, "configs": {}}
b = json.loads(open(os.path.join(R5, "r05_bench_1m.json")).read().strip().splitlines()[-1])
out["configs"]["c3"] = {"reads": b["config"]["reads_clustered_per_gpu"], "read_set_gb": round(2 * 750e6 / 1e9, 2), "reads_per_s": b["value"], "library_hbm_peak_gb": b["config"]["hbm_gb"]["peak_in_timed_steps"],
                        "held_after_gb": b["config"]["hbm_gb"]["held_after"], "context_scratch_gb_by_purpose": b["config"]["hbm_gb"].get("context_scratch_by_purpose")}
for name, n in (("c4", 10000000), ("c5", 2000000)):
    f = os.path.join(R5, "r05_full_%s_%d.json" % (name, n))
    if not os.path.exists(f): continue
    d = json.load(open(f))
    out["configs"][name] = {k: d.get(k) for k in ("reads", "bases", "species", "one_context", "wall_s_first_pass_incl_allocation", "wall_s", "reads_per_s", "stage_s", "every_consensus_equals_its_amplicon", "clusters_pure_and_complete", "read_set_gb",
                                                   "library_hbm_peak_gb", "library_hbm_held_after_gb", "device_total_gb", "context_scratch_gb_by_purpose", "poa_tiles_redone")}
comp = os.path.join(ROOT, "gpurun_out", "composed_c4_8_shards_one_gpu.json")
if os.path.exists(comp): out["c4_composed_eight_shards_on_one_gpu"] = json.load(open(comp))
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_memory.json"), "w"), indent=1)
print(json.dumps({k: (v.get("library_hbm_peak_gb"), v.get("reads_per_s")) for k, v in out["configs"].items()}))
