#!/bin/bash
# GPU box (round 5): where a step's time goes OUTSIDE the kernels.  Kernel trace + memory-copy trace of a short bench run (1 warm-up + 1 timed step): idle gaps of the GPU
# between consecutive operations (kernels AND copies), summed per transition, the 50 largest single gaps with their position in the step, and the k_poa_tile launches in order.
# -> gpurun_out/r6/r06_timeline.json.   Usage: bash tools/r06_timeline.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
rm -rf $O/prof_tl; timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_tl -o tl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step "$@" > $O/prof_tl.log 2>&1
cd $R
python - <<PY
import csv, glob, json, collections, re
fs = glob.glob("gpurun_out/r6/prof_tl/**/*kernel_trace.csv", recursive=True)
def short(nm):
    m = re.search(r"(k_[a-z0-9_]+)", nm)
    return m.group(1) if m else nm.split("(")[0][-40:]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(fs[0]))]
nk = len(rows)
mc = glob.glob("gpurun_out/r6/prof_tl/**/*memory_copy_trace.csv", recursive=True)
ncopy = 0
if mc:
    for r in csv.DictReader(open(mc[0])):
        d = r.get("Direction", r.get("Name", "copy")); rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + d.replace("MEMORY_COPY_", ""))); ncopy += 1
rows.sort(key=lambda x: x[0])
ei = [i for i, r in enumerate(rows) if r[2] == "k_eidx"][-1]
start = [i for i, r in enumerate(rows[:ei]) if r[2] == "k_hpc_minimizers"][-1]
step = rows[start:]
t0 = step[0][0]
busy_end = step[0][1]; tot_gap = 0; per = collections.Counter(); cnt = collections.Counter(); prev = step[0][2]; gaps = []
for s, e, nm in step[1:]:
    g = s - busy_end
    if g > 0:
        tot_gap += g; per[(prev, nm)] += g; cnt[(prev, nm)] += 1; gaps.append((g, busy_end - t0, prev, nm))
    if e > busy_end: busy_end = e; prev = nm
wall = busy_end - t0
kern = collections.Counter(); kc = collections.Counter()
for s, e, nm in step: kern[nm] += e - s; kc[nm] += 1
poa = [round((e - s) / 1e6, 3) for s, e, nm in step if nm.startswith("k_poa_tile")]
gaps.sort(reverse=True)
hist = collections.Counter()
for g, *_ in gaps: hist["<20us" if g < 20e3 else "<100us" if g < 100e3 else "<1ms" if g < 1e6 else ">=1ms"] += g
out = {"_how": "tools/r06_timeline.sh: rocprofv3 --kernel-trace --memory-copy-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step; the timed step = operations from the last clustering call's k_hpc_minimizers launch on; gap = start of an operation minus the latest end of all earlier ones",
       "step_wall_ms": round(wall / 1e6, 2), "gpu_idle_ms": round(tot_gap / 1e6, 2), "kernel_dispatches": sum(1 for r in step if not r[2].startswith("copy:")), "copies": sum(1 for r in step if r[2].startswith("copy:")),
       "idle_ms_by_gap_size": {k: round(v / 1e6, 2) for k, v in hist.items()},
       "idle_ms_by_transition": {"%s -> %s" % k: [round(v / 1e6, 2), cnt[k]] for k, v in per.most_common(30)},
       "largest_gaps_ms_at_ms_prev_next": [[round(g / 1e6, 3), round(at / 1e6, 1), p, n] for g, at, p, n in gaps[:50]],
       "op_ms": {k: [round(v / 1e6, 2), kc[k]] for k, v in kern.most_common(40)},
       "k_poa_tile_launch_ms_in_order": poa}
json.dump(out, open("gpurun_out/r6/r06_timeline.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("step_wall_ms", "gpu_idle_ms", "kernel_dispatches", "copies", "idle_ms_by_gap_size", "idle_ms_by_transition")}))
print(json.dumps(out["largest_gaps_ms_at_ms_prev_next"]))
PY
