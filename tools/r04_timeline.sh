#!/bin/bash
# GPU box: where the step's time goes OUTSIDE the kernels.  One kernel trace of a short bench run (1 warm-up + 1 timed step), then the idle gaps of the
# GPU between consecutive dispatches, summed per (previous kernel -> next kernel), and the duration of every k_poa_tile launch in order (hierarchy tails).
# -> gpurun_out/r4/r04_timeline.json.   Usage: bash tools/r04_timeline.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
rm -rf $O/prof_tl; timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o tl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step "$@" > $O/prof_tl.log 2>&1
cd $R
python - <<PY
import csv, glob, json, collections, re
fs = glob.glob("gpurun_out/r4/prof_tl/**/*kernel_trace.csv", recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(fs[0]))), key=lambda x: x[0])
def short(nm):
    m = re.search(r"(k_[a-z0-9_]+)", nm)
    return m.group(1) if m else nm.split("(")[0][-40:]
# the timed step = everything from the k_hpc_minimizers launch of the LAST clustering call on (k_eidx is launched by clustering only; the polisher launches
# the minimizer kernel too, for its backbones)
ei = [i for i, r in enumerate(rows) if "k_eidx" in r[2]][-1]
start = [i for i, r in enumerate(rows[:ei]) if "k_hpc_minimizers" in r[2]][-1]
step = rows[start:]
busy_end = step[0][1]; tot_gap = 0; per = collections.Counter(); cnt = collections.Counter(); prev = short(step[0][2])
for s, e, nm in step[1:]:
    g = s - busy_end
    if g > 0:
        tot_gap += g; per[(prev, short(nm))] += g; cnt[(prev, short(nm))] += 1
    if e > busy_end: busy_end = e; prev = short(nm)
wall = busy_end - step[0][0]
kern = collections.Counter(); kc = collections.Counter()
for s, e, nm in step: kern[short(nm)] += e - s; kc[short(nm)] += 1
poa = [round((e - s) / 1e6, 3) for s, e, nm in step if "k_poa_tile" in nm]
out = {"_how": "tools/r04_timeline.sh: rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step; the timed step = dispatches from the last clustering call's k_hpc_minimizers launch on; gap = start of a dispatch minus the latest end of all earlier ones",
       "step_wall_ms": round(wall / 1e6, 2), "gpu_idle_ms": round(tot_gap / 1e6, 2), "dispatches": len(step),
       "idle_ms_by_transition": {"%s -> %s" % k: [round(v / 1e6, 2), cnt[k]] for k, v in per.most_common(25)},
       "kernel_ms": {k: [round(v / 1e6, 2), kc[k]] for k, v in kern.most_common(30)},
       "k_poa_tile_launch_ms_in_order": poa}
json.dump(out, open("gpurun_out/r4/r04_timeline.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("step_wall_ms", "gpu_idle_ms", "dispatches", "idle_ms_by_transition")}))
print(poa)
PY
