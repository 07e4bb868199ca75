#!/bin/bash
# GPU box: lanes x context options (does a POA launch that leaves wave slots free let the other lane's HBM-bound aligner run beside it?).  usage: bash tools/r06_ab_lanes_opt.sh "lanes:opt=val,..." ...
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
for X in "$@"; do
  L=${X%%:*}; OPT=${X#*:}; [ "$OPT" = "$X" ] && OPT=""
  NGSID_OPTIONS="$OPT" NGSID_LANES=$L timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-cli --no-extra-step 2> $O/ab_lanes_opt.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['config']['kernel_ms_per_step']
print('%-36s %9.0f reads/s %7.2f ms/step | stages %s | poa %.1f ed %.1f | edits %s' % ('$X', d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], k.get('k_poa_tile', 0), k.get('k_ed_align', 0), d['config']['check']['consensus_edit_distance_vs_truth']))"
done
