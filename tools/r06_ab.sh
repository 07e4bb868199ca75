#!/bin/bash
# GPU box: A/B of context options on the bench workload (one box, back to back).  usage: bash tools/r06_ab.sh "name=value[,name=value]" ...   ("" = defaults)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
for OPT in "$@"; do
  NGSID_OPTIONS="$OPT" timeout 600 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-cli --no-extra-step 2> $O/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['config']['kernel_ms_per_step']
print('%-40s %9.0f reads/s %7.2f ms/step | stages %s | poa %.1f sg %.1f ed %.1f | edits %s' % ('$OPT' or '(defaults)', d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], k.get('k_poa_tile', 0), k.get('k_sg_align', 0), k.get('k_ed_align', 0), d['config']['check']['consensus_edit_distance_vs_truth']))"
done
