#!/bin/bash
# GPU box: A/B of the lanes of the consensus / polishing calls (_capi.Api lanes: NGSID_LANES = 1 | 2 | ...) on the bench workload, one box, back to back.
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
for L in ${LANES:-1 2 1 2}; do
  NGSID_LANES=$L timeout 600 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-cli ${EXTRA} 2> $O/ab_lanes.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['config']['kernel_ms_per_step']; r = d['roofline']
print('lanes %-3s %9.0f reads/s %7.2f ms/step | stages %s | poa %.1f sg %.1f ed %.1f | frac %s one_lane %s | edits %s' % ('$L', d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], k.get('k_poa_tile', 0), k.get('k_sg_align', 0), k.get('k_ed_align', 0), r['frac'], json.dumps(r.get('one_lane', {}).get('kernel_ms_per_step')), d['config']['check']['consensus_edit_distance_vs_truth']))"
  tail -2 $O/ab_lanes.err | grep -v amdgpu.ids | cut -c1-300
done
