#!/bin/bash
# GPU box: A/B of library builds on the bench workload (one box, alternating).  usage: bash tools/r06_ab_lib.sh <lib.so> <lib.so> ...   ("main" = ngspeciesid_amd/libngsid_hip.so)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
for ROUND in 1 2; do for L in "$@"; do
  LIB=$L; [ "$L" = main ] && LIB=ngspeciesid_amd/libngsid_hip.so
  timeout 600 python tools/micro/bench_with_lib.py $LIB --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-cli --no-extra-step 2> $O/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['config']['kernel_ms_per_step']
print('%-40s %9.0f reads/s %7.2f ms/step | poa %.2f sg %.1f ed %.1f | stages %s | edits %s' % ('$L', d['value'], d['ms_per_step'], k.get('k_poa_tile', 0), k.get('k_sg_align', 0), k.get('k_ed_align', 0), d['config']['stage_s_per_step'], d['config']['check']['consensus_edit_distance_vs_truth']))"
done; done
