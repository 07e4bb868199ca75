#!/bin/bash
# GPU box: A/B of library builds on the bench step, same box, interleaved twice:  bash tools/ab_poa.sh base lt ...   (base = the shipped library; others = build_alt/libngsid_hip_<name>.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for V in "$@"; do
  LIB=$R/build_alt/libngsid_hip_$V.so; [ $V = base ] && LIB=$R/ngspeciesid_amd/libngsid_hip.so
  timeout 600 python $R/tools/micro/bench_with_lib.py $LIB --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms_per_step']; print('$V', 'step', d['ms_per_step'], 'poa', k['k_poa_tile'], 'sg', k['k_sg_align'], 'ed', k['k_ed_align'], d['config']['check']['consensus_edit_distance_vs_truth'])"
done; done
