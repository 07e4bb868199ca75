#!/bin/bash
for d in ${1:-6 7 8}; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step --tile-depth $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d', d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], {k:v for k,v in d['config']['kernel_ms_per_step'].items() if v > 4}, d['config']['check']['consensus_edit_distance_vs_truth'])"
done
