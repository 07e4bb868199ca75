#!/bin/bash
# GPU box: clustering block cap on the NOISY profile (mu = 14, ~500 representatives): bash tools/r06_ab_mu14.sh "cluster_block_cap=65536" ...
for OPT in "$@"; do
NGSID_OPTIONS="$OPT" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step --mu ${MU:-14} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % ('$OPT' or '(default)'), d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], d['config']['kernel_ms_per_step'].get('k_sg_align'))"
done
