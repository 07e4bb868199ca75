#!/bin/bash
# the other BASELINE.json configurations on one GPU (per-GPU shard shapes of C4 / C5) and the noisy ONT profile: one line each
for c in "--config c2" "--config c4" "--config c5" "--config c3 --mu 14"; do
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ch=d['config']['check']; print('$c', '|', d['value'], 'reads/s', d['ms_per_step'], 'ms', d['config']['stage_s_per_step'], 'centres', ch['centers'], 'wrong', sum(1 for x in ch['consensus_edit_distance_vs_truth'] if x), 'purity', ch['cluster_purity'], 'f_aln', d['config']['f_aln'])"
done
