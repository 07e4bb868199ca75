#!/bin/bash
# round-3 dev helper: POA / polish parity subset, then the bench line's POA figures
timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_anchor_poa_hand.py tests/test_gpu_edge.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], {k:v for k,v in d['config']['kernel_ms_per_step'].items() if 'poa' in k}, d['config']['check']['consensus_edit_distance_vs_truth'])"
