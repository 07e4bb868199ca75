#!/bin/bash
# round-3 dev helper: consensus / edge / noisy subsets + bench with device-driven and host-driven hierarchy levels
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_edge.py tests/test_anchor_poa_hand.py tests/test_anchor_gotoh.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_s1_tests.log 2>&1
tail -15 gpurun_out/r3_s1_tests.log
for mode in dev host; do
  if [ $mode = host ]; then export NGSID_OPTIONS="poa_host_levels=1"; else unset NGSID_OPTIONS; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step > gpurun_out/r3_s1_bench_$mode.json 2> gpurun_out/r3_s1_bench_$mode.err
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3_s1_bench_$mode.json').read().strip().splitlines()[-1]); print('$mode', d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], {k:v for k,v in d['config']['kernel_ms_per_step'].items() if 'poa' in k}, d['config']['check'])" || tail -5 gpurun_out/r3_s1_bench_$mode.err
done
