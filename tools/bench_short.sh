#!/bin/bash
# dev helper: GPU tests + a short bench, compact output
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 800 python bench.py --reads ${1:-200000} --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['stage_s_per_step'], d['config']['kernel_ms_per_step'], d['config']['check'])"
