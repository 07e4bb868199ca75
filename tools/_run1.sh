cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_align_paired.py tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_stress.py -m gpu -q -x -k "align or cluster or pipeline" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], d['config']['kernel_ms_per_step']['k_sg_align'], d['config']['check'])"
