#!/bin/bash
# single-rank RCCL path (NGSID_FORCE_DIST=1): the sharded code path with its collectives on one GPU, weak and strong + membership check
export MASTER_ADDR=127.0.0.1 NGSID_FORCE_DIST=1
for mode in "--scaling weak" "--scaling strong --check-membership"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step $mode 2>gpurun_out/r3_nccl1.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['scaling'], d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], d['config']['check'])" || tail -5 gpurun_out/r3_nccl1.err
done
