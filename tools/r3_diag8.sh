#!/bin/bash
# diagnose the 8-ranks-on-one-GPU composed run: small set, per-rank progress trace, hard time limit
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 NGSID_DIST_BACKEND=gloo NGSID_BENCH_TRACE=1
timeout -s KILL ${2:-240} python -m torch.distributed.run --nnodes=1 --nproc-per-node ${3:-8} --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus ${3:-8} --steps 1 --warmup 0 --config ${4:-c4} --reads ${1:-400000} --scaling strong --check-membership --no-cpu-baseline --no-extra-step > gpurun_out/r3_diag8.out 2> gpurun_out/r3_diag8.err
echo rc=$?
grep "bench rank" gpurun_out/r3_diag8.err | tail -40
tail -c 600 gpurun_out/r3_diag8.out
