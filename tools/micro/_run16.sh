mkdir -p gpurun_out/r5
export MASTER_ADDR=127.0.0.1 NGSID_DIST_BACKEND=gloo
for W in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29600+W)) bench.py --gpus $W --steps 1 --warmup 0 --reads 480000 --scaling strong --check-membership --no-cpu-baseline --no-extra-step > gpurun_out/r5/strong_${W}proc_one_gpu.json 2> gpurun_out/r5/strong_${W}proc_one_gpu.err; echo "W=$W rc=$?"; tail -c 900 gpurun_out/r5/strong_${W}proc_one_gpu.json | head -c 900; echo
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cli --no-extra-step 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['config']['stage_s_per_step'],d['config']['kernel_ms_per_step'])"
