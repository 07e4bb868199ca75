mkdir -p gpurun_out/r5
timeout 900 python tools/r05_c4_full.py 10000000 c4 > gpurun_out/r5/c4_full.log 2>&1; tail -c 300 gpurun_out/r5/c4_full.log
timeout 900 python tools/r05_c4_full.py 2000000 c5 > gpurun_out/r5/c5_full.log 2>&1; tail -c 300 gpurun_out/r5/c5_full.log
