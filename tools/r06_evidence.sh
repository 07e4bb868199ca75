#!/bin/bash
# GPU box: the round-6 evidence set in one call -> gpurun_out/r6/ (summaries are copied into profiles/ afterwards).   NGSID_COMMIT=<rev> bash tools/r06_evidence.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
bash tools/r06_profiles.sh > $O/profiles.log 2>&1
bash tools/r06_timeline.sh --warmup 3 > $O/timeline.log 2>&1
cd $R
bash tools/r3_configs.sh > $O/r06_other_configs_one_gpu.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_1m_steps20.json 2> $O/r06_bench_1m_steps20.err
NGSID_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --reads 480000 --scaling strong --check-membership --steps 2 --warmup 1 --no-cpu-baseline > $O/r06_two_ranks_one_gpu_strong_c3_480k.json 2> $O/two_ranks.err
tail -3 $O/profiles.log | cut -c1-600; tail -2 $O/timeline.log | cut -c1-300; cat $O/r06_other_configs_one_gpu.txt | cut -c1-300; cut -c1-300 $O/r06_bench_1m_steps20.json; cut -c1-300 $O/r06_two_ranks_one_gpu_strong_c3_480k.json
