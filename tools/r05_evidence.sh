#!/bin/bash
# GPU box: the round-5 evidence set in one call -> gpurun_out/r5/ (summaries are copied into profiles/ afterwards).   NGSID_COMMIT=<rev> bash tools/r05_evidence.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
bash tools/r05_profiles.sh > $O/profiles.log 2>&1
bash tools/r05_timeline.sh > $O/timeline.log 2>&1
cd $R
bash tools/r3_configs.sh > $O/r05_other_configs_one_gpu.txt 2>&1
timeout 900 python tools/r05_c4_full.py 10000000 c4 > $O/c4_full.log 2>&1
timeout 900 python tools/r05_c4_full.py 2000000 c5 > $O/c5_full.log 2>&1
python tools/r05_memory.py > $O/memory.log 2>&1
tail -3 $O/profiles.log | cut -c1-600; tail -2 $O/timeline.log | cut -c1-400; cat $O/r05_other_configs_one_gpu.txt | cut -c1-300; cat $O/memory.log
