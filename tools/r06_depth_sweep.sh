#!/bin/bash
# GPU box: accuracy of the whole path by POA order on SHALLOW clusters: one graph per unit in file order (0:160 = the restated spoa / racon order), fixed tile depths,
# and the adaptive rule of round 6 (aT: units below T sequences as one graph).  -> gpurun_out/r6/r06_tile_depth_sweep.txt
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
SEEDS=${SEEDS:-20}; DEPTHS=${DEPTHS:-0:160,4,6,8,a64,a128,a256}
{
echo "== tools/micro/depth_sweep.py N $SEEDS $DEPTHS 17,13,10: $SEEDS seeds x 5 clusters of N / 5 reads BEFORE the quality filter; 'wrong' adds missing / extra clusters (it can exceed the cluster count at mu 10, where half of the reads are filtered)"
for N in 150 250 500 1000 5000; do
  echo "== n $N ($((N / 5)) reads per cluster)"
  timeout 1500 python tools/micro/depth_sweep.py $N $SEEDS $DEPTHS 17,13,10 2>&1 | grep -v "^  mu"
done
} > $O/r06_tile_depth_sweep.txt 2>&1
tail -60 $O/r06_tile_depth_sweep.txt
