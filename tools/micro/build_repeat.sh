#!/bin/bash
# dev helper (build container): timing builds of the POA tile kernel that run ONE phase twice (same results) -> build_alt/libngsid_hip_rep{1,2,3,4}.so.
# The time / counter difference to the shipped build is that phase's cost in the real, contended setting (tools/micro/bench_with_lib.py runs bench.py on one of them).
set -e
cd "$(dirname "$0")/../../ngspeciesid_amd/csrc"
mkdir -p ../../build_alt
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w"
for R in ${@:-1 2 3 4}; do
  /opt/rocm/bin/hipcc $FLAGS -DPOA_REPEAT=$R -c -o ../../build_alt/k_poa_rep$R.o k_poa.hip &
done
wait
for R in ${@:-1 2 3 4}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_alt/libngsid_hip_rep$R.so ../../build_alt/k_poa_rep$R.o $(ls *.o | grep -v '^k_poa.o$')
done
ls -la ../../build_alt/
