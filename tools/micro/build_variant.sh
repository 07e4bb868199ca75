#!/bin/bash
# dev helper (build container): build_alt/libngsid_hip_<name>.so = the library with k_poa.hip compiled with extra -D flags.   tools/micro/build_variant.sh lt -DPOA_LT=1
set -e
NAME=$1; shift
cd "$(dirname "$0")/../../ngspeciesid_amd/csrc"
mkdir -p ../../build_alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w "$@" -c -o ../../build_alt/k_poa_$NAME.o ${SRC:-k_poa.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_alt/libngsid_hip_$NAME.so ../../build_alt/k_poa_$NAME.o $(ls *.o | grep -v '^k_poa.o$')
