#!/bin/bash
# dev helper (build container): build_alt/libngsid_hip_<name>.so = the library with ONE source (default k_poa.hip; SRC=k_ed_align.hip ...) compiled with extra -D flags.
#   tools/micro/build_variant.sh lt -DPOA_LT=1        SRC=k_ed_align.hip tools/micro/build_variant.sh ck8 -DED_CK=8
set -e
NAME=$1; shift
SRC=${SRC:-k_poa.hip}; BASE=${SRC%.hip}
cd "$(dirname "$0")/../../ngspeciesid_amd/csrc"
mkdir -p ../../build_alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w "$@" -c -o ../../build_alt/${BASE}_$NAME.o $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_alt/libngsid_hip_$NAME.so ../../build_alt/${BASE}_$NAME.o $(ls *.o | grep -v "^${BASE}.o\$")
