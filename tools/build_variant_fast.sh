#!/bin/bash
# A/B build that recompiles ONLY har_kernels.hip with the extra flags and links it with the default build's other objects (mitsuba3_amd/csrc/obj, `make` first):
# tools/build_variant_fast.sh <name> <extra hipcc flags...>  -> tools/variants/lib_<name>.so   (for switches that only har_kernels.hip reads)
set -e
cd "$(dirname "$0")/../mitsuba3_amd/csrc"
NAME=$1; shift
mkdir -p ../../tools/variants obj_variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o obj_variants/har_kernels_$NAME.o har_kernels.hip
/opt/rocm/bin/hipcc $FLAGS -shared -o ../../tools/variants/lib_$NAME.so obj_variants/har_kernels_$NAME.o $(ls obj/*.o | grep -v har_kernels.o) -lz -ldl
