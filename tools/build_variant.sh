#!/bin/bash
# A/B builds of libhip_ad_rgb.so: tools/build_variant.sh <name> <extra hipcc flags...>  -> gpurun_variants/lib_<name>.so
set -e
cd "$(dirname "$0")/../mitsuba3_amd/csrc"
NAME=$1; shift
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function "$@" -shared \
  -o ../../tools/variants/lib_$NAME.so har_kernels.hip har_refit.hip har_capi.hip har_multi.hip har_scene_host.cpp har_accel_build.cpp har_host.cpp har_mesh_io.cpp har_mesh_formats.cpp har_image_io.cpp har_scalar.cpp -lz -ldl
