#!/bin/bash
# Dev helper: a second build of the device library with extra -D flags, for experiments and diagnosis:
#   tools/build_variant.sh <name> [-DFLAG=VALUE ...]   ->  raytracer_amd/lib/librtx_amd_<name>.so
# Select it at run time with RTX_AMD_LIB=<path>.
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-strict-aliasing \
  -munsafe-fp-atomics -fPIC -Iinclude "$@" -c -o /tmp/rtx_core_$name.o raytracer_amd/csrc/rtx_core.hip &&
/opt/rocm/bin/hipcc -shared -fPIC -o raytracer_amd/lib/librtx_amd_$name.so /tmp/rtx_core_$name.o raytracer_amd/lib/rtx_bvh_build.o
