#!/bin/bash
# Dev helper: a diagnostic build of librtx_amd.so with extra -D flags, beside the product build.
# Usage: tools/diag_build.sh <name> <flags...>   ->  build_diag/librtx_amd_<name>.so  (use it with RTX_AMD_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_diag
flags=$(make -pn product 2>/dev/null | sed -n 's/^DEVFLAGS *:\?= *//p' | head -n 1)
hipcc=$(make -pn product 2>/dev/null | sed -n 's/^HIPCC *:\?= *//p' | head -n 1)
${hipcc:-hipcc} $flags "$@" -c -o build_diag/rtx_core_$name.o raytracer_amd/csrc/rtx_core.hip
${hipcc:-hipcc} -shared -fPIC -o build_diag/librtx_amd_$name.so build_diag/rtx_core_$name.o raytracer_amd/lib/rtx_bvh_build.o
rm -f build_diag/rtx_core_$name.o
ls -la build_diag/librtx_amd_$name.so
