#!/bin/bash
# Dev helper (GPU box): the scene view of some workloads through every library in build_diag/ and the product
# build, three rounds each (ms per frame, kernels' ms).  Usage: tools/variant_sweep.sh <workload>...
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in raytracer_amd/lib/librtx_amd.so build_diag/librtx_amd_*.so; do
    for w in "$@"; do
      RTX_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 - "$w" "$lib" <<'PY' || exit 1
import sys, time
sys.path.insert(0, ".")
import bench
import raytracer_amd as rt
name, lib = sys.argv[1], sys.argv[2]
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
for _ in range(3):
    ds.render(cam)
n = 5 if name == "grass_desert" else 20
t0 = time.perf_counter()
for _ in range(n):
    _, _, st = ds.render(cam)
print(f"{lib.split('/')[-1]:28s} {name:20s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms/frame  trace {st['trace_ms']:.3f}", flush=True)
PY
    done
  done
done
