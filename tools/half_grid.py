#!/usr/bin/env python3
"""Dev helper (GPU box): a workload's frame with the trace grid at two waves per SIMD and at one
(RTX_DEBUG_HALF_GRID): if half the waves take nearly twice the time the kernel is bound by per-wave
latency, not by issue.  Usage: tools/half_grid.py <workload>"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "dragon_metal"
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
for flags, label in ((0, "two waves per SIMD"), (_abi.DEBUG_HALF_GRID, "one wave per SIMD"), (_abi.DEBUG_GEN_TIMING, "generations"),
                     (_abi.DEBUG_GEN_TIMING | _abi.DEBUG_HALF_GRID, "generations, one wave per SIMD")):
    ds.set_debug(flags)
    ds.render(cam)
    t0 = time.perf_counter()
    for _ in range(5):
        _, _, st = ds.render(cam)
    print(f"{name}: {label}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/frame, trace {st['trace_ms']:.3f} ms, kernels {st['kernel_ms']:.3f} ms",
          file=sys.stderr, flush=True)
