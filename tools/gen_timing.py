#!/usr/bin/env python3
"""Dev helper (GPU box): per-generation timing and wave-time split of one workload's frame
(RTX_DEBUG_GEN_TIMING: once with counters on, once in the default mode).
Usage: tools/gen_timing.py <workload> [reduce_factor]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "dragon_metal"
f = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
ds.set_debug(_abi.DEBUG_GEN_TIMING)
cam, _ = hs.camera(0)
if f > 1:
    cam = bench.reduced_camera(cam, f)
ds.render(cam)
print("==== counters on", file=sys.stderr, flush=True)
ds.set_counters(True)
_, _, st = ds.render(cam)
print({k: st[k] for k in ("rays", "node_visits", "tlas_node_visits", "tri_tests", "world_box_tests", "object_space_traversals")}, file=sys.stderr)
print("==== default mode", file=sys.stderr, flush=True)
ds.set_counters(False)
for _ in range(2):
    _, _, st = ds.render(cam)
print("kernel_ms", st["kernel_ms"], "trace_ms", st["trace_ms"], file=sys.stderr)
