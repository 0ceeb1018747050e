#!/usr/bin/env python3
"""Dev helper (GPU box): rays and ms per frame of N views around a workload's scene (the frames of
the turntable that bench.py's whole-frame mode times), all on one GPU: how much the views differ.
Usage: tools/orbit_cost.py [workload] [N]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "dragon_metal"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam0, _ = hs.camera(0)
for r in range(n):
    cam = bench.orbit_camera(cam0, 360.0 * r / n)
    ds.render(cam)
    ms, rays = [], 0
    for _ in range(5):
        _, _, st = ds.render(cam)
        ms.append(st["wall_ms"] if "wall_ms" in st else st["kernel_ms"])
        rays = int(st["rays"])
    ms.sort()
    print(f"view {r}/{n}: rays {rays} kernel_ms {ms[len(ms) // 2]:.3f} Mrays/s {rays / ms[len(ms) // 2] / 1e3:.1f}", flush=True)
