#!/usr/bin/env python3
"""Dev helper (GPU box): the consistent mode on one workload -- primary hits and picture against its
CPU checker (on the frame's view reduced by `reduce_factor`), and ms/frame of both modes at full size.
Usage: tools/consistent_check.py <workload> [reduce_factor] [frames]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import _checkers as ck  # noqa: E402
import raytracer_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "grass_desert"
f = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
small = bench.reduced_camera(cam, f) if f > 1 else cam

ds.set_consistent(True)
rgb, hits, _ = ds.render(small, want_hits=True)
o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, small, consistent=True)
bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
same, mx = ck.image_diff(rgb, o_rgb)
print(f"{name} consistent vs checker at 1/{f}: {len(bad)} of {small.width * small.height} primary hits differ; "
      f"picture identical {same:.6f} max delta {mx}", flush=True)
for b in bad[:5]:
    print("   pixel", b, "device", hits[b], "checker", o_hits[b])

for mode in (False, True):
    ds.set_consistent(mode)
    ds.render(cam)
    t0 = time.perf_counter()
    for _ in range(frames):
        _, _, st = ds.render(cam)
    dt = (time.perf_counter() - t0) / frames * 1e3
    print(f"{name} {'consistent' if mode else 'exact'}: {dt:.3f} ms/frame, {st['rays'] / dt / 1e3:.1f} Mrays/s, rays {st['rays']}", flush=True)
