"""Dev helper (GPU box): exact mode's instance search against the plain replay in index order, same
library, same frame (the view reduced by `reduce_factor`): primary-hit records and pictures must be equal.
Usage: tools/search_check.py <workload> [reduce_factor]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "grass_desert"
f = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
small = bench.reduced_camera(cam, f) if f > 1 else cam

ds.set_debug(0)
a_rgb, a_hits, _ = ds.render(small, want_hits=True)
ds.set_debug(_abi.DEBUG_NO_SEARCH)
b_rgb, b_hits, _ = ds.render(small, want_hits=True)
bad = np.nonzero(a_hits != b_hits)[0]
px = np.nonzero((a_rgb != b_rgb).any(axis=2).ravel())[0]
print(f"{name} at 1/{f}: {len(bad)} of {small.width * small.height} primary-hit records differ, {len(px)} pixels differ", flush=True)
for i in bad[:12]:
    print("  pixel", int(i), "search", a_hits[i], "replay", b_hits[i])
only_px = [int(i) for i in px if i not in set(bad.tolist())][:12]
for i in only_px:
    y, x = divmod(i, small.width)
    print("  pixel", i, "same primary hit", a_hits[i], "colours", a_rgb[y, x], b_rgb[y, x])
