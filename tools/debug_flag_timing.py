"""Dev helper (GPU box): frame time of a workload's own view with one rtx_scene_set_debug switch set, against none.
Usage: python tools/debug_flag_timing.py <workload> <flag name without DEBUG_, e.g. NO_SCATTER> [frames]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402

name, flag = sys.argv[1], getattr(_abi, "DEBUG_" + sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
ref = ds.render(cam)[0].copy()
for label, f in (("none", 0), (sys.argv[2], flag), ("none", 0), (sys.argv[2], flag)):
    ds.set_debug(f)
    rgb, _, st = ds.render(cam)
    assert np.array_equal(rgb, ref)
    t0 = time.perf_counter()
    for _ in range(n):
        _, _, st = ds.render(cam)
    print(f"{name}: {label}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame, trace {st['trace_ms']:.3f} ms, "
          f"records {st['record_fetches']}", flush=True)
ds.close(); hs.close()
