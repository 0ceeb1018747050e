"""Dev helper (GPU box): a workload's scene view with and without the shading kernel settling the shadow
rays that no object can hold up (RTX_DEBUG_NO_EARLY_SHADOW): ms per frame, share of such rays, same bytes?
Usage: python tools/early_shadow.py <workload>..."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402

for name in sys.argv[1:] or ["dragon_metal"]:
    path, _ = bench.make_workload(name, bench.scene_cache_dir())
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    pics = {}
    for label, flags in (("settled early", 0), ("all queued", _abi.DEBUG_NO_EARLY_SHADOW), ("settled early", 0), ("all queued", _abi.DEBUG_NO_EARLY_SHADOW)):
        ds.set_debug(flags)
        for _ in range(2):
            ds.render(cam)
        n = 5 if name == "grass_desert" else 20
        t0 = time.perf_counter()
        for _ in range(n):
            rgb, _, st = ds.render(cam)
        ms = (time.perf_counter() - t0) / n * 1e3
        pics[label] = rgb.copy()
        print(f"{name}: {label}: {ms:.3f} ms/frame, kernels {st['kernel_ms']:.3f}, trace {st['trace_ms']:.3f}, rays {st['rays']}, shadow {st['shadow_rays']}, "
              f"settled early {st['shadow_rays_settled_early']} ({100.0 * st['shadow_rays_settled_early'] / max(st['shadow_rays'], 1):.1f} %), "
              f"trace launches {st['trace_launches']}, total_ms {st['total_ms']:.3f}", flush=True)
    print(f"{name}: same bytes: {np.array_equal(pics['settled early'], pics['all queued'])}", flush=True)
    ds.close(); hs.close()
