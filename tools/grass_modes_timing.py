"""Dev helper (GPU box): per-generation kernel times of grass_desert in the three ways its instances can be
taken -- exact mode's nearest-first search, the index-order replay (RTX_DEBUG_NO_SEARCH), consistent mode.
Usage: python tools/grass_modes_timing.py   (the figures come on stderr)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402

path, _ = bench.make_workload("grass_desert", bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
for label, cons, flags in (("search", False, 0), ("replay", False, _abi.DEBUG_NO_SEARCH), ("consistent", True, 0)):
    ds.set_consistent(cons)
    ds.set_debug(flags)
    ds.render(cam)
    print("====", label, file=sys.stderr, flush=True)
    ds.set_debug(flags | _abi.DEBUG_GEN_TIMING)
    _, _, st = ds.render(cam)
    print(f"{label}: kernels {st['kernel_ms']:.2f} ms, records {st['record_fetches']}", file=sys.stderr, flush=True)
ds.close(); hs.close()
