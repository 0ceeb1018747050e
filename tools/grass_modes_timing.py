#!/usr/bin/env python3
"""Dev helper (GPU box): per-generation kernel times of grass_desert in the three ways its instances can be
taken -- exact mode's nearest-first search, the index-order replay (RTX_AMD_NO_SEARCH=1), consistent mode.
Usage: tools/grass_modes_timing.py   (the figures come on stderr, RTX_AMD_GEN_TIMING lines)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTX_AMD_GEN_TIMING"] = "1"
import bench, raytracer_amd as rt
path, _ = bench.make_workload("grass_desert", bench.scene_cache_dir())
hs = rt.HostScene(path); ds = rt.DeviceScene(hs.desc, 0); cam, _ = hs.camera(0)
for label, cons, nos in (("search", False, False), ("replay", False, True), ("consistent", True, False)):
    if nos: os.environ["RTX_AMD_NO_SEARCH"] = "1"
    else: os.environ.pop("RTX_AMD_NO_SEARCH", None)
    ds.set_consistent(cons)
    ds.render(cam)
    print("====", label, file=sys.stderr, flush=True)
    ds.render(cam)
