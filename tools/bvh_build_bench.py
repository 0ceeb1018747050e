#!/usr/bin/env python3
"""Dev helper (GPU box): time rtx_build_bvh on the benchmark mesh against the host builder.
Usage: tools/bvh_build_bench.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "dragon_metal"
path, _ = bench.make_workload(w, bench.scene_cache_dir())
t = time.time(); host = rt.HostScene(path, use_bvh=True); t_host = time.time() - t
t = time.time(); dev = rt.HostScene(path, use_bvh=False); t_nobvh = time.time() - t
for rep in range(3):
    t = time.time(); ms = dev.build_bvh_on_device(0); wall = time.time() - t
    print(f"{w}: device build {ms:.1f} ms (python wall {1000 * wall:.1f} ms); host load with BVH {1000 * t_host:.0f} ms, "
          f"without {1000 * t_nobvh:.0f} ms, host preprocess {host.timings()['preprocess_ms']:.0f} ms", flush=True)
