#!/usr/bin/env python3
"""Dev helper (GPU box): per-generation trace times of a workload's frame with the eight-wide BLAS, with
the four-wide records, and with the four-wide records without the wave-wide finishing of drained waves.
Usage: tools/b8_timing.py <workload>"""
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
ds = rt.DeviceScene(hs.desc, 0, wide_blas=True)
cam, _ = hs.camera(0)
for wide, flags, label in ((True, 0, "eight-wide"), (False, 0, "four-wide"), (True, _abi.DEBUG_NO_WIDE_OLD, "eight-wide, no wide finishing"),
                           (False, _abi.DEBUG_NO_WIDE_OLD, "four-wide, no wide finishing")):
    ds.set_wide_blas(wide)
    ds.set_debug(flags)
    ds.render(cam)
    t0 = time.perf_counter()
    for _ in range(5):
        _, _, st = ds.render(cam)
    print(f"==== {name}: {label}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/frame, trace {st['trace_ms']:.3f} ms, records {st['record_fetches']}",
          file=sys.stderr, flush=True)
    ds.set_debug(flags | _abi.DEBUG_GEN_TIMING)
    ds.render(cam)
    # throughput with three frames in flight (tails filled by the neighbouring frames)
    ds.set_debug(flags)
    ds.set_generation_timing(False)
    frames = [ds.frame() for _ in range(3)]
    def stream(n):
        q = []
        for i in range(n):
            fr = frames[i % 3]
            if len(q) == 3:
                q.pop(0).wait()
            fr.submit(cam)
            q.append(fr)
        for fr in q:
            fr.wait()
    stream(3)
    t0 = time.perf_counter()
    stream(30)
    print(f"==== {name}: {label}: three in flight {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/frame", file=sys.stderr, flush=True)
    for fr in frames:
        fr.close()
    ds.set_generation_timing(True)
