"""Dev helper (GPU box): frames per second of a workload's scene view with K = 1..6 frames in flight on one
uploaded scene (rtx_frame_create / rtx_render_async / rtx_frame_wait), generation timing off.
Usage: python tools/inflight_sweep.py <workload> [<workload> ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402

for name in sys.argv[1:] or ["dragon_metal"]:
    path, _ = bench.make_workload(name, bench.scene_cache_dir())
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    ds.set_generation_timing(False)
    cam, _ = hs.camera(0)
    ds.render(cam)
    for k in (1, 2, 3, 4, 6):
        frames = [ds.frame() for _ in range(k)]

        def stream(n):
            q = []
            for i in range(n):
                fr = frames[i % k]
                if len(q) == k:
                    q.pop(0).wait()
                fr.submit(cam)
                q.append(fr)
            for fr in q:
                fr.wait()
        stream(2 * k)
        n = 40
        t0 = time.perf_counter()
        stream(n)
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{name}: {k} in flight: {ms:.3f} ms/frame ({1e3 / ms:.0f} frames/s)", flush=True)
        for fr in frames:
            fr.close()
    ds.close(); hs.close()
