"""Dev experiment (GPU box): how much does a frame gain when it is rendered as K interleaved
tile subsets concurrently (K scene contexts, K streams, K host threads)?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
import bench

d = bench.scene_cache_dir()
path, _ = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "dragon_metal", d)
hs = rt.HostScene(path)
cam, _ = hs.camera(0)
for K in (1, 2, 4, 8):
    scenes = [rt.DeviceScene(hs.desc, 0) for _ in range(K)]
    def work(p, out):
        _, st = scenes[p].render_tiles(cam, 8, 8, p, K)
        out[p] = st
    best = 1e9
    for rep in range(6):
        out = [None] * K
        th = [threading.Thread(target=work, args=(p, out)) for p in range(K)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = (time.perf_counter() - t0) * 1e3
        if rep >= 2: best = min(best, dt)
    print(f"K={K}: best wall {best:.2f} ms; per-context kernel ms:", " ".join(f"{o['kernel_ms']:.2f}" for o in out), flush=True)
    for s in scenes: s.close()
