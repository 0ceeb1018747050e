"""Dev helper (GPU box): one frame rendered as K interleaved tile subsets on K frame contexts at once
(tile t in subset t mod K), against the frame rendered whole.  The tile buffers are pageable: their host-side
copies overlap the other subsets' kernels, which flatters the subsets by 3-4 % against the whole frame's one copy.
Usage: python tools/subframes.py <workload> [tile_w tile_h [half]]   (half: every subset on a grid of one wave per SIMD)"""
import ctypes as C
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon_metal"
tw = int(sys.argv[2]) if len(sys.argv) > 3 else 64
th = int(sys.argv[3]) if len(sys.argv) > 3 else 8
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0)
ds.set_generation_timing(False)
cam, _ = hs.camera(0)
lib = rt.amd_lib()
ds.render(cam)
t0 = time.perf_counter()
for _ in range(10):
    ds.render(cam)
print(f"{name}: whole frame {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
if len(sys.argv) > 4 and sys.argv[4] == "half":
    from raytracer_amd import _abi
    ds.set_debug(_abi.DEBUG_HALF_GRID)
    print("subsets on half grids", flush=True)
for k in (2, 3, 4, 6, 8):
    frames = [ds.frame() for _ in range(k)]
    bufs = [np.empty(lib.rtx_render_output_bytes(C.byref(cam), tw, th, i, k), dtype=np.uint8) for i in range(k)]

    def once():
        for i, fr in enumerate(frames):
            fr.submit_tiles(cam, tw, th, i, k, bufs[i].ctypes.data, False)
        for fr in frames:
            fr.wait()
    once()
    t0 = time.perf_counter()
    for _ in range(10):
        once()
    print(f"{name}: {k} subsets of {tw}x{th} tiles at once {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms (pageable tile buffers)", flush=True)
    for fr in frames:
        fr.close()
ds.close(); hs.close()
