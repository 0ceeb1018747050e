"""Dev helper (GPU box): random scenes with detailed meshes, HIP core against the CPU checker in the
default and the consistent mode.  Usage: python tools/soak.py [first_seed] [count] [detail]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _checkers as ck          # noqa: E402
import raytracer_amd as rt      # noqa: E402
from raytracer_amd import scenes  # noqa: E402

first, count, detail = (int(x) for x in (sys.argv[1:] + ["0", "100", "12"])[:3])
bad = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(first, first + count):
        path = scenes.scene_random(d, seed, res=(288, 208), detail=detail)
        hs = rt.HostScene(path, True, bool(seed & 1))
        ds = rt.DeviceScene(hs.desc, 0)
        cam, _ = hs.camera(0)
        for consistent in (False, True):
            ds.set_consistent(consistent)
            o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam, consistent=consistent)
            rgb, hits, _ = ds.render(cam, want_hits=True)
            n_hits = len(ck.compare_hits(hits, o_hits, allow_src_unknown=False))
            same, mx = ck.image_diff(rgb, o_rgb)
            if n_hits or mx > 2 or same < 0.999:
                bad += 1
                print(f"seed {seed} consistent {consistent}: {n_hits} hit records differ, same {same:.5f}, max delta {mx}", flush=True)
        ds.close(); hs.close()
        os.remove(path)
        if seed % 20 == 0:
            print(f"seed {seed} done", flush=True)
print(f"{count} scenes, {bad} failures")
sys.exit(1 if bad else 0)
