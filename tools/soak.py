"""Dev helper (GPU box): random scenes with detailed meshes, HIP core against the CPU checker in the
default and the consistent mode, over the four-wide and (where the scene qualifies) the eight-wide records.
Usage: python tools/soak.py [first_seed] [count] [detail]"""
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
bad = used = redone = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(first, first + count):
        path = scenes.scene_random(d, seed, res=(288, 208), detail=detail)
        hs = rt.HostScene(path, True, bool(seed & 1))
        ds = rt.DeviceScene(hs.desc, 0, wide_blas=True)
        cam, _ = hs.camera(0)
        for consistent in (False, True):
            ds.set_consistent(consistent)
            o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam, consistent=consistent)
            for wide in (False, True):
                ds.set_wide_blas(wide)
                rgb, hits, st = ds.render(cam, want_hits=True)
                used += st["wide_blas"]; redone += st["wide_blas_redone"]
                n_hits = len(ck.compare_hits(hits, o_hits, allow_src_unknown=False))
                same, mx = ck.image_diff(rgb, o_rgb)
                if n_hits or mx > 2 or same < 0.999:
                    bad += 1
                    print(f"seed {seed} consistent {consistent} wide {wide}: {n_hits} hit records differ, same {same:.5f}, max delta {mx}", flush=True)
        ds.close(); hs.close()
        os.remove(path)
        if seed % 20 == 0:
            print(f"seed {seed} done", flush=True)
print(f"{count} scenes, {bad} failures; frames over the eight-wide records {used}, redone {redone}")
sys.exit(1 if bad else 0)
