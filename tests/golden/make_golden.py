"""Regenerates tests/golden/*.npz from the compiled reference (oracle/_ref/ref_dump).

Run in the build container, where /root/reference exists:  python tests/golden/make_golden.py
For every unit scene (raytracer_amd.scenes.UNIT_SCENES, seeded generators) and every
camera it stores the reference's 8-bit image and its per-pixel primary-hit records.  The
fixtures are data only; the scenes themselves are regenerated from the same generators
wherever the tests run.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _checkers as ck  # noqa: E402
from raytracer_amd import HostScene, scenes  # noqa: E402


def main():
    assert ck.have_reference(), "build oracle/_ref first (make -C oracle ref)"
    with tempfile.TemporaryDirectory() as d:
        paths = scenes.write_unit_scenes(d)
        for name, path in paths.items():
            for cull in (False, True):
                if cull and name not in ("prims", "tiny"):
                    continue
                prefix = os.path.join(d, name + ("_cull" if cull else ""))
                ck.reference_dump(path, prefix, cull=cull)
                hs = HostScene(path, True, cull)
                out = {}
                for k in range(hs.n_cameras):
                    cam, _ = hs.camera(k)
                    out[f"rgb{k}"] = ck.read_rgb(f"{prefix}.cam{k}.rgb", cam.width, cam.height)
                    out[f"hits{k}"] = ck.read_hits(f"{prefix}.cam{k}.hits")
                np.savez_compressed(os.path.join(HERE, os.path.basename(prefix) + ".npz"), **out)
                print("wrote", os.path.basename(prefix), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
