"""integration/rtx_bridge.cpp -- the binding a maintainer of the reference adds to its build --
compiled against the reference's own headers and run on its own Scene objects
(oracle/_ref/bridge_check, built by `make -C oracle bridge`; skipped where neither the reference
sources nor a prebuilt binary are present)."""
import os
import subprocess

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt

BRIDGE = os.path.join(ck.REF_DIR, "bridge_check")
REFERENCE = "/root/reference"


def _run(args):
    return subprocess.run([BRIDGE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bridge_compiles_against_the_reference_headers():
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference sources not present")
    subprocess.check_call(["make", "-C", os.path.join(ck.ROOT, "oracle"), "bridge"], stdout=subprocess.DEVNULL)
    assert os.access(BRIDGE, os.X_OK)


@pytest.mark.parametrize("extra", [[], ["--cull", "1"], ["--bvh", "0"]])
def test_bridge_description_equals_the_host_librarys(unit_scenes, extra):
    """The description filled from the reference's own Scene equals, field by field, the one this
    repository's loader produces for the same file (BVH nodes included: the 72-byte records are
    passed as they stand)."""
    if not os.access(BRIDGE, os.X_OK):
        pytest.skip("bridge_check not built")
    for name, path in unit_scenes.items():
        r = _run([path] + extra)
        assert r.returncode == 0, f"{name} {extra}: {r.stdout[-300:]} {r.stderr[-1500:]}"
        assert "0 mismatches" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "prims", "marching_small"])
def test_bridge_renders_the_references_scene_on_the_device(unit_scenes, tmp_path, name):
    """Upload through the bridge, render through deviceRayTracing: the picture of the reference's
    own Scene object equals the oracle's to the parity bar."""
    if not os.access(BRIDGE, os.X_OK):
        pytest.skip("bridge_check not built")
    out = str(tmp_path / "bridge.rgb")
    r = _run([unit_scenes[name], "--render", out])
    assert r.returncode == 0 and "rendered" in r.stdout, r.stdout[-300:] + r.stderr[-1500:]
    hs = rt.HostScene(unit_scenes[name])
    cam, _ = hs.camera(0)
    rgb = ck.read_rgb(out, cam.width, cam.height)
    o_rgb, _, _, _ = ck.oracle_render(hs.desc, cam, want_hits=False)
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999, (same, mx)
    hs.close()
