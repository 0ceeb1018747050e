"""Edge cases of the hot path on the GPU, each checked against the CPU oracle: empty and
degenerate inputs, image sizes that do not fill a wave or an 8x8 block, recursion depth 0,
meshes without faces and instances of them, scenes without lights, many tiny tiles."""
import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.scenes import _camera, _dump, _flat, _light, _mat

pytestmark = pytest.mark.gpu

GROUND = [(-6, 0, 6), (6, 0, 6), (6, 0, -6), (-6, 0, -6)]
PYRAMID = [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1.6, 0)]


def base_scene(res, depth=3, lights=True, name="edge.png"):
    return {
        "BackgroundColor": "7 11 13", "MaxRecursionDepth": str(depth), "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0.3, 2.5, 7), (0, 0.6, 0), (0, 1, 0), 40, res, name)},
        "Lights": {"AmbientLight": "30 30 30",
                   **({"PointLight": [_light(1, (4, 6, 5), (9000, 9000, 9000)), _light(2, (-5, 3, 2), (4000, 3000, 2000))]}
                      if lights else {})},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.6, 0.6, 0.6), (0.2, 0.2, 0.2), phong=8, mirror=(0.5, 0.5, 0.5)),
            _mat(2, (1, 0.5, 0.5), (0.7, 0.3, 0.3), (0.4, 0.4, 0.4), phong=20)]},
        "VertexData": {"_data": _flat(np.array(GROUND + PYRAMID, dtype=np.float64))},
    }


def check(path, cull=False):
    hs = rt.HostScene(path, True, cull)
    ds = rt.DeviceScene(hs.desc, 0)
    out = []
    for counters in (True, False):
        ds.set_counters(counters)
        for k in range(hs.n_cameras):
            cam, _ = hs.camera(k)
            rgb, hits, st = ds.render(cam, want_hits=True)
            o_rgb, o_hits, o_ctr, _ = ck.oracle_render(hs.desc, cam)
            assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0
            same, mx = ck.image_diff(rgb, o_rgb)
            assert mx <= 2 and same >= 0.999, (same, mx)
            if counters:
                assert st["rays"] == o_ctr["rays"] and st["tri_tests"] == o_ctr["tri_tests"]
            out.append((rgb, hits))
    ds.close(); hs.close()
    return out


def test_scene_without_objects(tmp_path):
    sc = base_scene((7, 5))
    path = _dump(str(tmp_path / "empty.json"), sc)
    (rgb, hits), _ = check(path)
    assert np.all(rgb.reshape(-1, 3) == np.array([7, 11, 13], dtype=np.uint8)) and np.all(hits["kind"] == 0)


@pytest.mark.parametrize("res", [(1, 1), (13, 7), (65, 3), (8, 8), (9, 64)])
def test_image_sizes_off_the_block_grid(tmp_path, res):
    sc = base_scene(res)
    sc["Objects"] = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": "1 2 3 1 3 4"}},
                              {"_id": "2", "Material": "2", "Faces": {"_data": "5 6 9 6 7 9 7 8 9 8 5 9"}}]}
    check(_dump(str(tmp_path / "odd.json"), sc))


def test_recursion_depth_zero_and_no_lights(tmp_path):
    sc = base_scene((40, 30), depth=0)
    sc["Objects"] = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": "1 2 3 1 3 4"}},
                              {"_id": "2", "Material": "2", "Faces": {"_data": "5 6 9 6 7 9 7 8 9 8 5 9"}}]}
    check(_dump(str(tmp_path / "d0.json"), sc))
    sc2 = base_scene((40, 30), depth=2, lights=False)
    sc2["Objects"] = sc["Objects"]
    check(_dump(str(tmp_path / "nolights.json"), sc2))


def test_meshes_without_faces_and_instances_of_them(tmp_path):
    sc = base_scene((48, 36))
    sc["Transformations"] = {"Translation": [{"_id": "1", "_data": "2.5 0 0"}, {"_id": "2", "_data": "-2.5 0 -1"}],
                             "Scaling": [{"_id": "1", "_data": "0.5 2 0.5"}]}
    sc["Objects"] = {
        "Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": "1 2 3 1 3 4"}},
                 {"_id": "2", "Material": "2", "Faces": {"_data": ""}},                       # no faces at all
                 {"_id": "3", "Material": "2", "Faces": {"_data": "5 6 9 6 7 9 7 8 9 8 5 9"}}],
        "MeshInstance": [{"_id": "4", "_baseMeshId": "2", "Material": "1", "Transformations": "t1"},        # of the empty mesh
                         {"_id": "5", "_baseMeshId": "3", "Material": "1", "Transformations": "s1 t1"},
                         {"_id": "6", "_baseMeshId": "77", "Transformations": "t2"},                        # unresolved base
                         {"_id": "7", "_baseMeshId": "5", "_resetTransform": "false", "Transformations": "t2"}]}
    check(_dump(str(tmp_path / "emptymesh.json"), sc))
    check(_dump(str(tmp_path / "emptymesh.json"), sc), cull=True)


def test_many_tiny_tiles(tmp_path):
    path = scenes.scene_tiny(str(tmp_path), res=(37, 29))
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    full, _, _ = ds.render(cam)
    for tile, world in (((1, 1), 3), ((5, 3), 7), ((64, 64), 2), ((37, 1), 29)):
        frame = np.zeros_like(full)
        for rank in range(world):
            packed, _ = ds.render_tiles(cam, tile[0], tile[1], rank, world)
            if packed.size:
                rt.unpack_tiles(packed, cam.width, cam.height, tile[0], tile[1], rank, world, frame)
        same, mx = ck.image_diff(frame, full)
        assert mx == 0 and same == 1.0, (tile, world, same, mx)
    ds.close(); hs.close()


_ASSEMBLE_SCRIPT = r"""
import sys
import numpy as np
import torch
torch.cuda.init()                      # torch's HIP runtime first: the library then shares it (as in bench.py)
import raytracer_amd as rt
hs = rt.HostScene(sys.argv[1])
ds = rt.DeviceScene(hs.desc, 0)
cam, _ = hs.camera(0)
full, _, _ = ds.render(cam)
for tile, world in (((1, 1), 3), ((5, 3), 7), ((64, 64), 2), ((37, 1), 40), ((16, 8), 4)):
    parts = []
    for rank in range(world):
        n = ds.tile_bytes(cam, tile[0], tile[1], rank, world)
        buf = torch.full((max(n, 1),), 0xAB, dtype=torch.uint8, device="cuda:0")
        if n:
            ds.render_tiles(cam, tile[0], tile[1], rank, world, out_ptr=buf.data_ptr(), out_is_device=True)
        parts.append((buf, n))
    torch.cuda.synchronize()
    frame = np.full_like(full, 0xCD)
    rt.assemble_tiles(0, [b.data_ptr() if n else 0 for b, n in parts], cam.width, cam.height, tile[0], tile[1], frame)
    assert np.array_equal(frame, full), (tile, world)
try:
    rt.assemble_tiles(0, [0], cam.width, cam.height, 8, 8, np.zeros_like(full))       # a rank with tiles but no buffer
    raise SystemExit("a missing buffer was accepted")
except rt.RtxError:
    pass
host = np.zeros(full.size, dtype=np.uint8)                                            # host memory where device memory is due
try:
    rt.assemble_tiles(0, [host.ctypes.data], cam.width, cam.height, cam.width, cam.height, np.zeros_like(full))
    raise SystemExit("a host pointer was accepted as a tile buffer")
except rt.RtxError as e:
    assert "not device memory" in str(e), str(e)
print("ASSEMBLE-OK")
"""


def test_tiles_assembled_on_the_device(tmp_path, torch_warm):
    """rtx_assemble_tiles: the packed tile buffers of all ranks, in device memory as an RCCL gather
    leaves them on rank 0, scattered into the frame by one kernel -- the same bytes as the whole
    frame, with clipped edge tiles and with more ranks than tiles; a buffer that is missing, or
    is not device memory of that GPU as the HIP runtime knows it, is refused with RTX_EINVAL.  In a
    process of its own, in the order bench.py has (torch first, then the library: both then use
    the HIP runtime torch brought along)."""
    import os
    import subprocess
    import sys
    path = scenes.scene_tiny(str(tmp_path), res=(37, 29))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", _ASSEMBLE_SCRIPT, path], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "ASSEMBLE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_generations_launched_follow_the_previous_frame(tmp_path):
    """A frame launches one ray generation more than the scene's previous frame had rays for and
    launches the rest only when the next queue turns out to hold rays: a frame that looks at
    nothing (camera rays only) followed by the full view must give the full view's bytes and
    hit records, as a fresh scene does and as the RTX_DEBUG_ALL_GENERATIONS switch (every
    generation launched) does."""
    from raytracer_amd import _abi
    hs = rt.HostScene(scenes.scene_tiny(str(tmp_path)))
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    ref, ref_hits, st = ds.render(cam, want_hits=True)
    assert st["rays"] > cam.width * cam.height          # (the view spawns shadow and secondary rays)
    away = type(cam).from_buffer_copy(cam)
    for k in range(3):
        away.gaze[k] = -cam.gaze[k]
    _, _, st_away = ds.render(away)
    again, again_hits, st2 = ds.render(cam, want_hits=True)      # the previous frame needed fewer generations
    assert np.array_equal(again, ref) and np.array_equal(again_hits, ref_hits) and st2["rays"] == st["rays"]
    third, _, st3 = ds.render(cam)                               # ... and this one's predecessor all of them
    assert np.array_equal(third, ref) and st3["rays"] == st["rays"]
    ds.set_debug(_abi.DEBUG_ALL_GENERATIONS)
    full, _, st4 = ds.render(cam)
    ds.set_debug(0)
    assert np.array_equal(full, ref) and st4["rays"] == st["rays"]
    ds.close(); hs.close()


def test_generation_timing_off(tmp_path):
    """rtx_scene_set_generation_timing(0): the same frame, no per-generation trace time."""
    hs = rt.HostScene(scenes.scene_tiny(str(tmp_path)))
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    a, _, st_a = ds.render(cam)
    ds.set_generation_timing(False)
    b, _, st_b = ds.render(cam)
    ds.set_generation_timing(True)
    c, _, st_c = ds.render(cam)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert st_a["trace_ms"] > 0 and st_b["trace_ms"] == 0 and st_c["trace_ms"] > 0
    assert st_b["kernel_ms"] > 0 and st_b["rays"] == st_a["rays"]
    ds.close(); hs.close()


def test_queues_regrow_when_a_larger_frame_follows(tmp_path):
    """The ray queues are sized by the first frame; a much larger one overflows them, the
    library grows them and renders the frame again -- with the same result as a fresh scene."""
    path = scenes.scene_tiny(str(tmp_path), res=(16, 12))
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    ds.render(cam)
    big = type(cam).from_buffer_copy(cam)
    big.width, big.height = 400, 300
    rgb, hits, st = ds.render(big, want_hits=True)
    assert st["queue_overflows"] >= 1
    fresh = rt.DeviceScene(hs.desc, 0)
    rgb2, hits2, st2 = fresh.render(big, want_hits=True)
    assert st2["queue_overflows"] == 0
    assert np.array_equal(hits, hits2)
    same, mx = ck.image_diff(rgb, rgb2)
    assert mx == 0 and same == 1.0
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, big)
    assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999
    again, _, st3 = ds.render(big)                       # the grown queues are kept
    assert st3["queue_overflows"] == 0 and ck.image_diff(again, rgb)[1] <= 1
    ds.close(); fresh.close(); hs.close()
