"""Several cameras of one uploaded scene in flight at once (rtx_frame_create / rtx_render_async /
rtx_frame_wait): the reference renders the cameras of a file one after the other
(raytracer.cpp:252-264) and the frames of an animation one process each (runner.py:105-141).
Every picture must be the bytes the blocking call gives, and the oracle's picture to the bar."""
import ctypes as C

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import _abi, scenes

pytestmark = pytest.mark.gpu


def _orbit(cam, k, n):
    """Camera k of n on a circle round the scene's vertical axis (same distance, same height)."""
    import math
    c = type(cam).from_buffer_copy(cam)
    a = 2.0 * math.pi * k / n
    x, z = cam.position[0], cam.position[2]
    c.position[0] = x * math.cos(a) + z * math.sin(a)
    c.position[2] = -x * math.sin(a) + z * math.cos(a)
    gx, gz = cam.gaze[0], cam.gaze[2]
    c.gaze[0] = gx * math.cos(a) + gz * math.sin(a)
    c.gaze[2] = -gx * math.sin(a) + gz * math.cos(a)
    return c


@pytest.mark.parametrize("name", ["tiny", "marching_small", "grass_small"])
def test_three_frames_in_flight(unit_scenes, name):
    hs = rt.HostScene(unit_scenes[name])
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    cams = [_orbit(cam, k, 7) for k in range(7)]
    blocking = [ds.render(c, want_hits=True) for c in cams]
    frames = [ds.frame() for _ in range(3)]
    got = [None] * len(cams)
    in_flight = []                                  # (camera number, frame)
    for i, c in enumerate(cams):
        fr = frames[i % 3]
        if len(in_flight) == 3:
            j, f = in_flight.pop(0)
            assert f is fr
            got[j] = f.wait()
        fr.submit(c, want_hits=True)
        in_flight.append((i, fr))
    for j, f in in_flight:
        got[j] = f.wait()
    for i, c in enumerate(cams):
        rgb, hits, st = got[i]
        b_rgb, b_hits, b_st = blocking[i]
        assert np.array_equal(rgb, b_rgb) and np.array_equal(hits, b_hits), (name, i)
        assert st["rays"] == b_st["rays"]
        o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, c)
        assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0, (name, i)
        same, mx = ck.image_diff(rgb, o_rgb)
        assert mx <= 2 and same >= 0.999, (name, i, same, mx)
    for f in frames:
        f.close()
    ds.close(); hs.close()


def test_a_frame_keeps_the_lights_it_was_submitted_with(unit_scenes):
    """rtx_scene_update_lights publishes a new version; a render in flight finishes with the old one."""
    hs = rt.HostScene(unit_scenes["tiny"])
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    big = type(cam).from_buffer_copy(cam)
    big.width, big.height = 640, 480                # (long enough to be in flight when the lights change)
    lights_a = hs.lights()
    lights_b = [type(lights_a[0]).from_buffer_copy(lights_a[0])]
    lights_b[0].position[0] = -lights_a[0].position[0]
    lights_b[0].intensity[1] *= 0.25
    with_a = ds.render(big)[0]
    ds.update_lights(lights_b)
    with_b = ds.render(big)[0]
    assert not np.array_equal(with_a, with_b)
    ds.update_lights(lights_a)
    f1, f2 = ds.frame(), ds.frame()
    f1.submit(big)
    ds.update_lights(lights_b)
    f2.submit(big)
    ds.update_lights(lights_a)
    a, _, _ = f1.wait()
    b, _, _ = f2.wait()
    assert np.array_equal(a, with_a) and np.array_equal(b, with_b)
    assert np.array_equal(ds.render(big)[0], with_a)
    ds.close(); hs.close()


def test_frame_contract(unit_scenes):
    """One render per frame at a time; a wait needs a render; pageable and page-locked output
    buffers give the same bytes; a frame outlives neither its render nor its scene."""
    hs = rt.HostScene(unit_scenes["tiny"])
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    ref = ds.render(cam)[0]
    fr = ds.frame()
    with pytest.raises(rt.RtxError):
        fr.wait()
    fr.submit(cam)
    with pytest.raises(rt.RtxError):
        fr.submit(cam)
    rgb, _, st = fr.wait()
    assert np.array_equal(rgb, ref) and st["rays"] > 0
    # a page-locked buffer (torch is only the allocator here)
    try:
        import torch
        pinned = torch.empty(ref.size, dtype=torch.uint8).pin_memory().numpy()
    except Exception:
        pinned = None
    if pinned is not None:
        fr.submit(cam, out=pinned)
        out, _, _ = fr.wait()
        assert np.array_equal(out.reshape(ref.shape), ref)
    # tiles through a frame: the packed bytes of the blocking call
    n = ds.tile_bytes(cam, 16, 8, 1, 3)
    want, _ = ds.render_tiles(cam, 16, 8, 1, 3)
    buf = np.zeros(n, dtype=np.uint8)
    fr.submit_tiles(cam, 16, 8, 1, 3, buf.ctypes.data, False)
    fr.wait()
    assert np.array_equal(buf, want)
    # a frame left alive goes with its scene
    ds.frame().submit(cam)
    ds.close(); hs.close()


def test_frame_contexts_from_several_host_threads(unit_scenes):
    """Four host threads, each with a frame context of the same scene, render orbit cameras in a loop
    while a fifth thread keeps publishing one of two light versions: every picture is the blocking call's
    bytes for one of the two versions (a frame renders with the version current at its submit), the
    blocking call itself may run beside them, and nothing is torn."""
    import threading
    hs = rt.HostScene(unit_scenes["marching_small"])
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    cams = [_orbit(cam, k, 5) for k in range(5)]
    lights_a = hs.lights()
    lights_b = [type(l).from_buffer_copy(l) for l in lights_a]
    lights_b[0].intensity[0] *= 0.5
    ds.update_lights(lights_b)
    want_b = [ds.render(c)[0].copy() for c in cams]
    ds.update_lights(lights_a)
    want_a = [ds.render(c)[0].copy() for c in cams]
    assert not np.array_equal(want_a[0], want_b[0])
    errors, seen = [], {"a": 0, "b": 0}
    stop = threading.Event()

    def renderer(t):
        try:
            fr = ds.frame()
            for i in range(24):
                k = (t + i) % len(cams)
                fr.submit(cams[k])
                rgb, _, _ = fr.wait()
                if np.array_equal(rgb, want_a[k]):
                    seen["a"] += 1
                elif np.array_equal(rgb, want_b[k]):
                    seen["b"] += 1
                else:
                    errors.append((t, i, "a picture of neither light version"))
            fr.close()
        except Exception as e:                       # noqa: BLE001
            errors.append((t, repr(e)))

    def lights():
        n = 0
        while not stop.is_set():
            ds.update_lights(lights_b if n % 2 == 0 else lights_a)
            n += 1

    threads = [threading.Thread(target=renderer, args=(t,)) for t in range(4)]
    lt = threading.Thread(target=lights)
    lt.start()
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    stop.set()
    lt.join(timeout=30)
    assert not errors, errors[:3]
    assert not any(th.is_alive() for th in threads) and not lt.is_alive()
    assert seen["a"] + seen["b"] == 4 * 24
    ds.update_lights(lights_a)
    assert np.array_equal(ds.render(cams[0])[0], want_a[0])
    ds.close(); hs.close()


def test_stacked_untransformed_instances(tmp_path):
    """64 untransformed instances of 64 small meshes behind one another, nearer ones with higher
    indices, all of them on every central ray: in the default mode (nearest-first search, from 32
    instances on), under the index-order replay, and against the oracle."""
    n = 64
    verts, meshes, insts = [], [], []
    for k in range(n):
        z = -0.05 * (n - 1 - k)                      # mesh k lies in front of mesh k - 1
        b = len(verts) + 1
        s = 1.0 + 0.01 * k
        verts += [(-s, -s, z), (s, -s, z), (s, s, z), (-s, s, z)]
        meshes.append({"_id": str(k + 1), "Material": str(1 + k % 2), "_shadingMode": "flat",
                       "Faces": {"_data": f"{b} {b + 1} {b + 2} {b} {b + 2} {b + 3}"}})
        insts.append({"_id": str(n + k + 1), "_baseMeshId": str(k + 1), "_resetTransform": "true", "Material": str(2 - k % 2)})
    sc = {
        "BackgroundColor": "7 7 7", "MaxRecursionDepth": "1", "ShadowRayEpsilon": "1e-3", "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": scenes._camera((0.3, 0.2, 6.0), (0, 0, -1), (0, 1, 0), 35, (80, 60), "stacked.png")},
        "Lights": {"AmbientLight": "30 30 30", "PointLight": [scenes._light(1, (2, 3, 8), (3000, 3000, 3000))]},
        "Materials": {"Material": [scenes._mat(1, (1, 0.4, 0.2), (0.5, 0.5, 0.5), (0.1, 0.1, 0.1), phong=5),
                                   scenes._mat(2, (0.2, 0.4, 1), (0.3, 0.6, 0.3), (0.2, 0.2, 0.2), phong=20)]},
        "VertexData": {"_data": scenes._flat(np.array(verts, dtype=np.float64))},
        "Objects": {"Mesh": meshes, "MeshInstance": insts}}
    path = scenes._dump(str(tmp_path / "stacked.json"), sc)
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    rgb, hits, st = ds.render(cam, want_hits=True)
    assert st["instance_path"] == 1
    ds.set_debug(_abi.DEBUG_NO_SEARCH)
    r_rgb, r_hits, r_st = ds.render(cam, want_hits=True)
    assert r_st["instance_path"] == 0
    assert np.array_equal(hits, r_hits) and np.array_equal(rgb, r_rgb)
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam)
    assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999, (same, mx)
    ds.close(); hs.close()
