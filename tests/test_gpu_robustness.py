"""Failure modes of the hot path on the GPU: a frame that cannot be rendered completely must
be redone or refused, never returned wrong.  Queue overflow with dirty queue memory, the
deepest BVH the reference's builder can produce (bvh.cpp:210: depth 64), a tree deeper than
the traversal stack, and the packed-tile contract of rtx_render."""
import ctypes as C
import os

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import _abi, scenes
from raytracer_amd.scenes import _camera, _dump, _flat, _light, _mat

pytestmark = pytest.mark.gpu


def test_overflowing_frame_with_dirty_queues(tmp_path):
    """The queues are sized by a small first frame; a larger one overflows them.  With the
    queue and hit-record memory filled with 0xFF beforehand, slots the overflowed attempt
    reserved but never wrote hold garbage: no kernel may consume them."""
    path = scenes.scene_tiny(str(tmp_path), res=(16, 12))
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    ds.set_debug(_abi.DEBUG_POISON_QUEUES)
    cam, _ = hs.camera(0)
    ds.render(cam)
    big = type(cam).from_buffer_copy(cam)
    big.width, big.height = 400, 300
    rgb, hits, st = ds.render(big, want_hits=True)
    assert st["queue_overflows"] >= 1
    fresh = rt.DeviceScene(hs.desc, 0)
    fresh.set_debug(0)
    rgb2, hits2, st2 = fresh.render(big, want_hits=True)
    assert st2["queue_overflows"] == 0
    assert np.array_equal(hits, hits2)
    same, mx = ck.image_diff(rgb, rgb2)
    assert mx == 0 and same == 1.0
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, big)
    assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999
    ds.close(); fresh.close(); hs.close()


def _chain_scene(tmp_path, n_faces, res=(48, 40)):
    """A staircase of n_faces triangles receding from the camera, as one mesh."""
    v, f = [], []
    for i in range(n_faces):
        x = -2.0 + 4.0 * i / max(n_faces - 1, 1)
        z = -0.15 * i
        y = 0.2 + 0.03 * i
        b = len(v) + 1
        v += [(x - 0.5, y - 0.4, z), (x + 0.5, y - 0.4, z), (x, y + 0.6, z)]
        f += [b, b + 1, b + 2]
    sc = {
        "BackgroundColor": "3 5 9", "MaxRecursionDepth": "2", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0.2, 1.2, 6.0), (0, 0.8, -3), (0, 1, 0), 50, res, "chain.png")},
        "Lights": {"AmbientLight": "30 30 30", "PointLight": [_light(1, (3, 6, 5), (9000, 9000, 9000))]},
        "Materials": {"Material": [_mat(1, (1, 1, 1), (0.6, 0.5, 0.4), (0.3, 0.3, 0.3), phong=10, mirror=(0.4, 0.4, 0.4))]},
        "VertexData": {"_data": _flat(np.array(v, dtype=np.float64))},
        "Objects": {"Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat",
                              "Faces": {"_data": " ".join(str(k) for k in f)}}]}}
    return _dump(str(tmp_path / f"chain_{n_faces}.json"), sc)


def _attach_chain(hs, n_faces):
    """A maximally unbalanced tree over the mesh: inner node k has leaf {face k} on the left and
    the rest on the right, so the last leaves sit at depth n_faces - 1."""
    d = hs.desc.contents
    m = d.meshes[0]
    assert m.n_faces == n_faces
    verts = np.ctypeslib.as_array(d.vertices, shape=(d.n_vertices, 3))
    faces = np.ctypeslib.as_array(m.faces, shape=(m.n_faces, 3))
    lo = verts[faces].min(axis=1)            # per-face boxes
    hi = verts[faces].max(axis=1)
    suffix_lo = np.minimum.accumulate(lo[::-1], axis=0)[::-1]
    suffix_hi = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
    nodes = (_abi.BvhNode * (2 * n_faces - 1))()
    k = 0
    for face in range(n_faces - 1):          # inner node for faces [face, n), then its left leaf
        inner = nodes[k]
        inner.bounds_min[:] = suffix_lo[face]; inner.bounds_max[:] = suffix_hi[face]
        inner.left, inner.right = k + 1, k + 2
        inner.first_face, inner.face_count, inner.split_axis = -1, 0, face % 3
        leaf = nodes[k + 1]
        leaf.bounds_min[:] = lo[face]; leaf.bounds_max[:] = hi[face]
        leaf.left = leaf.right = -1
        leaf.first_face, leaf.face_count, leaf.split_axis = face, 1, 0
        k += 2
    last = nodes[k]
    last.bounds_min[:] = lo[n_faces - 1]; last.bounds_max[:] = hi[n_faces - 1]
    last.left = last.right = -1
    last.first_face, last.face_count = n_faces - 1, 1
    order = (C.c_int32 * n_faces)(*range(n_faces))
    assert rt.host_lib().rtx_host_attach_bvh(hs._h, 0, nodes, 2 * n_faces - 1, order) == 0, \
        rt.host_lib().rtx_host_last_error()
    return nodes, order


def test_deepest_reference_tree_renders_exactly(tmp_path):
    """bvh.cpp:210 stops splitting at depth 64 and bvh.cpp:124 walks with a 64-entry stack: a
    chain of that depth is the worst tree the reference can hand over.  It must render exactly."""
    n = 62                                     # leaves down to depth 61; the reference's own stack still fits
    hs = rt.HostScene(_chain_scene(tmp_path, n), use_bvh=False)
    keep = _attach_chain(hs, n)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    for counters in (True, False):
        ds.set_counters(counters)
        rgb, hits, st = ds.render(cam, want_hits=True)
        o_rgb, o_hits, o_ctr, _ = ck.oracle_render(hs.desc, cam)
        assert (hits["kind"] == _abi.KIND_MESH).sum() > 50
        assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0
        same, mx = ck.image_diff(rgb, o_rgb)
        assert mx <= 2 and same >= 0.999, (same, mx)
        if counters:
            assert st["rays"] == o_ctr["rays"] and st["tri_tests"] == o_ctr["tri_tests"]
    del keep
    ds.close(); hs.close()


def test_tree_deeper_than_the_stack_is_refused(tmp_path):
    n = 80                                     # 3 * (80 / 2) = 120 possible stack entries > 100
    hs = rt.HostScene(_chain_scene(tmp_path, n), use_bvh=False)
    keep = _attach_chain(hs, n)
    with pytest.raises(rt.RtxError) as e:
        rt.DeviceScene(hs.desc, 0)
    assert "(-5)" in str(e.value) and "deeper" in str(e.value)
    del keep
    hs.close()


def test_every_tile_selected_is_still_packed(tmp_path):
    """tile_first = 0, tile_stride = 1 with tiles smaller than the image selects every tile and
    writes them packed: the buffer is sized by rtx_render_output_bytes (larger than w*h*3 when
    edge tiles are clipped) and rtx_unpack_tiles brings it into image order."""
    path = scenes.scene_tiny(str(tmp_path), res=(100, 100))
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    full, _, _ = ds.render(cam)
    n = ds.tile_bytes(cam, 64, 8, 0, 1)
    assert n == 2 * 13 * 64 * 8 * 3 and n > 100 * 100 * 3
    packed, _ = ds.render_tiles(cam, 64, 8, 0, 1)
    assert packed.size == n
    frame = np.zeros_like(full)
    rt.unpack_tiles(packed, cam.width, cam.height, 64, 8, 0, 1, frame)
    same, mx = ck.image_diff(frame, full)
    assert mx == 0 and same == 1.0
    # hit records need the one-tile form
    hits = np.empty(cam.width * cam.height, dtype=_abi.HIT_DTYPE)
    rc = rt.amd_lib().rtx_render(ds._h, C.byref(cam), 64, 8, 0, 1, packed.ctypes.data, 0, hits.ctypes.data, None)
    assert rc == -1
    ds.close(); hs.close()
