"""The device-side BVH build (rtx_build_bvh) against the host builder, which is itself
bit-identical to the reference's MeshBVH::build (tests/test_host_pipeline.py).

The level-synchronous build must give the reference's tree: same boxes, topology, split axes,
pre-order numbering and leaf face sets.  Only the order of faces inside a leaf may differ (the
reference has whatever std::nth_element left, the device has them sorted), which can matter
only for hits that tie exactly within a leaf."""
import ctypes as C

import numpy as np
import pytest

import _checkers as ck
import bench
import raytracer_amd as rt
from raytracer_amd import _abi

pytestmark = pytest.mark.gpu

NODE_DTYPE = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("left", "<i4"), ("right", "<i4"),
                       ("first_face", "<i4"), ("face_count", "<i4"), ("split_axis", "<i4"), ("pad", "<i4")])


def mesh_arrays(desc, i):
    m = desc.contents.meshes[i]
    if m.n_nodes == 0:
        return np.zeros(0, NODE_DTYPE), np.zeros(0, np.int32)
    assert C.sizeof(_abi.BvhNode) == NODE_DTYPE.itemsize == 72
    nodes = np.frombuffer(C.string_at(m.nodes, m.n_nodes * 72), dtype=NODE_DTYPE)
    order = np.frombuffer(C.string_at(m.face_indices, m.n_faces * 4), dtype=np.int32)
    return nodes, order


def assert_same_tree(a_nodes, a_order, b_nodes, b_order, what):
    assert len(a_nodes) == len(b_nodes), f"{what}: {len(a_nodes)} vs {len(b_nodes)} nodes"
    for f in ("left", "right", "first_face", "face_count", "split_axis"):
        assert np.array_equal(a_nodes[f], b_nodes[f]), f"{what}: node field {f} differs"
    assert np.array_equal(a_nodes["bmin"], b_nodes["bmin"]) and np.array_equal(a_nodes["bmax"], b_nodes["bmax"]), \
        f"{what}: node boxes differ"
    assert np.array_equal(np.sort(a_order), np.sort(b_order))
    # leaf by leaf the same faces: sort inside every leaf range and compare whole arrays
    leaves = a_nodes[a_nodes["left"] < 0]
    a_sorted, b_sorted = a_order.copy(), b_order.copy()
    for first, count in zip(leaves["first_face"], leaves["face_count"]):
        a_sorted[first:first + count].sort()
        b_sorted[first:first + count].sort()
    assert np.array_equal(a_sorted, b_sorted), f"{what}: leaf face sets differ"


def assert_valid_tree(desc, i, what):
    """Structure of a reference-style tree over mesh i, whatever way medians with tied centroid
    coordinates were resolved: pre-order numbering, leaves that tile the face order, a
    permutation of the faces, leaf boxes = box of their faces, inner boxes = union of children."""
    d = desc.contents
    m = d.meshes[i]
    nodes, order = mesh_arrays(desc, i)
    assert np.array_equal(np.sort(order), np.arange(m.n_faces))
    verts = np.ctypeslib.as_array(d.vertices, shape=(d.n_vertices, 3))
    faces = np.ctypeslib.as_array(m.faces, shape=(m.n_faces, 3))
    inner = nodes["left"] >= 0
    idx = np.arange(len(nodes))
    assert np.all(nodes["left"][inner] == idx[inner] + 1) and np.all(nodes["right"][inner] > nodes["left"][inner])
    leaves = nodes[~inner]
    assert np.array_equal(np.cumsum(leaves["face_count"]) - leaves["face_count"], leaves["first_face"])
    assert leaves["face_count"].sum() == m.n_faces
    for n in leaves[:: max(1, len(leaves) // 2000)]:
        pts = verts[faces[order[n["first_face"]:n["first_face"] + n["face_count"]]].reshape(-1)]
        assert np.array_equal(pts.min(axis=0), n["bmin"]) and np.array_equal(pts.max(axis=0), n["bmax"]), what
    l, r = nodes["left"][inner], nodes["right"][inner]
    assert np.array_equal(np.minimum(nodes["bmin"][l], nodes["bmin"][r]), nodes["bmin"][inner]), what
    assert np.array_equal(np.maximum(nodes["bmax"][l], nodes["bmax"][r]), nodes["bmax"][inner]), what
    # a leaf holds at most four faces unless its centroids coincide (bvh.cpp:205,229)
    assert leaves["face_count"].max() <= 4 or what.startswith("degenerate")


# Meshes whose face centroids tie exactly at a median (the grass clump: blades with equal
# heights) can be partitioned either way; there std::nth_element's choice is not reproducible.
TIE_FREE = {"tiny", "bumpy_small", "david_small", "marching_small"}


@pytest.mark.parametrize("name", ["tiny", "bumpy_small", "grass_small", "david_small", "marching_small"])
def test_unit_scene_trees_equal_the_host_builder(unit_scenes, name):
    host = rt.HostScene(unit_scenes[name], use_bvh=True)
    dev = rt.HostScene(unit_scenes[name], use_bvh=False)
    dev.build_bvh_on_device(0)
    n = host.desc.contents.n_meshes
    assert n == dev.desc.contents.n_meshes
    for i in range(n):
        assert_valid_tree(dev.desc, i, f"{name} mesh {i}")
        if name in TIE_FREE:
            assert_same_tree(*mesh_arrays(host.desc, i), *mesh_arrays(dev.desc, i), f"{name} mesh {i}")
    # and the pictures: rendered from the device-built trees, checked against the oracle on the host-built ones
    ds = rt.DeviceScene(dev.desc, 0)
    for k in range(dev.n_cameras):
        cam, _ = dev.camera(k)
        rgb, hits, _ = ds.render(cam, want_hits=True)
        o_rgb, o_hits, _, _ = ck.oracle_render(host.desc, cam)
        bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
        assert len(bad) == 0, f"{name}: {len(bad)} primary hits differ"
        same, mx = ck.image_diff(rgb, o_rgb)
        assert mx <= 2 and same >= 0.999
    ds.close(); host.close(); dev.close()


@pytest.mark.parametrize("workload", ["dragon_metal", "camera_around_david", "grass_desert"])
def test_full_size_trees_equal_the_host_builder(workload):
    path, _ = bench.make_workload(workload, bench.scene_cache_dir())
    host = rt.HostScene(path, use_bvh=True)
    dev = rt.HostScene(path, use_bvh=False)
    ms = dev.build_bvh_on_device(0)
    for i in range(host.desc.contents.n_meshes):
        assert_valid_tree(dev.desc, i, f"{workload} mesh {i}")
        a_nodes, a_order = mesh_arrays(host.desc, i)
        b_nodes, b_order = mesh_arrays(dev.desc, i)
        # the shape of the tree follows from the face counts alone
        for f in ("left", "right", "first_face", "face_count"):
            assert np.array_equal(a_nodes[f], b_nodes[f]), f"{workload} mesh {i}: node field {f} differs"
        differ = np.any(a_nodes["bmin"] != b_nodes["bmin"], axis=1) | np.any(a_nodes["bmax"] != b_nodes["bmax"], axis=1) | \
            (a_nodes["split_axis"] != b_nodes["split_axis"])
        print(f"{workload} mesh {i}: {int(differ.sum())} of {len(a_nodes)} nodes differ from the host builder's")
        if workload in TIE_FREE:
            assert_same_tree(a_nodes, a_order, b_nodes, b_order, f"{workload} mesh {i}")
        elif len(a_nodes) > 1000:
            # tied centroid coordinates at a median: a few ranges are cut differently
            # (1.8 M-face mesh: 3 nodes of a million; the 30-face grass clump, all symmetry, is
            # cut differently in most of its 15 nodes and is covered by the validity check alone)
            assert differ.mean() < 1e-4
    print(f"{workload}: device build {ms:.1f} ms, host build+precompute {host.timings()['preprocess_ms']:.1f} ms")
    host.close(); dev.close()


def test_bad_arguments_are_rejected():
    lib = rt.amd_lib()
    n_nodes = C.c_int32(0)
    verts = (C.c_double * 9)(0, 0, 0, 1, 0, 0, 0, 1, 0)
    faces = (C.c_int32 * 3)(0, 1, 5)                      # vertex 5 does not exist
    nodes = (_abi.BvhNode * 2)()
    order = (C.c_int32 * 1)()
    assert lib.rtx_build_bvh(0, verts, 3, faces, 1, nodes, C.byref(n_nodes), order, None) != 0
    assert b"vertex" in lib.rtx_last_error()
    faces = (C.c_int32 * 3)(0, 1, 2)
    assert lib.rtx_build_bvh(0, verts, 3, faces, 1, nodes, C.byref(n_nodes), order, None) == 0
    assert n_nodes.value == 1 and nodes[0].left == -1 and nodes[0].face_count == 1 and order[0] == 0
    assert lib.rtx_build_bvh(99, verts, 3, faces, 1, nodes, C.byref(n_nodes), order, None) != 0
