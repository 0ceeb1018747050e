"""Procedural scene generators in the reference's own input format.

The reference renders JSON scene files whose every scalar is a *string*
(scene.cpp:25-59) and whose meshes may live in PLY files next to the scene
(scene.cpp:500-505, 699-733).  The scenes behind the reference's published
timings are not distributed with it, so the parity tests and the benchmark use
the seeded stand-ins generated here: same file format, same object mix
(resolution, materials, instancing pattern) as the published renders.

Every generator returns the path of the written ``.json`` file.  ``Mesh``
entries always carry ``_shadingMode`` because the reference leaves the field
uninitialised when the key is absent (scene.cpp:572-575).
"""
from __future__ import annotations

import json
import math
import os
import struct

import numpy as np

__all__ = [
    "write_ply", "bumpy_sphere", "scene_tiny", "scene_q1", "scene_q2", "scene_prims",
    "scene_plates", "scene_dragon_metal", "scene_marching", "scene_david_frame",
    "scene_grass", "UNIT_SCENES", "write_unit_scenes",
]


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def _s(*vals) -> str:
    return " ".join(repr(float(v)) if isinstance(v, (float, np.floating)) else str(v) for v in vals)


def _mat(mid, ka, kd, ks, phong=None, mirror=None, mtype=None, n=None, k=None, sigma=None):
    m = {"_id": str(mid), "AmbientReflectance": _s(*ka), "DiffuseReflectance": _s(*kd),
         "SpecularReflectance": _s(*ks)}
    if phong is not None:
        m["PhongExponent"] = _s(phong)
    if mirror is not None:
        m["MirrorReflectance"] = _s(*mirror)
    if mtype is not None:
        m["_type"] = mtype
    if n is not None:
        m["RefractionIndex"] = _s(n)
    if k is not None:
        m["AbsorptionIndex"] = _s(k)
    if sigma is not None:
        m["AbsorptionCoefficient"] = _s(*sigma)
    return m


def _camera(pos, gaze_point, up, fovy, res, name, near=1.0, cid=1):
    return {"_id": str(cid), "_type": "lookAt", "Position": _s(*pos), "GazePoint": _s(*gaze_point),
            "Up": _s(*up), "FovY": _s(fovy), "NearDistance": _s(near),
            "ImageResolution": f"{res[0]} {res[1]}", "ImageName": name}


def _light(lid, pos, inten):
    return {"_id": str(lid), "Position": _s(*pos), "Intensity": _s(*inten)}


def _flat(a) -> str:
    a = np.asarray(a)
    if a.dtype.kind == "f":
        return " ".join(repr(float(x)) for x in a.reshape(-1))
    return " ".join(str(int(x)) for x in a.reshape(-1))


def _dump(path, scene):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"Scene": scene}, f, indent=1)
    return path


def write_ply(path, verts, faces, fmt="binary_little_endian"):
    """Write a triangle mesh as PLY (float x,y,z + ``list uchar int vertex_indices``)."""
    verts = np.asarray(verts, dtype=np.float32)
    faces = np.asarray(faces, dtype=np.int32)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    hdr = (f"ply\nformat {fmt} 1.0\nelement vertex {len(verts)}\n"
           "property float x\nproperty float y\nproperty float z\n"
           f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if fmt == "ascii":
            for v in verts:
                f.write(("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2]))).encode())
            for t in faces:
                f.write(("3 %d %d %d\n" % (t[0], t[1], t[2])).encode())
        else:
            big = fmt == "binary_big_endian"
            f.write(verts.astype(">f4" if big else "<f4").tobytes())
            rec = np.zeros(len(faces), dtype=[("n", "u1"), ("i", (">i4" if big else "<i4"), 3)])
            rec["n"] = 3
            rec["i"] = faces
            f.write(rec.tobytes())
    return path


def bumpy_sphere(nu, nv, radius=1.0, bump=0.08, seed=795, center=(0.0, 0.0, 0.0)):
    """Lat-long sphere displaced by a seeded sum of sines: 2*nu*(nv-1) triangles."""
    rng = np.random.default_rng(seed)
    u = np.arange(nu) * (2.0 * np.pi / nu)
    v = (np.arange(1, nv)) * (np.pi / nv)              # interior rings
    uu, vv = np.meshgrid(u, v)                           # [nv-1][nu]
    r = np.full_like(uu, radius)
    for _ in range(6):
        fu, fv = rng.integers(2, 13, size=2)
        ph = rng.uniform(0, 2 * np.pi, size=2)
        r = r + (bump / 6.0) * radius * np.sin(fu * uu + ph[0]) * np.sin(fv * vv + ph[1]) * 2.0
    x = r * np.sin(vv) * np.cos(uu) + center[0]
    y = r * np.cos(vv) + center[1]
    z = r * np.sin(vv) * np.sin(uu) + center[2]
    ring = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    top = np.array([[center[0], center[1] + radius, center[2]]])
    bot = np.array([[center[0], center[1] - radius, center[2]]])
    verts = np.concatenate([ring, top, bot], axis=0)
    itop, ibot = len(ring), len(ring) + 1
    faces = []
    j = np.arange(nu)
    jn = (j + 1) % nu
    faces.append(np.stack([np.full(nu, itop), jn, j], axis=1))            # top cap
    for i in range(nv - 2):
        a = i * nu + j
        b = i * nu + jn
        c = (i + 1) * nu + j
        d = (i + 1) * nu + jn
        faces.append(np.stack([a, b, d], axis=1))
        faces.append(np.stack([a, d, c], axis=1))
    last = (nv - 2) * nu
    faces.append(np.stack([np.full(nu, ibot), last + j, last + jn], axis=1))  # bottom cap
    return verts.astype(np.float32), np.concatenate(faces, axis=0).astype(np.int32)


def _box_tris(base):
    """12 triangles (1-based vertex ids starting at ``base``) of an 8-vertex box."""
    q = [(0, 1, 2, 3), (4, 7, 6, 5), (0, 4, 5, 1), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 4, 0)]
    out = []
    for a, b, c, d in q:
        out += [(base + a, base + b, base + c), (base + a, base + c, base + d)]
    return out


def _box_verts(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    return [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0),
            (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]


# ----------------------------------------------------------------------------
# unit scenes (tiny, exercise every branch of the hot path)
# ----------------------------------------------------------------------------
def scene_tiny(out_dir, res=(64, 48)):
    """Plane-like ground mesh, transformed smooth conductor mesh, negatively scaled
    instance (dips below the ground), dielectric sphere."""
    sc = {
        "BackgroundColor": "10 20 30", "MaxRecursionDepth": "6", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0, 3, 10), (0, 0, 0), (0, 1, 0), 45, res, "tiny.png")},
        "Lights": {"AmbientLight": "25 25 25",
                   "PointLight": [_light(1, (5, 8, 5), (20000, 20000, 20000))]},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.6, 0.6, 0.6), (0.2, 0.2, 0.2), phong=10),
            _mat(2, (0, 0, 0), (0, 0, 0), (0, 0, 0), phong=50, mirror=(0.9, 0.8, 0.7),
                 mtype="conductor", n=2.5, k=3.1),
            _mat(3, (0, 0, 0), (0, 0, 0), (0, 0, 0), mtype="dielectric", n=1.5,
                 sigma=(0.01, 0.02, 0.01))]},
        "Transformations": {
            "Translation": [{"_id": "1", "_data": "2 1 0"}, {"_id": "2", "_data": "-3 0.5 1"}],
            "Scaling": {"_id": "1", "_data": "1.5 -1.5 1.5"},
            "Rotation": {"_id": "1", "_data": "30 0 1 0"}},
        "VertexData": {"_data": "-50 0 50  50 0 50  50 0 -50  -50 0 -50  0 1 0  -1 0 -1  1 0 -1  "
                                "1 0 1  -1 0 1  0 2 0"},
        "Objects": {
            "Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat",
                      "Faces": {"_data": "1 2 3 1 3 4"}},
                     {"_id": "2", "Material": "2", "_shadingMode": "smooth", "Transformations": "t1",
                      "Faces": {"_data": "6 7 10 7 8 10 8 9 10 9 6 10"}}],
            "MeshInstance": [{"_id": "3", "_baseMeshId": "2", "_resetTransform": "true",
                              "Transformations": "s1 r1 t2"}],
            "Sphere": [{"_id": "4", "Material": "3", "Center": "5", "Radius": "1"}]}}
    return _dump(os.path.join(out_dir, "tiny.json"), sc)


def _pyramid_scene(name, scale, ty, out_dir, res=(32, 32)):
    """Top-down camera over a 2-triangle ground mesh with one scaled pyramid instance.
    ``scale`` < 1 exercises the object-space-t vs world-space-t comparison of
    utils.cpp:483-489,510; a pyramid mostly under the ground at ``scale`` > 1 exercises the
    rejected-hit record of utils.cpp:567-570."""
    sc = {
        "BackgroundColor": "0 0 0", "MaxRecursionDepth": "0", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0, 10, 0), (0, 0, 0), (0, 0, -1), 40, res, name + ".png")},
        "Lights": {"AmbientLight": "100 100 100"},
        "Materials": {"Material": [
            _mat(1, (1, 0.5, 0.25), (0, 0, 0), (0, 0, 0)),
            _mat(2, (0.25, 0.5, 2.0), (0, 0, 0), (0, 0, 0))]},
        "Transformations": {"Scaling": {"_id": "1", "_data": _s(scale, scale, scale)},
                            "Translation": [{"_id": "1", "_data": _s(0, ty, 0)},
                                            {"_id": "2", "_data": "100 0 0"}]},
        "VertexData": {"_data": "-20 0 20  20 0 20  20 0 -20  -20 0 -20  "
                                "-1 0 -1  1 0 -1  1 0 1  -1 0 1  0 2 0"},
        "Objects": {
            "Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat",
                      "Faces": {"_data": "1 2 3 1 3 4"}},
                     {"_id": "2", "Material": "2", "_shadingMode": "flat", "Transformations": "t2",
                      "Faces": {"_data": "5 6 9 6 7 9 7 8 9 8 5 9 5 7 6 5 8 7"}}],
            "MeshInstance": [{"_id": "3", "_baseMeshId": "2", "_resetTransform": "true",
                              "Material": "2", "Transformations": "s1 t1"}]}}
    return _dump(os.path.join(out_dir, name + ".json"), sc)


def scene_q1(out_dir, which="shrunk"):
    scale = {"shrunk": 0.25, "unit": 1.0, "magnified": 2.0}[which]
    return _pyramid_scene(which, scale, 0.5, out_dir)


def scene_q2(out_dir, which="buried"):
    scale = {"buried": 2.0, "buried_unit": 1.0}[which]
    return _pyramid_scene(which, scale, -1.9 * scale, out_dir)


def scene_prims(out_dir, res=(96, 64), cull=False):
    """Every primitive class with and without object transforms, every material type,
    two lights, a chained instance and an untransformed instance."""
    verts = [(-30, 0, 30), (30, 0, 30), (30, 0, -30), (-30, 0, -30),      # 1-4 ground quad
             (0, 1.2, 0),                                                    # 5 sphere centre
             (-3, 0.01, 2), (-1, 0.01, 2), (-2, 2.0, 2),                     # 6-8 free triangle
             (0, 0, 0),                                                      # 9 origin
             (0, 0, -8)]                                                     # 10 back plane point
    verts += _box_verts((2.0, 0.001, -1.0), (3.0, 1.5, 0.0))                # 11-18 glass box
    verts += [(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1.6, 0)]   # 19-23 pyramid
    box = _box_tris(11)
    pyr = [(19, 20, 23), (20, 21, 23), (21, 22, 23), (22, 19, 23), (19, 21, 20), (19, 22, 21)]
    sc = {
        "BackgroundColor": "30 40 60", "MaxRecursionDepth": "5", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": [
            _camera((1.5, 4.0, 11.0), (0.5, 0.8, 0), (0, 1, 0), 40, res, "prims.png"),
            {"_id": "2", "Position": "-6 3 9", "Gaze": "0.55 -0.2 -0.8", "Up": "0 1 0",
             "NearPlane": "-0.6 0.6 -0.4 0.4", "NearDistance": "1",
             "ImageResolution": f"{res[0]} {res[1]}", "ImageName": "prims_b.png",
             "Transformations": "t5"}]},
        "Lights": {"AmbientLight": "20 20 20",
                   "PointLight": [_light(1, (6, 9, 6), (22000, 22000, 22000)),
                                  {**_light(2, (-7, 6, 3), (9000, 12000, 15000)),
                                   "Transformations": "t5"}]},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.55, 0.5, 0.45), (0.3, 0.3, 0.3), phong=20),
            _mat(2, (0.2, 0.2, 0.2), (0.1, 0.1, 0.1), (0.5, 0.5, 0.5), phong=100,
                 mirror=(0.8, 0.8, 0.8), mtype="mirror"),
            _mat(3, (0, 0, 0), (0, 0, 0), (0, 0, 0), phong=60, mirror=(0.95, 0.7, 0.3),
                 mtype="conductor", n=0.37, k=2.82),
            _mat(4, (0, 0, 0), (0, 0, 0), (0, 0, 0), mtype="dielectric", n=1.5,
                 sigma=(0.05, 0.1, 0.02)),
            _mat(5, (1, 1, 1), (0.1, 0.6, 0.2), (0.4, 0.4, 0.4), phong=8),
            _mat(6, (1, 1, 1), (0.7, 0.2, 0.2), (0.1, 0.1, 0.1), phong=3,
                 mirror=(0.25, 0.25, 0.25))]},
        "Transformations": {
            "Translation": [{"_id": "1", "_data": "-2.5 0 -2"}, {"_id": "2", "_data": "3.5 0.9 2.5"},
                            {"_id": "3", "_data": "0 0 -1"}, {"_id": "4", "_data": "-4.5 0 0.5"},
                            {"_id": "5", "_data": "0.25 0.1 0"}, {"_id": "6", "_data": "0 0.4 3.5"}],
            "Scaling": [{"_id": "1", "_data": "0.6 1.3 0.6"}, {"_id": "2", "_data": "1.4 0.5 1.0"},
                        {"_id": "3", "_data": "-1 1 1"}, {"_id": "4", "_data": "0.5 0.5 0.5"}],
            "Rotation": [{"_id": "1", "_data": "35 0 1 0"}, {"_id": "2", "_data": "-20 1 0 1"}],
            "Composite": {"_id": "1", "_data": "1 0 0 0.5  0 1 0 0  0 0 1 0.25  0 0 0 1"}},
        "VertexData": {"_data": _flat(np.array(verts, dtype=np.float64))},
        "Objects": {
            "Mesh": [
                {"_id": "1", "Material": "1", "_shadingMode": "flat", "Faces": {"_data": "1 2 3 1 3 4"}},
                {"_id": "2", "Material": "4", "_shadingMode": "flat", "Faces": {"_data": _flat(np.array(box))}},
                {"_id": "3", "Material": "5", "_shadingMode": "smooth", "Transformations": "r1 t1",
                 "Faces": {"_data": _flat(np.array(pyr))}}],
            "MeshInstance": [
                {"_id": "4", "_baseMeshId": "3", "_resetTransform": "false", "Material": "3",
                 "Transformations": "t6"},
                {"_id": "5", "_baseMeshId": "4", "_resetTransform": "false", "Transformations": "s3 t3"},
                {"_id": "6", "_baseMeshId": "3", "_resetTransform": "true", "Material": "6",
                 "Transformations": "s4 r2 t4"},
                {"_id": "7", "_baseMeshId": "2", "_resetTransform": "true", "Material": "2"}],
            "Triangle": [
                {"_id": "8", "Material": "6", "Indices": "6 7 8"},
                {"_id": "9", "Material": "5", "Indices": "6 7 8", "Transformations": "c1 t3"}],
            "Sphere": [
                {"_id": "10", "Material": "4", "Center": "5", "Radius": "1.2"},
                {"_id": "11", "Material": "2", "Center": "9", "Radius": "1", "Transformations": "s2 t2"},
                {"_id": "12", "Material": "3", "Center": "9", "Radius": "0.7", "Transformations": "s1 r2 t4"}],
            "Plane": [
                {"_id": "13", "Material": "1", "Point": "10", "Normal": "0 0 1"},
                {"_id": "14", "Material": "6", "Point": "9", "Normal": "0 1 0",
                 "Transformations": "r2 t1"}]}}
    return _dump(os.path.join(out_dir, "prims.json"), sc)


def scene_plates(out_dir, res=(800, 800), name="metal_glass_plates"):
    """Stand-in for metal_glass_plates: ground plane, Phong/mirror spheres, thin
    dielectric boxes made of free triangles, one conductor plate, two lights."""
    verts = [(0, 0, 0), (0, 1.0, -2.0), (-2.2, 0.8, -1.0), (2.3, 0.7, -0.5)]   # 1-4
    tris = []
    mats = []
    rng = np.random.default_rng(795)
    for i in range(4):                                                         # glass plates
        x0 = -2.4 + 1.3 * i
        base = len(verts) + 1
        verts += _box_verts((x0, 0.001, 1.0 - 0.35 * i), (x0 + 1.0, 1.6, 1.12 - 0.35 * i))
        tris += _box_tris(base)
        mats += [3] * 12
    base = len(verts) + 1                                                      # metal plate
    verts += _box_verts((-1.5, 0.001, -3.2), (1.5, 2.4, -3.1))
    tris += _box_tris(base)
    mats += [4] * 12
    sc = {
        "BackgroundColor": "0 0 0", "MaxRecursionDepth": "6", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0.3, 2.6, 7.5), (0, 0.8, 0), (0, 1, 0), 38, res, name + ".png")},
        "Lights": {"AmbientLight": "25 25 25",
                   "PointLight": [_light(1, (4, 7, 6), (18000, 18000, 18000)),
                                  _light(2, (-5, 5, 2), (9000, 9000, 11000))]},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.5, 0.5, 0.5), (0.2, 0.2, 0.2), phong=10),
            _mat(2, (1, 1, 1), (0.7, 0.15, 0.15), (0.6, 0.6, 0.6), phong=60),
            _mat(3, (0, 0, 0), (0, 0, 0), (0, 0, 0), mtype="dielectric", n=1.5, sigma=(0.1, 0.2, 0.1)),
            _mat(4, (0, 0, 0), (0, 0, 0), (0, 0, 0), phong=80, mirror=(0.9, 0.85, 0.6),
                 mtype="conductor", n=0.2, k=3.6),
            _mat(5, (0.1, 0.1, 0.1), (0.05, 0.05, 0.05), (0.8, 0.8, 0.8), phong=200,
                 mirror=(0.9, 0.9, 0.9), mtype="mirror")]},
        "VertexData": {"_data": _flat(np.array(verts, dtype=np.float64))},
        "Objects": {
            "Plane": [{"_id": "1", "Material": "1", "Point": "1", "Normal": "0 1 0"}],
            "Sphere": [{"_id": "2", "Material": "2", "Center": "2", "Radius": "1.0"},
                       {"_id": "3", "Material": "5", "Center": "3", "Radius": "0.8"},
                       {"_id": "4", "Material": "3", "Center": "4", "Radius": "0.7"}],
            "Triangle": [{"_id": str(10 + i), "Material": str(mats[i]),
                          "Indices": f"{t[0]} {t[1]} {t[2]}"} for i, t in enumerate(tris)]}}
    del rng
    return _dump(os.path.join(out_dir, name + ".json"), sc)


def _ground(y=0.0, half=60.0):
    return [(-half, y, half), (half, y, half), (half, y, -half), (-half, y, -half)]


def scene_dragon_metal(out_dir, faces=1_800_000, res=(1920, 1152), name="dragon_metal", depth=6):
    """Stand-in for dragon_metal: one large smooth conductor PLY mesh, one Lambertian
    MeshInstance of it (``_resetTransform``), a 2-triangle ground mesh, 2 lights."""
    nv = max(4, int(round(math.sqrt(faces / 4.0))))
    nu = 2 * nv
    v, f = bumpy_sphere(nu, nv, radius=1.5, bump=0.18, seed=795, center=(0.0, 1.75, 0.0))
    ply = f"{name}_{len(f)}.ply"
    write_ply(os.path.join(out_dir, ply), v, f)
    sc = {
        "BackgroundColor": "0 0 0", "MaxRecursionDepth": str(depth), "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0.8, 3.2, 8.5), (1.2, 1.5, 0), (0, 1, 0), 45, res, name + ".png")},
        "Lights": {"AmbientLight": "25 25 25",
                   "PointLight": [_light(1, (6, 10, 7), (35000, 35000, 35000)),
                                  _light(2, (-7, 6, 4), (14000, 14000, 16000))]},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.5, 0.5, 0.5), (0.1, 0.1, 0.1), phong=5),
            _mat(2, (0, 0, 0), (0, 0, 0), (0, 0, 0), phong=80, mirror=(0.95, 0.8, 0.45),
                 mtype="conductor", n=2.5, k=3.1),
            _mat(3, (1, 1, 1), (0.45, 0.2, 0.6), (0.05, 0.05, 0.05), phong=3)]},
        "Transformations": {"Translation": [{"_id": "1", "_data": "3.6 0 -0.5"}],
                            "Rotation": [{"_id": "1", "_data": "40 0 1 0"}]},
        "VertexData": {"_data": _flat(np.array(_ground(), dtype=np.float64))},
        "Objects": {
            "Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat", "Faces": {"_data": "1 2 3 1 3 4"}},
                     {"_id": "2", "Material": "2", "_shadingMode": "smooth", "Faces": {"_plyFile": ply}}],
            "MeshInstance": [{"_id": "3", "_baseMeshId": "2", "_resetTransform": "true",
                              "Material": "3", "Transformations": "r1 t1"}]}}
    return _dump(os.path.join(out_dir, name + ".json"), sc)


def scene_marching(out_dir, faces=400_000, res=(1920, 1080), name="marching_dragons"):
    """Stand-in for marching_dragons: one smooth base PLY mesh + 8 instances on a 3x3 grid:
    translations, a rotation, uniform scales below and above 1, one negative scale."""
    nv = max(4, int(round(math.sqrt(faces / 4.0))))
    v, f = bumpy_sphere(2 * nv, nv, radius=1.0, bump=0.2, seed=11, center=(0.0, 1.2, 0.0))
    ply = f"{name}_{len(f)}.ply"
    write_ply(os.path.join(out_dir, ply), v, f)
    mats = [_mat(1, (1, 1, 1), (0.5, 0.5, 0.5), (0.1, 0.1, 0.1), phong=5)]
    rng = np.random.default_rng(4)
    for i in range(9):
        kd = rng.uniform(0.15, 0.8, size=3)
        mats.append(_mat(2 + i, (1, 1, 1), kd, (0.4, 0.4, 0.4), phong=40))
    trans, insts = [], []
    k = 0
    scalings = [{"_id": "1", "_data": "0.7 0.7 0.7"}, {"_id": "2", "_data": "1.3 1.3 1.3"},
                {"_id": "3", "_data": "-1 1 1"}, {"_id": "4", "_data": "1.15 0.85 1.0"}]
    rots = [{"_id": "1", "_data": "45 0 1 0"}, {"_id": "2", "_data": "-30 0 1 0"}]
    extra = {1: "s1", 2: "s2", 3: "s3", 4: "r1", 5: "s4 r2", 6: "r2", 7: "s2 r1", 8: "s1"}
    for gz in range(3):
        for gx in range(3):
            if gx == 1 and gz == 1:
                continue                                   # the base mesh sits in the middle
            k += 1
            trans.append({"_id": str(k), "_data": _s(3.2 * (gx - 1), 0.0, 3.2 * (gz - 1))})
            insts.append({"_id": str(10 + k), "_baseMeshId": "2", "_resetTransform": "true",
                          "Material": str(2 + k), "Transformations": f"{extra[k]} t{k}"})
    sc = {
        "BackgroundColor": "0 0 0", "MaxRecursionDepth": "6", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0, 7.5, 12.5), (0, 0.8, 0), (0, 1, 0), 40, res, name + ".png")},
        "Lights": {"AmbientLight": "25 25 25",
                   "PointLight": [_light(1, (8, 12, 8), (55000, 55000, 55000)),
                                  _light(2, (-9, 8, 2), (20000, 20000, 24000))]},
        "Materials": {"Material": mats},
        "Transformations": {"Translation": trans, "Scaling": scalings, "Rotation": rots},
        "VertexData": {"_data": _flat(np.array(_ground(), dtype=np.float64))},
        "Objects": {
            "Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat", "Faces": {"_data": "1 2 3 1 3 4"}},
                     {"_id": "2", "Material": "2", "_shadingMode": "smooth", "Faces": {"_plyFile": ply}}],
            "MeshInstance": insts}}
    return _dump(os.path.join(out_dir, name + ".json"), sc)


def scene_david_frame(out_dir, frame, n_frames=360, faces=100_000, res=(800, 800),
                      name="davids_camera", write_mesh=True):
    """One frame of the camera_around_david stand-in: the camera orbits a ~100k-face
    smooth mesh in 1-degree steps; every frame shares the same PLY file."""
    nv = max(4, int(round(math.sqrt(faces / 4.0))))
    ply = f"{name}_mesh.ply"
    if write_mesh and not os.path.exists(os.path.join(out_dir, ply)):
        v, f = bumpy_sphere(2 * nv, nv, radius=1.0, bump=0.25, seed=21, center=(0.0, 1.3, 0.0))
        v[:, 1] = (v[:, 1] - 1.3) * 1.5 + 1.8                  # a taller, statue-like blob
        write_ply(os.path.join(out_dir, ply), v, f)
    ang = 2.0 * math.pi * frame / n_frames
    pos = (7.0 * math.sin(ang), 2.6, 7.0 * math.cos(ang))
    sc = {
        "BackgroundColor": "0 0 0", "MaxRecursionDepth": "6", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera(pos, (0, 1.6, 0), (0, 1, 0), 40, res, f"{name}_{frame:03d}.png")},
        "Lights": {"AmbientLight": "25 25 25",
                   "PointLight": [_light(1, (5, 9, 6), (30000, 30000, 30000)),
                                  _light(2, (-6, 5, -4), (14000, 14000, 14000))]},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.5, 0.5, 0.5), (0.1, 0.1, 0.1), phong=5, mirror=(0.3, 0.3, 0.3)),
            _mat(2, (1, 1, 1), (0.75, 0.72, 0.65), (0.3, 0.3, 0.3), phong=30)]},
        "VertexData": {"_data": _flat(np.array(_ground(), dtype=np.float64))},
        "Objects": {
            "Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat", "Faces": {"_data": "1 2 3 1 3 4"}},
                     {"_id": "2", "Material": "2", "_shadingMode": "smooth", "Faces": {"_plyFile": ply}}]}}
    return _dump(os.path.join(out_dir, f"{name}_{frame:03d}.json"), sc)


def scene_grass(out_dir, n_instances=10_000, res=(3840, 2160), name="grass_desert", seed=795):
    """Stand-in for grass_desert: a tiny blade-clump base mesh, thousands of randomly
    placed/rotated/scaled MeshInstances (scales >= 1 so that the reference's
    object-space-t comparison does not erase blades), sand ground mesh, one low light."""
    rng = np.random.default_rng(seed)
    # base clump: 5 bent blades, 6 triangles each
    bv, bf = [], []
    for b in range(5):
        a = rng.uniform(0, 2 * np.pi)
        lean = rng.uniform(0.1, 0.35)
        wdt = 0.035
        ca, sa = math.cos(a), math.sin(a)
        base = len(bv)
        for s in range(4):
            h = 0.33 * s
            off = lean * (h ** 2)
            w = wdt * (1.0 - s / 3.6)
            cx, cz = 0.08 * math.cos(a * 1.7 + b) + off * ca, 0.08 * math.sin(a * 1.7 + b) + off * sa
            bv.append((cx - w * sa, h, cz + w * ca))
            bv.append((cx + w * sa, h, cz - w * ca))
        for s in range(3):
            i0 = base + 2 * s
            bf.append((i0, i0 + 1, i0 + 3))
            bf.append((i0, i0 + 3, i0 + 2))
    ply = f"{name}_blade.ply"
    write_ply(os.path.join(out_dir, ply), np.array(bv), np.array(bf))
    trans, rots, scals, insts = [], [], [], []
    xs = rng.uniform(-18, 18, size=n_instances)
    zs = rng.uniform(-30, 6, size=n_instances)
    angs = rng.uniform(0, 360, size=n_instances)
    ss = rng.uniform(1.0, 2.2, size=n_instances)
    for i in range(n_instances):
        k = i + 1
        trans.append({"_id": str(k), "_data": _s(float(xs[i]), 0.0, float(zs[i]))})
        rots.append({"_id": str(k), "_data": _s(float(angs[i]), 0, 1, 0)})
        sy = float(ss[i] * rng.uniform(0.8, 1.4))
        scals.append({"_id": str(k), "_data": _s(float(ss[i]), sy, float(ss[i]))})
        insts.append({"_id": str(100 + k), "_baseMeshId": "2", "_resetTransform": "true",
                      "Material": "2" if i % 3 else "3", "Transformations": f"s{k} r{k} t{k}"})
    sc = {
        "BackgroundColor": "140 170 220", "MaxRecursionDepth": "2", "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": _camera((0, 2.2, 11.0), (0, 0.6, 0), (0, 1, 0), 42, res, name + ".png")},
        "Lights": {"AmbientLight": "40 40 40",
                   "PointLight": [_light(1, (25, 7, 10), (220000, 200000, 160000))]},
        "Materials": {"Material": [
            _mat(1, (1, 1, 1), (0.76, 0.6, 0.36), (0.05, 0.05, 0.05), phong=4),
            _mat(2, (1, 1, 1), (0.2, 0.55, 0.12), (0.15, 0.2, 0.1), phong=12),
            _mat(3, (1, 1, 1), (0.35, 0.6, 0.15), (0.15, 0.2, 0.1), phong=12)]},
        "Transformations": {"Translation": trans, "Rotation": rots, "Scaling": scals},
        "VertexData": {"_data": _flat(np.array(_ground(half=80.0), dtype=np.float64))},
        "Objects": {
            "Mesh": [{"_id": "1", "Material": "1", "_shadingMode": "flat", "Faces": {"_data": "1 2 3 1 3 4"}},
                     {"_id": "2", "Material": "2", "_shadingMode": "flat",
                      "Transformations": "t1",
                      "Faces": {"_plyFile": ply}}],
            "MeshInstance": insts}}
    return _dump(os.path.join(out_dir, name + ".json"), sc)


# name -> callable(out_dir) for the small scenes the parity tests iterate over
UNIT_SCENES = {
    "tiny": scene_tiny,
    "shrunk": lambda d: scene_q1(d, "shrunk"),
    "unit": lambda d: scene_q1(d, "unit"),
    "magnified": lambda d: scene_q1(d, "magnified"),
    "buried": lambda d: scene_q2(d, "buried"),
    "buried_unit": lambda d: scene_q2(d, "buried_unit"),
    "prims": scene_prims,
    "plates_small": lambda d: scene_plates(d, res=(80, 80), name="plates_small"),
    "bumpy_small": lambda d: scene_dragon_metal(d, faces=3000, res=(96, 64), name="bumpy_small"),
    "marching_small": lambda d: scene_marching(d, faces=2000, res=(96, 54), name="marching_small"),
    "david_small": lambda d: scene_david_frame(d, 17, faces=1500, res=(64, 64), name="david_small"),
    "grass_small": lambda d: scene_grass(d, n_instances=300, res=(96, 54), name="grass_small"),
    # more than 1024 instances: the device's instance bitmap needs more than one summary word
    "grass_many": lambda d: scene_grass(d, n_instances=1500, res=(64, 36), name="grass_many", seed=4242),
}


def scene_random(out_dir, seed, res=(56, 40), detail=1, crowd=0):
    """Seeded random scene for fuzzing the hot path against its checkers: a ground quad, 1-4
    small bumpy meshes (flat or smooth, some with transforms of their own), 0-30 instances
    (of meshes and of other instances, with and without ``_resetTransform``, scales below and
    above one, mirrored ones), up to two spheres, sometimes a plane and a free triangle, 1-3
    lights, every material type, recursion depth 0-4.  ``detail`` multiplies the meshes' resolution
    in both directions (deep trees, long traversals).  ``crowd`` appends that many more instances
    (drawn after everything else, so the scene without them is unchanged): random transforms as
    above, and among them exact copies of earlier instances (hits at equal distances), instances
    without any transform (copies of their base), and chains of instances of instances."""
    rng = np.random.default_rng(10_000 + seed)
    name = f"random_{seed:03d}" + (f"_x{detail}" if detail != 1 else "") + (f"_c{crowd}" if crowd else "")
    verts = [(-25.0, 0.0, 25.0), (25.0, 0.0, 25.0), (25.0, 0.0, -25.0), (-25.0, 0.0, -25.0)]
    mats = [_mat(1, (1, 1, 1), (0.55, 0.5, 0.45), (0.3, 0.3, 0.3), phong=12),
            _mat(2, (0.2, 0.2, 0.2), (0.1, 0.1, 0.1), (0.5, 0.5, 0.5), phong=80, mirror=(0.8, 0.8, 0.8), mtype="mirror"),
            _mat(3, (0, 0, 0), (0, 0, 0), (0, 0, 0), phong=60, mirror=(0.95, 0.7, 0.3), mtype="conductor", n=0.37, k=2.82),
            _mat(4, (0, 0, 0), (0, 0, 0), (0, 0, 0), mtype="dielectric", n=1.5, sigma=(0.05, 0.1, 0.02)),
            _mat(5, (1, 1, 1), (0.1, 0.6, 0.2), (0.4, 0.4, 0.4), phong=8),
            _mat(6, (1, 1, 1), (0.7, 0.2, 0.2), (0.1, 0.1, 0.1), phong=3, mirror=(0.25, 0.25, 0.25))]
    trans, scal, rots = [], [], []

    def new_transform():
        """A random chain 'sK rK tK' (parts optional) -> its reference string."""
        parts = []
        if rng.random() < 0.7:
            k = rng.choice([0.3, 0.6, 1.0, 1.5, 2.2])
            sx, sy, sz = (k, k, k) if rng.random() < 0.6 else tuple(k * rng.uniform(0.6, 1.4, size=3))
            if rng.random() < 0.2:
                sx = -sx
            scal.append({"_id": str(len(scal) + 1), "_data": _s(float(sx), float(sy), float(sz))})
            parts.append(f"s{len(scal)}")
        if rng.random() < 0.6:
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            rots.append({"_id": str(len(rots) + 1), "_data": _s(float(rng.uniform(-180, 180)), *map(float, ax))})
            parts.append(f"r{len(rots)}")
        trans.append({"_id": str(len(trans) + 1),
                      "_data": _s(float(rng.uniform(-6, 6)), float(rng.uniform(-0.5, 2.5)), float(rng.uniform(-6, 4)))})
        parts.append(f"t{len(trans)}")
        return " ".join(parts)

    meshes = [{"_id": "1", "Material": "1", "_shadingMode": "flat", "Faces": {"_data": "1 2 3 1 3 4"}}]
    next_id = 2
    n_base = int(rng.integers(1, 5))
    for b in range(n_base):
        v, f = bumpy_sphere(detail * int(rng.integers(6, 14)), detail * int(rng.integers(4, 9)), radius=float(rng.uniform(0.5, 1.3)),
                            bump=0.25, seed=int(rng.integers(1 << 30)),
                            center=(float(rng.uniform(-4, 4)), float(rng.uniform(0.6, 1.6)), float(rng.uniform(-4, 3))))
        base = len(verts)
        verts += [tuple(map(float, x)) for x in v]
        m = {"_id": str(next_id), "Material": str(int(rng.integers(1, 7))),
             "_shadingMode": "smooth" if rng.random() < 0.5 else "flat",
             "Faces": {"_data": _flat(f + base + 1)}}
        if rng.random() < 0.4:
            m["Transformations"] = new_transform()
        meshes.append(m)
        next_id += 1
    base_ids = [int(m["_id"]) for m in meshes[1:]]
    insts = []
    for _ in range(int(rng.choice([0, 1, 3, 4, 5, 9, 14, 30]))):
        inst = {"_id": str(next_id), "_baseMeshId": str(int(rng.choice(base_ids))),
                "_resetTransform": "true" if rng.random() < 0.6 else "false", "Transformations": new_transform()}
        if rng.random() < 0.7:
            inst["Material"] = str(int(rng.integers(1, 7)))
        insts.append(inst)
        if rng.random() < 0.3:
            base_ids.append(next_id)                       # later instances may instance this one
        next_id += 1
    objects = {"Mesh": meshes}
    if insts:
        objects["MeshInstance"] = insts
    spheres = []
    for _ in range(int(rng.integers(0, 3))):
        verts.append((float(rng.uniform(-4, 4)), float(rng.uniform(0.5, 2)), float(rng.uniform(-3, 4))))
        sp = {"_id": str(next_id), "Material": str(int(rng.integers(1, 7))), "Center": str(len(verts)),
              "Radius": _s(float(rng.uniform(0.4, 1.1)))}
        if rng.random() < 0.4:
            sp["Transformations"] = new_transform()
        spheres.append(sp)
        next_id += 1
    if spheres:
        objects["Sphere"] = spheres
    if rng.random() < 0.3:
        verts.append((0.0, 0.0, -9.0))
        objects["Plane"] = [{"_id": str(next_id), "Material": "1", "Point": str(len(verts)), "Normal": "0 0 1"}]
        next_id += 1
    if rng.random() < 0.4:
        verts += [(-3.0, 0.01, 2.5), (-1.0, 0.01, 2.5), (-2.0, 2.0, 2.0)]
        tri = {"_id": str(next_id), "Material": str(int(rng.integers(1, 7))),
               "Indices": f"{len(verts) - 2} {len(verts) - 1} {len(verts)}"}
        if rng.random() < 0.5:
            tri["Transformations"] = new_transform()
        objects["Triangle"] = [tri]
        next_id += 1
    lights = [_light(k + 1, (float(rng.uniform(-9, 9)), float(rng.uniform(5, 11)), float(rng.uniform(-4, 9))),
                     tuple(float(x) for x in rng.uniform(8000, 26000, size=3))) for k in range(int(rng.integers(1, 4)))]
    depth = str(int(rng.integers(0, 5)))
    camera = _camera((float(rng.uniform(-3, 3)), float(rng.uniform(2, 6)), float(rng.uniform(9, 13))),
                     (0.0, 0.8, 0.0), (0, 1, 0), 42, res, name + ".png")
    for _ in range(crowd):
        kind = rng.random()
        if kind < 0.12 and insts:
            inst = dict(insts[int(rng.integers(len(insts)))])          # an exact copy: every hit ties
            inst["_id"] = str(next_id)
        elif kind < 0.2:
            inst = {"_id": str(next_id), "_baseMeshId": str(int(rng.choice(base_ids))), "_resetTransform": "false"}
        else:
            inst = {"_id": str(next_id), "_baseMeshId": str(int(rng.choice(base_ids))),
                    "_resetTransform": "true" if rng.random() < 0.6 else "false", "Transformations": new_transform()}
        if rng.random() < 0.7:
            inst["Material"] = str(int(rng.integers(1, 7)))
        insts.append(inst)
        if rng.random() < 0.15:
            base_ids.append(next_id)
        next_id += 1
    if insts:
        objects["MeshInstance"] = insts
    sc = {
        "BackgroundColor": "12 18 30", "MaxRecursionDepth": depth, "ShadowRayEpsilon": "1e-3",
        "IntersectionTestEpsilon": "1e-6",
        "Cameras": {"Camera": camera},
        "Lights": {"AmbientLight": "20 20 20", "PointLight": lights},
        "Materials": {"Material": mats},
        "Transformations": {k: v for k, v in (("Translation", trans), ("Scaling", scal), ("Rotation", rots)) if v},
        "VertexData": {"_data": _flat(np.array(verts, dtype=np.float64))},
        "Objects": objects}
    return _dump(os.path.join(out_dir, name + ".json"), sc)


def write_unit_scenes(out_dir):
    return {k: fn(out_dir) for k, fn in UNIT_SCENES.items()}
