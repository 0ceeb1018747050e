"""Host scene pipeline (JSON/PLY parsing, BVH build, transform + normal precompute):
bit-for-bit against the reference's own Scene object where the compiled reference is
available, plus format/edge-case tests that need no reference."""
import json
import os

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import scenes


@pytest.mark.skipif(not ck.have_reference(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("name", list(scenes.UNIT_SCENES))
def test_finished_scene_is_bit_identical_to_the_reference(unit_scenes, tmp_path, name):
    path = unit_scenes[name]
    ck.reference_dump(path, str(tmp_path / "ref"), render=False)
    hs = rt.HostScene(path)
    hs.dump(str(tmp_path / "mine.scene"))
    a = ck.read_sections(str(tmp_path / "ref.scene"))
    b = ck.read_sections(str(tmp_path / "mine.scene"))
    for sec, blob in a.items():
        if sec.endswith(".cornernrm") and sec not in b:
            continue            # vertex normals are only built for smooth meshes here
        assert sec in b, f"section {sec} missing"
        assert blob == b[sec], f"section {sec} differs"
    assert not [s for s in b if s not in a]


@pytest.mark.skipif(not ck.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_without_bvh_matches_reference_scene(unit_scenes, tmp_path):
    path = unit_scenes["tiny"]
    ck.reference_dump(path, str(tmp_path / "ref"), bvh=False, render=False)
    hs = rt.HostScene(path, use_bvh=False)
    hs.dump(str(tmp_path / "mine.scene"))
    a = ck.read_sections(str(tmp_path / "ref.scene"))
    b = ck.read_sections(str(tmp_path / "mine.scene"))
    assert not any(k.endswith(".bvhbox") for k in b)
    for sec, blob in a.items():
        if sec.endswith(".cornernrm") and sec not in b:
            continue
        assert blob == b[sec], sec


def test_ply_formats_agree(tmp_path):
    v, f = scenes.bumpy_sphere(24, 12, seed=3)
    descs = []
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        d = tmp_path / fmt
        d.mkdir()
        scenes.write_ply(str(d / "m.ply"), v, f, fmt=fmt)
        sc = {"Scene": {"Cameras": {"Camera": {"_id": "1", "Position": "0 0 5", "Gaze": "0 0 -1", "Up": "0 1 0",
                                                 "NearPlane": "-1 1 -1 1", "NearDistance": "1",
                                                 "ImageResolution": "8 8", "ImageName": "x.png"}},
                        "Materials": {"Material": {"_id": "1", "AmbientReflectance": "1 1 1",
                                                   "DiffuseReflectance": "1 1 1", "SpecularReflectance": "0 0 0"}},
                        "VertexData": {"_data": "0 0 0"},
                        "Objects": {"Mesh": {"_id": "1", "Material": "1", "_shadingMode": "smooth",
                                             "Faces": {"_plyFile": "m.ply"}}}}}
        (d / "s.json").write_text(json.dumps(sc))
        hs = rt.HostScene(str(d / "s.json"))
        hs.dump(str(d / "dump"))
        descs.append(ck.read_sections(str(d / "dump")))
    for other in descs[1:]:
        assert other.keys() == descs[0].keys()
        for k in other:
            assert other[k] == descs[0][k], k
    verts = np.frombuffer(descs[0]["vertices"], dtype=np.float64).reshape(-1, 3)
    assert len(verts) == len(v) + 1                       # PLY vertices appended to the pool
    assert np.array_equal(verts[1:], v.astype(np.float64))
    faces = np.frombuffer(descs[0]["mesh0.faces"], dtype=np.int32).reshape(-1, 3)
    assert np.array_equal(faces, f + 1)                   # indices offset by the pool size


def test_loader_errors_are_reported(tmp_path):
    with pytest.raises(rt.RtxError, match="cannot be loaded"):
        rt.HostScene(str(tmp_path / "missing.json"))
    (tmp_path / "bad.json").write_text('{"NotScene": {}}')
    with pytest.raises(rt.RtxError, match="Scene root"):
        rt.HostScene(str(tmp_path / "bad.json"))
    (tmp_path / "trunc.json").write_text('{"Scene": {')
    with pytest.raises(rt.RtxError, match="JSON"):
        rt.HostScene(str(tmp_path / "trunc.json"))
    (tmp_path / "ply.json").write_text(json.dumps({"Scene": {"Objects": {"Mesh": {
        "_id": "1", "_shadingMode": "flat", "Faces": {"_plyFile": "nope.ply"}}}}}))
    with pytest.raises(rt.RtxError, match="PLY file does not exist"):
        rt.HostScene(str(tmp_path / "ply.json"))


def test_single_object_and_array_forms_are_equivalent(tmp_path):
    def scene(as_array):
        light = {"_id": "1", "Position": "1 2 3", "Intensity": "5 5 5"}
        mat = {"_id": "1", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "1 1 1", "SpecularReflectance": "0 0 0"}
        sph = {"_id": "1", "Material": "1", "Center": "1", "Radius": "1"}
        wrap = (lambda x: [x]) if as_array else (lambda x: x)
        return {"Scene": {"Lights": {"PointLight": wrap(light)}, "Materials": {"Material": wrap(mat)},
                          "VertexData": {"_data": "0 0 0"}, "Objects": {"Sphere": wrap(sph)},
                          "Cameras": {"Camera": wrap({"_id": "1", "Position": "0 0 5", "Gaze": "0 0 -1", "Up": "0 1 0",
                                                      "NearPlane": "-1 1 -1 1", "NearDistance": "1",
                                                      "ImageResolution": "4 4", "ImageName": "x.png"})}}}
    dumps = []
    for k, arr in enumerate((False, True)):
        p = tmp_path / f"s{k}.json"
        p.write_text(json.dumps(scene(arr)))
        hs = rt.HostScene(str(p))
        hs.dump(str(tmp_path / f"d{k}"))
        dumps.append(ck.read_sections(str(tmp_path / f"d{k}")))
    assert dumps[0] == dumps[1]


def test_ppm_and_png_writers(tmp_path):
    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    rt.write_ppm(str(tmp_path / "img.png"), rgb)            # extension is replaced, as the reference does
    text = (tmp_path / "img.ppm").read_text()
    lines = text.split("\n")
    assert lines[0] == "P3" and lines[1] == "7 5" and lines[2] == "255"
    assert lines[3] == " ".join(str(int(v)) for v in rgb[0].reshape(-1)) + " "
    assert text.endswith("\n\n")
    rt.write_png(str(tmp_path / "img2.png"), rgb)
    from PIL import Image
    back = np.asarray(Image.open(str(tmp_path / "img2.png")))
    assert np.array_equal(back, rgb)
