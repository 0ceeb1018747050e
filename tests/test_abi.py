"""CPU-side checks of the C-ABI: libraries load, export every declared symbol, and the
ctypes mirrors have the sizes the C headers have.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from raytracer_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", text)))


def test_host_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(_abi.LIB_DIR, "librtx_host.so"))
    names = _declared("rtx_host.h")
    assert set(names) == set(_abi.HOST_SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None


def test_device_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(_abi.LIB_DIR, "librtx_amd.so"))
    names = _declared("rtx_amd.h")
    assert set(names) == set(_abi.AMD_SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_sizes_match_the_headers():
    src = r'''
    #include <stdio.h>
    #include "rtx_amd.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rtx_material), sizeof(rtx_light),
        sizeof(rtx_camera), sizeof(rtx_xform), sizeof(rtx_plane), sizeof(rtx_sphere), sizeof(rtx_triangle),
        sizeof(rtx_bvh_node), sizeof(rtx_mesh), sizeof(rtx_instance), sizeof(rtx_scene_desc), sizeof(rtx_hit_id),
        sizeof(rtx_stats));
      return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    want = [C.sizeof(t) for t in (_abi.Material, _abi.Light, _abi.Camera, _abi.Xform, _abi.Plane, _abi.Sphere,
                                  _abi.Triangle, _abi.BvhNode, _abi.Mesh, _abi.Instance, _abi.SceneDesc,
                                  _abi.HitId, _abi.Stats)]
    assert got == want
    assert C.sizeof(_abi.BvhNode) == 72          # the reference's BVHNode (bvh.h:50-61)


def test_device_entry_points_fail_loudly_without_a_gpu(unit_scenes):
    """On a box without a HIP device the product must refuse, not fall back."""
    import raytracer_amd as rt
    lib = rt.amd_lib()
    if lib.rtx_device_count() > 0:
        pytest.skip("a GPU is present")
    hs = rt.HostScene(unit_scenes["tiny"])
    with pytest.raises(rt.RtxError) as e:
        rt.DeviceScene(hs.desc, 0)
    assert "no HIP device" in str(e.value)


def test_product_never_touches_the_oracle():
    """Nothing under raytracer_amd/ or include/ may reference the checkers."""
    bad = []
    for base in ("raytracer_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".c")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"rt_oracle|librt_oracle|oracle/|_checkers", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
