"""rtx_render_noalloc (the entry point that allocates no device memory: every buffer of the
frame comes from one fixed array per GPU) against rtx_render, through the C-ABI: two calls
in a row, the second with a smaller input, each compared with the allocating entry point."""
import ctypes as C

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import _abi, scenes

pytestmark = pytest.mark.gpu


def _noalloc(lib, ds, cam, tw, th, first, stride):
    n = lib.rtx_render_output_bytes(C.byref(cam), tw, th, first, stride)
    out = np.full(n, 0xAB, dtype=np.uint8)
    st = _abi.Stats()
    rc = lib.rtx_render_noalloc(ds._h, C.byref(cam), tw, th, first, stride, out.ctypes.data, 0, C.byref(st))
    return rc, out, st.as_dict()


def test_noalloc_matches_render_on_two_calls_in_a_row(tmp_path):
    lib = rt.amd_lib()
    assert lib.rtx_noalloc_scratch_bytes() == 64 << 20
    # first call: an instanced scene (TLAS walk, instance bitmaps in the scratch), whole frame
    big = scenes.scene_grass(str(tmp_path), n_instances=300, res=(160, 120), name="na_grass")
    hs1 = rt.HostScene(big)
    ds1 = rt.DeviceScene(hs1.desc, 0)
    cam1, _ = hs1.camera(0)
    ref1, _, _ = ds1.render(cam1)
    rc, out1, st1 = _noalloc(lib, ds1, cam1, cam1.width, cam1.height, 0, 1)
    assert rc == 0, lib.rtx_last_error()
    assert np.array_equal(out1.reshape(ref1.shape), ref1)          # bit-exact: the same kernels, order-independent sums
    o_rgb, _, _, _ = ck.oracle_render(hs1.desc, cam1, want_hits=False)
    same, mx = ck.image_diff(out1.reshape(ref1.shape), o_rgb)
    assert mx <= 2 and same >= 0.999, (same, mx)
    # second call, smaller input, another scene: every third 16x8 tile of a 64x48 frame
    small = scenes.scene_tiny(str(tmp_path), res=(64, 48))
    hs2 = rt.HostScene(small)
    ds2 = rt.DeviceScene(hs2.desc, 0)
    cam2, _ = hs2.camera(0)
    ref2, _ = ds2.render_tiles(cam2, 16, 8, 1, 3)
    rc, out2, st2 = _noalloc(lib, ds2, cam2, 16, 8, 1, 3)
    assert rc == 0, lib.rtx_last_error()
    assert out2.size == ref2.size
    assert np.array_equal(out2, ref2)
    # and the first scene again through the same array: what the second call left behind does not matter
    rc, out3, _ = _noalloc(lib, ds1, cam1, cam1.width, cam1.height, 0, 1)
    assert rc == 0
    assert np.array_equal(out3.reshape(ref1.shape), ref1)
    for x in (ds1, ds2):
        x.close()
    hs1.close(); hs2.close()


def _free_device_bytes():
    """hipMemGetInfo of the current device, from the HIP runtime this process has loaded."""
    hip = C.CDLL("libamdhip64.so.7")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_allocates_no_device_memory(tmp_path):
    """The property the entry point exists for: on a scene that has never been rendered (no frame
    buffers exist yet), two calls leave the device's free memory where it was -- while the same
    frame through rtx_render, which has to allocate its queues, does not."""
    lib = rt.amd_lib()
    path = scenes.scene_grass(str(tmp_path), n_instances=300, res=(160, 120), name="na_free")
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    before = _free_device_bytes()
    rc, out, _ = _noalloc(lib, ds, cam, cam.width, cam.height, 0, 1)
    assert rc == 0, lib.rtx_last_error()
    rc, out2, _ = _noalloc(lib, ds, cam, 32, 8, 1, 2)
    assert rc == 0, lib.rtx_last_error()
    assert _free_device_bytes() == before
    ref, _, _ = ds.render(cam)                       # the allocating entry point, for contrast
    assert _free_device_bytes() < before
    assert np.array_equal(out.reshape(ref.shape), ref)
    ds.close(); hs.close()


def test_noalloc_rejects_what_does_not_fit(tmp_path):
    lib = rt.amd_lib()
    path = scenes.scene_tiny(str(tmp_path), res=(1024, 1024))
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    rc, _, _ = _noalloc(lib, ds, cam, cam.width, cam.height, 0, 1)
    assert rc == -5
    msg = lib.rtx_last_error().decode()
    assert "rtx_render_noalloc" in msg and "bytes of scratch" in msg
    # one 128 x 128 tile of the same frame fits
    rc, out, _ = _noalloc(lib, ds, cam, 128, 128, 27, 64)
    assert rc == 0, lib.rtx_last_error()
    ref, _ = ds.render_tiles(cam, 128, 128, 27, 64)
    assert np.array_equal(out, ref)
    ds.close(); hs.close()
