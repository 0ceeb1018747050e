"""MI355X ray-casting core for the muratbayrktr/raytracer scene format.

Python is only the thin host mirror used by tests, bench and the multi-GPU driver:
``HostScene`` wraps the C++ scene pipeline (librtx_host.so: scene file -> finished
description, the data the reference holds at raytracer.cpp:246) and ``DeviceScene`` wraps
the HIP core (librtx_amd.so: the replacement for multiThreadedRayTracing,
raytracer.cpp:93-141).  Both are plain C-ABI libraries loaded with ctypes; there is no
CPU rendering path in this package -- ``DeviceScene`` raises when the HIP library or a
GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import Camera, HitId, Light, SceneDesc, Stats, HIT_DTYPE  # noqa: F401

__all__ = ["HostScene", "DeviceScene", "Frame", "render_file", "unpack_tiles", "assemble_tiles", "RtxError"]


class RtxError(RuntimeError):
    pass


_host_lib = None
_amd_lib = None


def host_lib():
    global _host_lib
    if _host_lib is None:
        _host_lib = _abi.load_host_lib()
    return _host_lib


def amd_lib():
    global _amd_lib
    if _amd_lib is None:
        try:
            _amd_lib = _abi.load_amd_lib()
        except OSError as e:  # fail loudly: no fallback renderer exists
            raise RtxError(f"HIP core library not loadable ({e}); build it with `make product`") from e
    return _amd_lib


class HostScene:
    """Scene file -> finished scene description (reference: raytracer.cpp:216-246)."""

    def __init__(self, json_path: str, use_bvh: bool = True, enable_cull: bool = False):
        self._lib = host_lib()
        h = C.c_void_p()
        rc = self._lib.rtx_host_load(os.fsencode(json_path), int(use_bvh), int(enable_cull), C.byref(h))
        if rc != 0:
            raise RtxError(self._lib.rtx_host_last_error().decode())
        self._h = h
        self.path = json_path

    def load_next(self, json_path: str) -> bool:
        """Next frame of an animation (rtx_host_load_next).  Returns True when the geometry was
        kept and only cameras and lights were re-read: a DeviceScene made from this scene then
        needs ``update_lights(hs.lights())`` only.  False: everything was reloaded and the
        description is a new one."""
        h, reused = C.c_void_p(), C.c_int(0)
        if self._lib.rtx_host_load_next(os.fsencode(json_path), self._h, C.byref(h), C.byref(reused)) != 0:
            raise RtxError(self._lib.rtx_host_last_error().decode())
        self._h = h
        self.path = json_path
        return bool(reused.value)

    def build_bvh_on_device(self, device: int = 0) -> float:
        """For a scene loaded with ``use_bvh=False``: build every mesh's BVH on the GPU
        (rtx_build_bvh) and attach it.  Returns the total build time in ms."""
        lib = amd_lib()
        d = self.desc.contents
        total = 0.0
        for i in range(d.n_meshes):
            m = d.meshes[i]
            if m.n_faces <= 0:
                continue
            nodes = (_abi.BvhNode * (2 * m.n_faces))()
            order = (C.c_int32 * m.n_faces)()
            n_nodes, ms = C.c_int32(0), C.c_double(0.0)
            rc = lib.rtx_build_bvh(device, d.vertices, d.n_vertices, m.faces, m.n_faces, nodes, C.byref(n_nodes), order,
                                   C.byref(ms))
            if rc != 0:
                raise RtxError(f"rtx_build_bvh failed ({rc}): {lib.rtx_last_error().decode()}")
            if self._lib.rtx_host_attach_bvh(self._h, i, nodes, n_nodes.value, order) != 0:
                raise RtxError(self._lib.rtx_host_last_error().decode())
            total += ms.value
        return total

    def lights(self):
        d = self.desc.contents
        return [d.lights[i] for i in range(d.n_lights)]

    @property
    def desc(self):
        return self._lib.rtx_host_desc(self._h)

    @property
    def n_cameras(self) -> int:
        return self._lib.rtx_host_camera_count(self._h)

    def camera(self, index: int = 0):
        cam = Camera()
        name = C.create_string_buffer(512)
        if self._lib.rtx_host_camera(self._h, index, C.byref(cam), name, 512) != 0:
            raise RtxError(self._lib.rtx_host_last_error().decode())
        return cam, name.value.decode()

    def timings(self):
        a, b = C.c_double(), C.c_double()
        self._lib.rtx_host_timings(self._h, C.byref(a), C.byref(b))
        return {"parse_ms": a.value, "preprocess_ms": b.value}

    def dump(self, path: str):
        if self._lib.rtx_host_dump(self._h, os.fsencode(path)) != 0:
            raise RtxError(self._lib.rtx_host_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtx_host_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ppm(path: str, rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    if host_lib().rtx_host_write_ppm(os.fsencode(path), rgb.ctypes.data, w, h) != 0:
        raise RtxError(host_lib().rtx_host_last_error().decode())


def write_png(path: str, rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    if host_lib().rtx_host_write_png(os.fsencode(path), rgb.ctypes.data, w, h) != 0:
        raise RtxError(host_lib().rtx_host_last_error().decode())


class DeviceScene:
    """Device copy of a scene + render calls (reference: multiThreadedRayTracing)."""

    def __init__(self, desc, device: int = 0, wide_blas: bool = False):
        """wide_blas: also build the eight-wide float-box records (RTX_SCENE_WIDE_BLAS), which
        set_wide_blas then switches the mesh traversal to."""
        self._lib = amd_lib()
        h = C.c_void_p()
        if wide_blas:
            # (a copy of the description with the build flag set; the arrays it points to are shared)
            src = desc.contents if hasattr(desc, "contents") else desc
            mine = _abi.SceneDesc()
            C.memmove(C.byref(mine), C.byref(src), C.sizeof(_abi.SceneDesc))
            mine.build_flags |= _abi.SCENE_WIDE_BLAS
            desc = C.pointer(mine)
        rc = self._lib.rtx_scene_create(desc, device, C.byref(h))
        if rc != 0:
            raise RtxError(f"rtx_scene_create failed ({rc}): {self._lib.rtx_last_error().decode()}")
        self._h = h
        self.device = device

    def set_counters(self, enable: bool):
        self._lib.rtx_scene_set_counters(self._h, int(enable))

    def set_consistent(self, enable: bool):
        """Consistent mode (not the reference's results for scaled instances; see rtx_amd.h)."""
        if self._lib.rtx_scene_set_consistent(self._h, int(enable)) != 0:
            raise RtxError(self._lib.rtx_last_error().decode())

    def set_generation_timing(self, enable: bool):
        """Events round every generation's trace launches (stats["trace_ms"]; on by default)."""
        self._lib.rtx_scene_set_generation_timing(self._h, int(enable))

    def set_queue_limit(self, nbytes: int):
        self._lib.rtx_scene_set_queue_limit(self._h, int(nbytes))

    def set_wide_blas(self, enable: bool):
        """Mesh traversal over the eight-wide float-box records (same pictures; rtx_scene_set_wide_blas)."""
        if self._lib.rtx_scene_set_wide_blas(self._h, int(enable)) != 0:
            raise RtxError(self._lib.rtx_last_error().decode())

    def record_bytes(self) -> int:
        """Bytes of the traversal's records (nodes, triangles, object records) on the device."""
        return int(self._lib.rtx_scene_record_bytes(self._h))

    def set_debug(self, flags: int):
        """Development switches (``_abi.DEBUG_*``; rtx_scene_set_debug): same pictures, other paths."""
        if self._lib.rtx_scene_set_debug(self._h, int(flags)) != 0:
            raise RtxError(self._lib.rtx_last_error().decode())

    def frame(self) -> "Frame":
        """A frame context of this scene: renders submitted through different frames run concurrently."""
        return Frame(self)

    def update_lights(self, lights):
        arr = (Light * len(lights))(*lights)
        if self._lib.rtx_scene_update_lights(self._h, arr, len(lights)) != 0:
            raise RtxError(self._lib.rtx_last_error().decode())

    def render(self, cam: Camera, want_hits: bool = False, out=None):
        """Whole frame -> (rgb[h][w][3] uint8, hit records or None, stats dict).
        ``out``: a C-contiguous uint8 array of h*w*3 bytes to receive the image instead of a new
        one (reused from frame to frame; page-locked memory makes the download a direct DMA)."""
        if out is None:
            rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
        else:
            if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or out.size != cam.height * cam.width * 3:
                raise RtxError("out must be a C-contiguous uint8 array of height * width * 3 bytes")
            rgb = out.reshape(cam.height, cam.width, 3)
        hits = np.empty(cam.height * cam.width, dtype=HIT_DTYPE) if want_hits else None
        st = Stats()
        rc = self._lib.rtx_render(self._h, C.byref(cam), cam.width, cam.height, 0, 1,
                                  rgb.ctypes.data, 0, hits.ctypes.data if want_hits else None,
                                  C.byref(st))
        if rc != 0:
            raise RtxError(f"rtx_render failed ({rc}): {self._lib.rtx_last_error().decode()}")
        return rgb, hits, st.as_dict()

    def tile_bytes(self, cam, tile_w, tile_h, first, stride) -> int:
        return int(self._lib.rtx_render_output_bytes(C.byref(cam), tile_w, tile_h, first, stride))

    def render_tiles(self, cam: Camera, tile_w: int, tile_h: int, first: int, stride: int,
                     out_ptr: int | None = None, out_is_device: bool = False):
        """Interleaved tile subset -> packed uint8 buffer (or into ``out_ptr``)."""
        n = self.tile_bytes(cam, tile_w, tile_h, first, stride)
        if n < 0:
            raise RtxError(f"bad tiling: {self._lib.rtx_last_error().decode()}")
        st = Stats()
        buf = None
        if out_ptr is None:
            buf = np.empty(n, dtype=np.uint8)
            out_ptr, out_is_device = buf.ctypes.data, False
        rc = self._lib.rtx_render(self._h, C.byref(cam), tile_w, tile_h, first, stride,
                                  C.c_void_p(out_ptr), int(out_is_device), None, C.byref(st))
        if rc != 0:
            raise RtxError(f"rtx_render failed ({rc}): {self._lib.rtx_last_error().decode()}")
        return buf, st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            for fr in list(getattr(self, "_frames", [])):
                fr.close()
            self._lib.rtx_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Frame:
    """One render in flight on a DeviceScene (rtx_frame_create / rtx_render_async / rtx_frame_wait).
    ``submit`` returns as soon as the work is queued; ``wait`` returns (rgb, hits, stats) like
    ``DeviceScene.render``.  Several frames of one scene overlap on the GPU."""

    def __init__(self, scene: DeviceScene):
        self._lib = scene._lib
        self._scene = scene
        h = C.c_void_p()
        if self._lib.rtx_frame_create(scene._h, C.byref(h)) != 0:
            raise RtxError(f"rtx_frame_create failed: {self._lib.rtx_last_error().decode()}")
        self._h = h
        self._out = None
        if not hasattr(scene, "_frames"):
            scene._frames = []
        scene._frames.append(self)

    def submit(self, cam: Camera, want_hits: bool = False, out=None):
        if out is None:
            rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
        else:
            if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or out.size != cam.height * cam.width * 3:
                raise RtxError("out must be a C-contiguous uint8 array of height * width * 3 bytes")
            rgb = out.reshape(cam.height, cam.width, 3)
        hits = np.empty(cam.height * cam.width, dtype=HIT_DTYPE) if want_hits else None
        rc = self._lib.rtx_render_async(self._h, C.byref(cam), cam.width, cam.height, 0, 1, rgb.ctypes.data, 0,
                                        hits.ctypes.data if want_hits else None)
        if rc != 0:
            raise RtxError(f"rtx_render_async failed ({rc}): {self._lib.rtx_last_error().decode()}")
        self._out = (rgb, hits)

    def submit_tiles(self, cam: Camera, tile_w: int, tile_h: int, first: int, stride: int, out_ptr: int, out_is_device: bool):
        rc = self._lib.rtx_render_async(self._h, C.byref(cam), tile_w, tile_h, first, stride, C.c_void_p(out_ptr),
                                        int(out_is_device), None)
        if rc != 0:
            raise RtxError(f"rtx_render_async failed ({rc}): {self._lib.rtx_last_error().decode()}")
        self._out = (None, None)

    def wait(self):
        st = Stats()
        rc = self._lib.rtx_frame_wait(self._h, C.byref(st))
        if rc != 0:
            self._out = None
            raise RtxError(f"rtx_frame_wait failed ({rc}): {self._lib.rtx_last_error().decode()}")
        rgb, hits = self._out
        self._out = None
        return rgb, hits, st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtx_frame_destroy(self._h)
            self._h = None
            try:
                self._scene._frames.remove(self)
            except ValueError:
                pass

    def __del__(self):
        try:
            if getattr(self._scene, "_h", None):
                self.close()
        except Exception:
            pass


def unpack_tiles(packed: np.ndarray, width, height, tile_w, tile_h, first, stride, frame: np.ndarray):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    rc = amd_lib().rtx_unpack_tiles(packed.ctypes.data, width, height, tile_w, tile_h, first, stride,
                                    frame.ctypes.data)
    if rc != 0:
        raise RtxError(amd_lib().rtx_last_error().decode())


def assemble_tiles(device: int, part_ptrs, width, height, tile_w, tile_h, frame: np.ndarray):
    """Rank 0 of a tiled frame: scatter the packed tile buffers of all ranks, given as DEVICE
    pointers on `device` (0 / None for a rank without tiles), into `frame` (host uint8
    [height][width][3]) with one kernel and one copy (rtx_assemble_tiles).  The buffers must be
    complete: synchronise whatever filled them first."""
    n = len(part_ptrs)
    arr = (C.c_void_p * n)(*[C.c_void_p(int(p) if p else 0) for p in part_ptrs])
    if frame.dtype != np.uint8 or not frame.flags["C_CONTIGUOUS"] or frame.size != height * width * 3:
        raise RtxError("frame must be a C-contiguous uint8 array of height * width * 3 bytes")
    rc = amd_lib().rtx_assemble_tiles(device, arr, n, width, height, tile_w, tile_h, frame.ctypes.data, 0)
    if rc != 0:
        raise RtxError(amd_lib().rtx_last_error().decode())


def render_file(json_path: str, device: int = 0, camera: int = 0, enable_cull: bool = False):
    """Convenience: load, upload, render one camera; returns (rgb, stats)."""
    hs = HostScene(json_path, True, enable_cull)
    ds = DeviceScene(hs.desc, device)
    cam, _ = hs.camera(camera)
    rgb, _, st = ds.render(cam)
    ds.close()
    hs.close()
    return rgb, st
