"""ctypes mirrors of the C-ABI structs in include/rtx_amd.h and include/rtx_host.h."""
from __future__ import annotations

import ctypes as C
import os

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")

c_double3 = C.c_double * 3
c_double4 = C.c_double * 4
c_double16 = C.c_double * 16

KIND_NONE, KIND_PLANE, KIND_SPHERE, KIND_TRIANGLE, KIND_MESH = range(5)


class Material(C.Structure):
    _fields_ = [("ambient", c_double3), ("diffuse", c_double3), ("specular", c_double3),
                ("mirror", c_double3), ("phong_exponent", C.c_double),
                ("refraction_index", C.c_double), ("absorption_index", C.c_double),
                ("absorption_coefficient", c_double3), ("type", C.c_int32),
                ("is_mirror", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("position", c_double3), ("intensity", c_double3)]


class Camera(C.Structure):
    _fields_ = [("position", c_double3), ("gaze", c_double3), ("up", c_double3),
                ("near_plane", c_double4), ("near_distance", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32)]


class Xform(C.Structure):
    _fields_ = [("transform", c_double16), ("inverse", c_double16), ("normal", c_double16),
                ("bounds_min", c_double3), ("bounds_max", c_double3),
                ("has_bounds", C.c_int32), ("negative_scale", C.c_int32)]


class Plane(C.Structure):
    _fields_ = [("normal", c_double3), ("point", C.c_int32), ("material", C.c_int32),
                ("xform", C.c_int32), ("pad", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("radius", C.c_double), ("center", C.c_int32), ("material", C.c_int32),
                ("xform", C.c_int32), ("pad", C.c_int32)]


class Triangle(C.Structure):
    _fields_ = [("v", C.c_int32 * 3), ("material", C.c_int32), ("xform", C.c_int32),
                ("pad", C.c_int32)]


class BvhNode(C.Structure):
    _fields_ = [("bounds_min", c_double3), ("bounds_max", c_double3), ("left", C.c_int32),
                ("right", C.c_int32), ("first_face", C.c_int32), ("face_count", C.c_int32),
                ("split_axis", C.c_int32), ("pad", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("faces", C.POINTER(C.c_int32)), ("nodes", C.POINTER(BvhNode)),
                ("face_indices", C.POINTER(C.c_int32)), ("vertex_normals", C.POINTER(C.c_double)),
                ("n_faces", C.c_int32), ("n_nodes", C.c_int32), ("material", C.c_int32),
                ("smooth", C.c_int32), ("xform", C.c_int32), ("pad", C.c_int32)]


class Instance(C.Structure):
    _fields_ = [("base_mesh", C.c_int32), ("material", C.c_int32), ("xform", C.c_int32),
                ("pad", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("max_recursion_depth", C.c_int32),
                ("enable_back_face_culling", C.c_int32), ("build_flags", C.c_int32),
                ("background", c_double3), ("ambient", c_double3),
                ("shadow_ray_epsilon", C.c_double), ("intersection_test_epsilon", C.c_double),
                ("vertices", C.POINTER(C.c_double)), ("n_vertices", C.c_int64),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_int32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_int32),
                ("xforms", C.POINTER(Xform)), ("n_xforms", C.c_int32),
                ("planes", C.POINTER(Plane)), ("n_planes", C.c_int32),
                ("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_int32),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_int32),
                ("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_int32),
                ("instances", C.POINTER(Instance)), ("n_instances", C.c_int32)]


class HitId(C.Structure):
    _fields_ = [("kind", C.c_int32), ("container", C.c_int32), ("face", C.c_int32),
                ("src", C.c_int32), ("distance", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("secondary_rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("world_box_tests", C.c_uint64),
                ("object_space_traversals", C.c_uint64), ("tlas_node_visits", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("generations", C.c_uint32), ("queue_overflows", C.c_uint32),
                ("wave_inner_steps", C.c_uint64), ("wave_inner_lanes", C.c_uint64),
                ("wave_leaf_steps", C.c_uint64), ("wave_leaf_lanes", C.c_uint64),
                ("trace_ms", C.c_double), ("trace_launches", C.c_uint32), ("instance_path", C.c_uint32),
                ("record_fetches", C.c_uint64), ("wide_blas", C.c_uint32), ("wide_blas_redone", C.c_uint32),
                ("shadow_rays_settled_early", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


HIT_DTYPE = [("kind", "<i4"), ("container", "<i4"), ("face", "<i4"), ("src", "<i4"),
             ("distance", "<f8")]

# Symbols every library must export (checked by the CPU-side ABI test).
AMD_SYMBOLS = ["rtx_build_bvh", "rtx_scene_create", "rtx_scene_update_lights", "rtx_render",
               "rtx_render_noalloc", "rtx_noalloc_scratch_bytes",
               "rtx_render_output_bytes", "rtx_unpack_tiles", "rtx_assemble_tiles", "rtx_scene_set_counters", "rtx_scene_set_consistent", "rtx_scene_set_generation_timing",
               "rtx_scene_set_queue_limit", "rtx_scene_set_debug", "rtx_scene_record_bytes", "rtx_scene_set_wide_blas", "rtx_scene_destroy", "rtx_last_error",
               "rtx_device_count", "rtx_frame_create", "rtx_render_async", "rtx_frame_wait", "rtx_frame_destroy"]

# rtx_scene_set_debug switches (include/rtx_amd.h)
DEBUG_NO_SCATTER, DEBUG_POISON_QUEUES, DEBUG_GEN_TIMING, DEBUG_NO_WIDE = 1, 2, 4, 8
DEBUG_NO_SEARCH, DEBUG_NO_WIDE_OLD, DEBUG_ALL_GENERATIONS, DEBUG_HALF_GRID, DEBUG_FAIL_WIDE_BLAS_CHECK = 16, 32, 64, 128, 256
DEBUG_NO_EARLY_SHADOW = 512
SCENE_WIDE_BLAS = 1          # rtx_scene_desc::build_flags
HOST_SYMBOLS = ["rtx_host_load", "rtx_host_load_next", "rtx_host_attach_bvh", "rtx_host_desc", "rtx_host_camera_count", "rtx_host_camera",
                "rtx_host_timings", "rtx_host_dump", "rtx_host_free", "rtx_host_write_ppm",
                "rtx_host_write_png", "rtx_host_last_error"]


def load_host_lib():
    lib = C.CDLL(os.path.join(LIB_DIR, "librtx_host.so"))
    lib.rtx_host_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.rtx_host_load.restype = C.c_int
    lib.rtx_host_load_next.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.rtx_host_load_next.restype = C.c_int
    lib.rtx_host_attach_bvh.argtypes = [C.c_void_p, C.c_int, C.POINTER(BvhNode), C.c_int, C.POINTER(C.c_int32)]
    lib.rtx_host_attach_bvh.restype = C.c_int
    lib.rtx_host_desc.argtypes = [C.c_void_p]
    lib.rtx_host_desc.restype = C.POINTER(SceneDesc)
    lib.rtx_host_camera_count.argtypes = [C.c_void_p]
    lib.rtx_host_camera_count.restype = C.c_int
    lib.rtx_host_camera.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera), C.c_char_p, C.c_int]
    lib.rtx_host_camera.restype = C.c_int
    lib.rtx_host_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rtx_host_timings.restype = C.c_int
    lib.rtx_host_dump.argtypes = [C.c_void_p, C.c_char_p]
    lib.rtx_host_dump.restype = C.c_int
    lib.rtx_host_free.argtypes = [C.c_void_p]
    lib.rtx_host_free.restype = None
    lib.rtx_host_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    lib.rtx_host_write_ppm.restype = C.c_int
    lib.rtx_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    lib.rtx_host_write_png.restype = C.c_int
    lib.rtx_host_last_error.argtypes = []
    lib.rtx_host_last_error.restype = C.c_char_p
    return lib


def load_amd_lib():
    """Load the device library.  Raises OSError when it (or the HIP runtime) is missing:
    there is no CPU fallback behind this API."""
    # RTX_AMD_LIB selects another build of the same library (kernel experiments).
    lib = C.CDLL(os.environ.get("RTX_AMD_LIB") or os.path.join(LIB_DIR, "librtx_amd.so"))
    lib.rtx_build_bvh.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.c_int32,
                                  C.POINTER(BvhNode), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    lib.rtx_build_bvh.restype = C.c_int
    lib.rtx_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    lib.rtx_scene_create.restype = C.c_int
    lib.rtx_scene_update_lights.argtypes = [C.c_void_p, C.POINTER(Light), C.c_int32]
    lib.rtx_scene_update_lights.restype = C.c_int
    lib.rtx_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Stats)]
    lib.rtx_render.restype = C.c_int
    lib.rtx_render_noalloc.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_int32, C.POINTER(Stats)]
    lib.rtx_render_noalloc.restype = C.c_int
    lib.rtx_noalloc_scratch_bytes.argtypes = []
    lib.rtx_noalloc_scratch_bytes.restype = C.c_int64
    lib.rtx_render_output_bytes.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32]
    lib.rtx_render_output_bytes.restype = C.c_int64
    lib.rtx_unpack_tiles.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p]
    lib.rtx_unpack_tiles.restype = C.c_int
    lib.rtx_assemble_tiles.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_int32]
    lib.rtx_assemble_tiles.restype = C.c_int
    lib.rtx_scene_set_counters.argtypes = [C.c_void_p, C.c_int32]
    lib.rtx_scene_set_counters.restype = C.c_int
    lib.rtx_scene_set_consistent.argtypes = [C.c_void_p, C.c_int32]
    lib.rtx_scene_set_consistent.restype = C.c_int
    lib.rtx_scene_set_generation_timing.argtypes = [C.c_void_p, C.c_int32]
    lib.rtx_scene_set_generation_timing.restype = C.c_int
    lib.rtx_scene_set_queue_limit.argtypes = [C.c_void_p, C.c_int64]
    lib.rtx_scene_set_queue_limit.restype = C.c_int
    lib.rtx_scene_set_wide_blas.argtypes = [C.c_void_p, C.c_int32]
    lib.rtx_scene_set_wide_blas.restype = C.c_int
    lib.rtx_scene_record_bytes.argtypes = [C.c_void_p]
    lib.rtx_scene_record_bytes.restype = C.c_int64
    lib.rtx_scene_set_debug.argtypes = [C.c_void_p, C.c_uint32]
    lib.rtx_scene_set_debug.restype = C.c_int
    lib.rtx_frame_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.rtx_frame_create.restype = C.c_int
    lib.rtx_render_async.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rtx_render_async.restype = C.c_int
    lib.rtx_frame_wait.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.rtx_frame_wait.restype = C.c_int
    lib.rtx_frame_destroy.argtypes = [C.c_void_p]
    lib.rtx_frame_destroy.restype = None
    lib.rtx_scene_destroy.argtypes = [C.c_void_p]
    lib.rtx_scene_destroy.restype = None
    lib.rtx_last_error.argtypes = []
    lib.rtx_last_error.restype = C.c_char_p
    lib.rtx_device_count.argtypes = []
    lib.rtx_device_count.restype = C.c_int
    return lib
